import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "large-steps-pytorch_amd")]
import numpy as np, torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.solvers import CholeskySolver
from oracle import solve as osv
dev = torch.device("cuda:0")
def run(name, v, f, **kw):
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, **kw)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    b = (np.random.default_rng(0).standard_normal(v.shape)).astype(np.float32)
    x64 = osv.from_differential(idx[0], idx[1], val, b)
    out = []
    for direct in (True, False):
        try:
            s = CholeskySolver(M, direct=direct)
            x = s.solve(torch.from_numpy(b).to(dev)).cpu().numpy()
            out.append(f"{s.method}: {np.abs(x - x64).max() / np.abs(x64).max():.1e}")
        except Exception as e:
            out.append(f"direct={direct}: {type(e).__name__} {str(e)[:60]}")
    print(name, kw, " | ".join(out), flush=True)
v, f = synthetic.plane(300)
for lam in (1.0, 50.0, 1000.0, 1e4, 1e5):
    run("plane300", v, f, lambda_=lam)
v, f = synthetic.icosphere(60)
v = synthetic.perturb(v, radial=0.05, tangential=0.2, edge=0.02, seed=1)
for a in (0.5, 0.95, 0.99, 0.999):
    run("ico60cot", v, f, lambda_=0.0, alpha=a, cotan=True)
