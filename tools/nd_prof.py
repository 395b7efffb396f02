"""one-mesh driver for profiling the nested-dissection re-solve: python tools/nd_prof.py [cfg] [leaf] [solves]"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential
from largesteps.solvers import NestedDissectionSolver
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "cfg4_plane1m"
leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 96
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
arity = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device("cuda:0")
v, f, cfg = synthetic.config_mesh(cfg_name)
tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
u = to_differential(M, tv)
s = NestedDissectionSolver(M, leaf_size=leaf, arity=arity)
for _ in range(3): x = s.solve(u)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): x = s.solve(u)
torch.cuda.synchronize()
inf = s.info()
print(f"{cfg_name} leaf {leaf} arity {arity}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms/solve, err {float((x - tv).abs().max()):.2e}, levels={inf.get('levels')}, words/V={inf['factor_entries'] / v.shape[0]:.1f}, launches={inf['launches']}, build {s.build_seconds:.2f}s")
print("constructor:", {k: round(v, 3) for k, v in s.timings.items()})
