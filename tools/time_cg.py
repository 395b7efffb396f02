import sys, os, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential
from largesteps.solvers import ConjugateGradientSolver
dev = torch.device("cuda:0")
for name in ("cfg4_plane1m", "cfg3_dragon250k"):
    v, f, cfg = synthetic.config_mesh(name)
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
    u = to_differential(M, tv)
    for cheb in (True, False):
        s = ConjugateGradientSolver(M, chebyshev=cheb)
        x = s.solve(u); cold = dict(s.last_info)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(10):
            s.guess_fwd = None
            x = s.solve(u)
        torch.cuda.synchronize(); t_cold = (time.perf_counter() - t0) / 10
        # warm: rhs perturbed a little each step, as in an optimisation loop
        t0 = time.perf_counter()
        for i in range(10):
            x = s.solve(u + 1e-4 * (i + 1) * torch.ones_like(u))
        torch.cuda.synchronize(); t_warm = (time.perf_counter() - t0) / 10
        print(f"{name} 'CG' chebyshev={cheb} ({s.chebyshev}): cold {t_cold*1e3:.2f} ms ({cold['method']}, {cold['iterations']} its), warm {t_warm*1e3:.2f} ms ({s.last_info['method']}, {s.last_info['iterations']} its), err {float((x - tv).abs().max()):.1e}")
