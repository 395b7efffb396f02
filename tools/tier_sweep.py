"""re-solve time of the direct solver against mesh size, tier height (LS_ND_TIER_H) and the lanes-along-the-reduction threshold
(LS_ND_LONG): python tools/tier_sweep.py [solves]"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential
from largesteps.solvers import NestedDissectionSolver
n_solves = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
for n in (24, 48, 80, 128, 200, 265, 400, 500, 700, 1000):
    v, f = synthetic.plane(n)
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, 20.0)
    u = to_differential(M, tv)
    row = []
    for H, long_red, long_up in ((-1, 64, 256), (-1, 64, 64), (-1, 64, 128), (-1, 32, 256), (2, 64, 256), (3, 64, 256), (3, 256, 256)):
        if True:
            os.environ["LS_ND_TIER_H"] = str(H); os.environ["LS_ND_LONG"] = str(long_red); os.environ["LS_ND_LONG_UP"] = str(long_up)
            s = NestedDissectionSolver(M)
            for _ in range(5): x = s.solve(u)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n_solves): x = s.solve(u)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n_solves * 1e3
            inf = s.info()
            row.append(f"H{H}->{inf['tier_levels']}/L{long_red}/U{long_up}: {ms * 1e3:6.1f} us ({inf['launches']:2d} launches)")
            err = float((x - tv).abs().max())
            assert err < 1e-4, err
            del s
    print(f"V {n * n:8d} levels {inf['levels']}: " + " | ".join(row), flush=True)
