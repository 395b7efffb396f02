"""re-solve time of the direct solver on small meshes against the leaf size (a leaf size >= V is ONE dense node: one launch):
python tools/leaf_sweep.py [solves]"""
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
cases = [int(a) for a in os.environ.get("LEAF_SWEEP_N", "8,24,32,48,64,80,100,128").split(",")]
if "LEAF_SWEEP_CONFIG" in os.environ:            # BASELINE configs instead of planes: LEAF_SWEEP_CONFIG=cfg2_bunny70k,cfg3_dragon250k
    cases = os.environ["LEAF_SWEEP_CONFIG"].split(",")
for n in cases:
    if isinstance(n, str):
        v, f, cfg = synthetic.config_mesh(n)
        V = v.shape[0]
        tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
        M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
    else:
        v, f = synthetic.plane(n)
        V = n * n
        tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
        M = compute_matrix(tv, tf, 20.0)
    u = to_differential(M, tv)
    row = []
    for leaf in ([int(a) for a in os.environ["LEAF_SWEEP_LEAF"].split(",")] if "LEAF_SWEEP_LEAF" in os.environ else (0, 64, 256, 1024, V // 4 + 1, V // 2 + 1, V)):          # 0 = the library picks
        if leaf > 8000:
            continue
        try:
            t0 = time.perf_counter(); s = NestedDissectionSolver(M, leaf_size=leaf, arity=int(os.environ.get("LEAF_SWEEP_ARITY", "4"))); x = s.solve(u); torch.cuda.synchronize(); tc = time.perf_counter() - t0
            for _ in range(5): x = s.solve(u)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n_solves): x = s.solve(u)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / n_solves * 1e6
            inf = s.info()
            err = float((x - tv).abs().max())
            row.append(f"leaf {leaf:5d}: {us:6.1f} us ({inf['launches']:2d} launches, {inf['levels']} levels, build {tc * 1e3:5.1f} ms, err {err:.1e})")
            del s
        except Exception as e:
            row.append(f"leaf {leaf:5d}: {type(e).__name__} {str(e)[:60]}")
    print(f"V {V:6d}: " + " | ".join(row), flush=True)
