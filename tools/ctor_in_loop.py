"""Where the wall clock of a solver construction INSIDE a remesh loop goes (round 6): the loop of tools/bench_remesh.py with
the rebuild split into its steps, the old solver either still referenced by the previous step's output (as in the reference's loop:
DifferentiableSolve keeps ctx.solver, scripts/main.py:172-208) or closed first.   python tools/ctor_in_loop.py [workload] [cycles]"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import numpy as np, torch
from largesteps import synthetic, parameterize
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential, from_differential
from largesteps.solvers import CholeskySolver
workload = sys.argv[1] if len(sys.argv) > 1 else "cfg4_plane1m"
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
v, f, cfg = synthetic.config_mesh(workload)
lam = cfg["lambda_"] if cfg["lambda_"] is not None else 0.0
tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for mode in ("old solver alive (held by the previous step's output)", "old solver closed before the rebuild"):
    x = None
    rows = []
    for c in range(cycles):
        t0 = sync()
        M = compute_matrix(tv, tf, lam, alpha=cfg["alpha"], cotan=cfg["cotan"])
        u = to_differential(M, tv).requires_grad_(True)
        t1 = sync()
        if mode.startswith("old solver closed") and x is not None:
            old = x.grad_fn
            x = None
            del old
        t2 = sync()
        solver = CholeskySolver(M)
        t3 = sync()
        parameterize.cache_put((id(M), "Cholesky"), solver, M)
        x_new = from_differential(M, u, "Cholesky")
        t4 = sync()
        x = x_new                                  # (the previous x, and with it the previous solver, dies here)
        t5 = sync()
        for _ in range(10):
            x = from_differential(M, u, "Cholesky")
            x.sum().backward()
        rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, solver.timings["plan_seconds"], solver.timings["factor_seconds"]))
        del M, solver
    r = np.array(rows[2:]) * 1e3
    print(f"{workload}, {mode}: steady cycles {len(r)}")
    for i, name in enumerate(("compute_matrix + to_differential", "release of the old solver", "CholeskySolver(M) [constructor]", "first solve", "old output dropped",
                              "  of the constructor: symbolic analysis (library's clock)", "  of the constructor: numeric + tables (library's clock)")):
        print(f"   {name:60s} min {r[:, i].min():7.2f}  median {np.median(r[:, i]):7.2f}  max {r[:, i].max():7.2f} ms")
