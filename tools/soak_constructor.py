"""Constructor soak: build / solve / destroy the direct solver over and over on meshes of changing size (what a remesh loop does,
scripts/main.py:137-169) and watch the device memory the process holds and the answers: python tools/soak_constructor.py [rounds]"""
import gc, os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential
from largesteps.solvers import NestedDissectionSolver, release_scratch
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
sizes = [24, 33, 64, 100, 130, 200, 265, 330, 500, 707]           # 576 ... 500k vertices: every branch of the tree-picking rule
free0 = None
worst = 0.0
t0 = time.perf_counter()
for r in range(rounds):
    n = sizes[r % len(sizes)] + (r // len(sizes))                   # never the same size twice
    v, f = synthetic.plane(n)
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, 20.0 + r)
    u = to_differential(M, tv)
    s = NestedDissectionSolver(M)
    x = s.solve(u)
    err = float((x - tv).abs().max())
    worst = max(worst, err)
    assert err < 1e-4, (n, err)
    del s, x, u, M, tv, tf
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if r == len(sizes):                                             # after one pass over every size (one-off allocations are done)
        release_scratch(); free0 = torch.cuda.mem_get_info()[0]
    if r % 10 == 9: print(f"round {r + 1}: free device memory {free / 2**30:.2f} GiB of {total / 2**30:.0f}, worst error so far {worst:.1e}", flush=True)
held = torch.cuda.mem_get_info()[0]
release_scratch()                                                   # what the library's buffer pool still holds goes back first
free, _ = torch.cuda.mem_get_info()
print(f"buffer pool held {(free - held) / 2**20:.0f} MiB")
print(f"{rounds} constructions in {time.perf_counter() - t0:.1f} s; device memory not returned since round {len(sizes) + 1}: {(free0 - free) / 2**20:.1f} MiB; worst error {worst:.1e}")
