"""soak: many constructions / destructions of the direct solver in one process -- sizes in random order, several solvers alive at once, solves on two
streams, the pool released now and then -- every solve checked against the vertices it was made from. Prints the worst error and what is left on the device.
   python tools/soak_constructor.py [iterations] [seed]"""
import gc, os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import numpy as np, torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential
from largesteps.solvers import NestedDissectionSolver, release_scratch
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
side = torch.cuda.Stream(dev)
meshes = {}
def mesh(n):
    if n not in meshes:
        v, f = synthetic.plane(n)
        meshes[n] = (torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev))
    return meshes[n]
sizes = [24, 40, 64, 100, 150, 220, 330, 500]
torch.zeros(1, device=dev); torch.cuda.synchronize()
alive, worst, t0 = [], 0.0, time.perf_counter()
free0 = None
for it in range(iters):
    n = int(rng.choice(sizes))
    tv, tf = mesh(n)
    lam = float(rng.choice([5.0, 20.0, 50.0]))
    M = compute_matrix(tv, tf, lam)
    u = to_differential(M, tv)
    s = NestedDissectionSolver(M)
    x = s.solve(u)
    with torch.cuda.stream(side):
        side.wait_stream(torch.cuda.current_stream(dev))
        x2 = s.solve(u)
    torch.cuda.current_stream(dev).wait_stream(side)
    e = max(float((x - tv).abs().max()), float((x2 - tv).abs().max()))
    worst = max(worst, e)
    assert e <= 5e-5, (it, n, lam, e)
    alive.append((s, u, tv, n))
    for (so, uo, vo, no) in alive[:-1]:                 # the older solvers still answer
        if rng.random() < 0.3:
            eo = float((so.solve(uo) - vo).abs().max())
            assert eo <= 5e-5, ("older solver", it, no, eo)
    while len(alive) > int(rng.integers(1, 4)):
        so = alive.pop(int(rng.integers(0, len(alive))))[0]
        if rng.random() < 0.5: so.close()
        del so
    if rng.random() < 0.1:
        gc.collect(); release_scratch(dev)
    if it == 10:
        alive.clear(); gc.collect(); release_scratch(dev); torch.cuda.synchronize(); torch.cuda.empty_cache()
        free0, _ = torch.cuda.mem_get_info()
alive.clear(); del s, x, x2, u, M
gc.collect(); release_scratch(dev); torch.cuda.synchronize(); torch.cuda.empty_cache()
free1, _ = torch.cuda.mem_get_info()
print(f"{iters} constructions in {time.perf_counter() - t0:.1f} s, worst |x - v| {worst:.2e}, device memory left behind since iteration 10: {(free0 - free1) / 2**20:.1f} MB")
