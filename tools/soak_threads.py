"""soak, two host threads: each makes, uses and drops direct solvers on its own torch stream at the same time (ctypes releases the GIL inside the library:
the analyses, factorisations and table uploads of the two threads really overlap). Every solve is checked.   python tools/soak_threads.py [iterations]"""
import gc, os, sys, threading, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import numpy as np, torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential
from largesteps.solvers import NestedDissectionSolver, release_scratch
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
torch.zeros(1, device=dev); torch.cuda.synchronize()
sizes = [40, 64, 100, 150, 220, 330]
host = {n: synthetic.plane(n) for n in sizes}
worst = [0.0, 0.0]
fail = []
def work(tid):
    try:
        rng = np.random.default_rng(100 + tid)
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            for it in range(iters):
                n = int(rng.choice(sizes))
                v, f = host[n]
                tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
                M = compute_matrix(tv, tf, float(rng.choice([5.0, 20.0, 50.0])))
                u = to_differential(M, tv)
                s = NestedDissectionSolver(M)
                e = float((s.solve(u) - tv).abs().max())
                worst[tid] = max(worst[tid], e)
                if e > 5e-5: fail.append((tid, it, n, e)); return
                if rng.random() < 0.5: s.close()
                del s, M, u
    except Exception as ex:                                   # noqa: BLE001
        fail.append((tid, repr(ex)))
t0 = time.perf_counter()
ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
for t in ts: t.start()
for t in ts: t.join()
torch.cuda.synchronize()
gc.collect(); release_scratch(dev)
print(f"2 threads x {iters} constructions in {time.perf_counter() - t0:.1f} s, worst |x - v| {max(worst):.2e}, failures: {fail if fail else 'none'}")
sys.exit(1 if fail else 0)
