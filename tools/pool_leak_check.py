"""device memory left behind by constructions of the direct solver after ls_release_scratch: one-off (runtime initialisation) or growing (a leak)?"""
import gc, os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential
from largesteps.solvers import NestedDissectionSolver, release_scratch
dev = torch.device("cuda:0")
torch.zeros(1, device=dev); torch.cuda.synchronize()
free0, _ = torch.cuda.mem_get_info()
print("free at start", free0 >> 20, "MB")
for rep, n in enumerate((330, 300, 330, 350, 330, 330, 1000, 330)):
    v, f = synthetic.plane(n)
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, 20.0)
    u = to_differential(M, tv)
    a, _ = torch.cuda.mem_get_info()
    s = NestedDissectionSolver(M)
    x = s.solve(u)
    del s, x, u, M, tv, tf
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    held, _ = torch.cuda.mem_get_info()
    release_scratch(dev)
    freed, _ = torch.cuda.mem_get_info()
    print(f"n={n}: before the constructor {(free0 - a) >> 20} MB in use, after destroy {(free0 - held) >> 20} MB, after release_scratch {(free0 - freed) >> 20} MB")

print("---- the sequence of test_buffer_pool_between_constructions (no release in between, one explicit close)")
release_scratch(); torch.cuda.synchronize()
free0, _ = torch.cuda.mem_get_info()
for n in (330, 300, 330, 350, 330):
    v, f = synthetic.plane(n)
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, 20.0)
    u = to_differential(M, tv)
    s = NestedDissectionSolver(M)
    x = s.solve(u)
    if n == 350 and len(sys.argv) < 2:
        s.close(); s.close()
    del s, x, u, M, tv, tf
    gc.collect()
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    held, _ = torch.cuda.mem_get_info()
    print(f"n={n}: after destroy {(free0 - held) >> 20} MB in use")
release_scratch(dev)
freed, _ = torch.cuda.mem_get_info()
print(f"after release_scratch {(free0 - freed) >> 20} MB in use")

print("---- folded / rolled surfaces: the trial-cut path with the graph distances on the device (nd_embed_device, round 6): 12 constructions, then release")
release_scratch(); torch.cuda.synchronize(); gc.collect(); torch.cuda.empty_cache()
base, _ = torch.cuda.mem_get_info()
for rep, (kind, n) in enumerate([("scroll", 300), ("folded", 330), ("scroll", 300), ("shells", 120), ("scroll", 500), ("folded", 330)] * 2):
    v, f = (synthetic.scroll(n, 3) if kind == "scroll" else synthetic.folded_sheet(n) if kind == "folded" else synthetic.shells(n))
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, 20.0)
    u = to_differential(M, tv)
    s = NestedDissectionSolver(M)
    assert s.plan_quality["ordering"] == "trial-cuts", (kind, n, s.plan_quality)
    x = s.solve(u)
    assert float((x - tv).abs().max()) < 1e-4
    del s, x, u, M, tv, tf
gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
held, _ = torch.cuda.mem_get_info()
release_scratch(dev)
freed, _ = torch.cuda.mem_get_info()
print(f"after 12 constructions: {(base - held) >> 20} MB held (pool), after release_scratch {(base - freed) >> 20} MB")
assert (base - freed) < (64 << 20), "device memory is left behind by the trial-cut path"
