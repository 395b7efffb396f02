"""The remesh cycle of the reference's optimisation loop (scripts/main.py:137-169) on the MI355X: every `period` steps the
mesh changes, and everything that hangs off the connectivity is rebuilt --
    remove_duplicates -> compute_matrix -> to_differential -> CholeskySolver (symbolic analysis + numeric factorisation)
    -> AdamUniform re-initialised
then `period` optimisation steps (from_differential -> normals -> loss -> backward incl. the adjoint solve -> AdamUniform).
No remesher is available in this environment (SURVEY: remesh_botsch is out of scope): a remesh event is emulated by handing
the loop the same surface as a triangle SOUP with the vertex storage reshuffled -- what a remesher returns -- so that all
rebuild work is real.      python tools/bench_remesh.py [workload] [period] [cycles]"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import numpy as np, torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.meshops import remove_duplicates
from largesteps.parameterize import to_differential, from_differential
from largesteps.normals import compute_face_normals, compute_vertex_normals
from largesteps.optimize import AdamUniform
workload = sys.argv[1] if len(sys.argv) > 1 else "cfg4_plane1m"
period = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
v, f, cfg = synthetic.config_mesh(workload)
lam = cfg["lambda_"] if cfg["lambda_"] is not None else 0.0
rng = np.random.default_rng(0)


def soup():
    sv = v[f.reshape(-1)]
    p = rng.permutation(sv.shape[0])
    inv = np.empty_like(p)
    inv[p] = np.arange(p.shape[0])
    return torch.from_numpy(sv[p].copy()).to(dev), torch.from_numpy(inv[np.arange(f.size).reshape(-1, 3)]).to(dev)


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


rebuild, steps_ms = [], []
for c in range(cycles):
    v_src, f_src = soup()
    t0 = sync()
    vu, fu, dup = remove_duplicates(v_src, f_src)
    t1 = sync()
    M = compute_matrix(vu, fu, lam, alpha=cfg["alpha"], cotan=cfg["cotan"])
    u = to_differential(M, vu).requires_grad_(True)
    t2 = sync()
    x = from_differential(M, u, "Cholesky")            # first call: constructs (factorises) the cached solver
    t3 = sync()
    opt = AdamUniform([u], 3e-2)
    target_v = vu + 0.01 * torch.randn_like(vu)
    target_n = compute_vertex_normals(vu, fu, compute_face_normals(vu, fu)).detach()
    t4 = sync()
    for _ in range(period):
        x = from_differential(M, u, "Cholesky")
        n = compute_vertex_normals(x, fu, compute_face_normals(x, fu))
        loss = (x - target_v).square().mean() + (n - target_n).square().mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
    t5 = sync()
    rebuild.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    steps_ms.append((t5 - t4) / period * 1e3)
    del M
# the first TWO cycles allocate (torch's caching allocator asks the driver for the blocks of cycle 0, and again for cycle 1 while cycle 0's
# tensors are still alive; a hipMalloc costs 3-10 ms on these hosts): a remesh loop is in its steady state from the third cycle on
skip = 2 if cycles > 3 else (1 if cycles > 1 else 0)
r = np.array(rebuild[skip:]).mean(0) * 1e3
s = float(np.mean(steps_ms[skip:]))
tot = float(r.sum())
print(f"{workload}: V={vu.shape[0]} (soup of {v_src.shape[0]} rows), remesh every {period} steps")
print(f"  rebuild per remesh: remove_duplicates {r[0]:.2f} ms | compute_matrix + to_differential {r[1]:.2f} ms | solver constructor "
      f"(analysis + factorisation, first solve) {r[2]:.2f} ms | optimizer + targets {r[3]:.2f} ms | total {tot:.1f} ms")
ctor = np.array(rebuild)[:, 2] * 1e3
steady = ctor[skip:]
print(f"  solver constructor per cycle (ms): {' '.join(f'{c:.1f}' for c in ctor)}   steady cycles: min {steady.min():.1f} median {np.median(steady):.1f} max {steady.max():.1f}")
print(f"  optimisation step: {s:.3f} ms  -> amortised over the period: {s + tot / period:.3f} ms per step ({100 * tot / period / (s + tot / period):.1f} % rebuild)")
