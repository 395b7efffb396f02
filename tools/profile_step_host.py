"""where the HOST spends an eager optimisation step (tools/bench_step.py's loop body) at a mesh size where the step is host-bound:
   cProfile over N steps, sorted by own time, plus the wall time per step.   python tools/profile_step_host.py [workload] [steps]"""
import cProfile, io, os, pstats, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential, from_differential
from largesteps.normals import compute_face_normals, compute_vertex_normals
from largesteps.optimize import AdamUniform
workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2_bunny70k"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
v, f, cfg = synthetic.config_mesh(workload)
tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
target_n = compute_vertex_normals(tv, tf, compute_face_normals(tv, tf)).detach()
target_v = tv + 0.01 * torch.randn_like(tv)
u = to_differential(M, tv).clone().requires_grad_(True)
opt = AdamUniform([u], 3e-2)


def step():
    x = from_differential(M, u, "Cholesky")
    n = compute_vertex_normals(x, tf, compute_face_normals(x, tf))
    loss = (x - target_v).square().mean() + (n - target_n).square().mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss


for _ in range(10): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
t_host = (time.perf_counter() - t0) / steps
torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / steps
print(f"{workload}: {t_all * 1e3:.3f} ms per eager step, host returns after {t_host * 1e3:.3f} ms per step")
# pieces, each timed on the host alone (no synchronisation inside the loop)
def piece(name, fn, n=300):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    th = (time.perf_counter() - t0) / n
    torch.cuda.synchronize(); ta = (time.perf_counter() - t0) / n
    print(f"  {name:58s} host {th * 1e6:7.1f} us   with the device {ta * 1e6:7.1f} us")
with torch.no_grad():
    piece("from_differential (no autograd)", lambda: from_differential(M, u, "Cholesky"))
    x0 = from_differential(M, u, "Cholesky")
    piece("face + vertex normals (no autograd)", lambda: compute_vertex_normals(x0, tf, compute_face_normals(x0, tf)))
    n0 = compute_vertex_normals(x0, tf, compute_face_normals(x0, tf))
    piece("loss (two square().mean() and an add)", lambda: (x0 - target_v).square().mean() + (n0 - target_n).square().mean())
piece("from_differential (autograd node recorded)", lambda: from_differential(M, u, "Cholesky"))
def fwd_bwd():
    x = from_differential(M, u, "Cholesky"); x.sum().backward(); u.grad = None
piece("from_differential + sum + backward (adjoint solve)", fwd_bwd)
def normals_fb():
    xx = x0.clone().requires_grad_(True)
    n = compute_vertex_normals(xx, tf, compute_face_normals(xx, tf)); n.sum().backward()
piece("normals forward + backward", normals_fb)
u.grad = torch.zeros_like(u)
piece("AdamUniform.step", lambda: opt.step())
pr = cProfile.Profile(); pr.enable()
for _ in range(steps): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:170] for l in s.getvalue().splitlines()[:48]))
