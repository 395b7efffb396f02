"""Where the HOST time of an eager optimisation step goes (cProfile of 300 steps at the 70k config; the GPU needs 0.23 ms per step there,
the eager loop 0.45-0.58): python tools/profile_step_host.py [workload] [steps]"""
import cProfile, os, pstats, sys, time
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


for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
t_host = (time.perf_counter() - t0) / steps          # the loop without the final wait: how fast the host enqueues
torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / steps
print(f"{workload}: host enqueues a step in {t_host * 1e3:.3f} ms, a step takes {t_all * 1e3:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(steps): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
