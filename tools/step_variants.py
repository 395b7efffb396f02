"""eager step, host-bound sizes: does keeping the previous loss alive (as the reference's loop does) cost time? python tools/step_variants.py [workload]"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential, from_differential
from largesteps.normals import compute_face_normals, compute_vertex_normals
from largesteps.optimize import AdamUniform
workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2_bunny70k"
dev = torch.device("cuda:0")
v, f, cfg = synthetic.config_mesh(workload)
tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
target_n = compute_vertex_normals(tv, tf, compute_face_normals(tv, tf)).detach()
target_v = tv + 0.01 * torch.randn_like(tv)
u = to_differential(M, tv).clone().requires_grad_(True)
opt = AdamUniform([u], 3e-2)


def step(u, opt):
    x = from_differential(M, u, "Cholesky")
    n = compute_vertex_normals(x, tf, compute_face_normals(x, tf))
    loss = (x - target_v).square().mean() + (n - target_n).square().mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss


def timed(name, body, n=200):
    for _ in range(20): body()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): body()
    th = (time.perf_counter() - t0) / n
    torch.cuda.synchronize(); ta = (time.perf_counter() - t0) / n
    print(f"{workload}: {name:70s} host {th*1e3:.3f} ms  with the device {ta*1e3:.3f} ms", flush=True)


keep = [None]
def a(): keep[0] = step(u, opt)
def b(): step(u, opt)
def c(): keep[0] = step(u, opt).detach()
for rep in range(2):
    timed("loss kept until the next step returns (l = step())", a)
    timed("loss dropped at once (step())", b)
    timed("detached loss kept", c)
