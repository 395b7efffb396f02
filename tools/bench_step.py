"""One optimisation step of the path, as the reference's loop runs it (scripts/main.py:170-200):
   v = from_differential(M, u) -> face / vertex normals -> loss -> backward (normals, then the adjoint solve) -> AdamUniform,
   eager and as ONE captured graph (torch.cuda.graph: every launch of the step recorded once, replayed per step).
   python tools/bench_step.py [workload] [steps]"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential, from_differential
from largesteps.normals import compute_face_normals, compute_vertex_normals
from largesteps.optimize import AdamUniform
workload = sys.argv[1] if len(sys.argv) > 1 else "cfg4_plane1m"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
v, f, cfg = synthetic.config_mesh(workload)
tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
target_n = compute_vertex_normals(tv, tf, compute_face_normals(tv, tf)).detach()
target_v = tv + 0.01 * torch.randn_like(tv)


def make(capturable):
    u = to_differential(M, tv).clone().requires_grad_(True)
    return u, AdamUniform([u], 3e-2, capturable=capturable)


def step(u, opt):
    x = from_differential(M, u, "Cholesky")
    n = compute_vertex_normals(x, tf, compute_face_normals(x, tf))
    loss = (x - target_v).square().mean() + (n - target_n).square().mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss


u, opt = make(False)
for _ in range(10): step(u, opt)
# three timed blocks: the eager step at the reference's mesh sizes is HOST-bound, and the host's first hundreds of steps are slower than its
# steady state (allocator, autograd engine, clocks): min / median / max over the blocks, the median is the figure quoted
blocks = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): l = step(u, opt)
    torch.cuda.synchronize(); blocks.append((time.perf_counter() - t0) / steps)
dt = sorted(blocks)[1]
print(f"{workload}: {dt*1e3:.3f} ms per optimisation step, eager (forward solve + normals + loss + backward incl. adjoint solve + AdamUniform), loss {float(l):.3e}"
      f"  [blocks of {steps} steps: {' '.join(f'{b*1e3:.3f}' for b in blocks)}]")

from largesteps.capture import CapturedStep
u, opt = make(True)
cs = CapturedStep(lambda: step(u, opt), warmup=3)            # largesteps/capture.py: warm-up on a side stream, capture, replay
for _ in range(3): loss = cs()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): loss = cs()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"{workload}: {dt*1e3:.3f} ms per optimisation step, captured graph replay (largesteps.capture.CapturedStep), loss {float(loss):.3e}")
