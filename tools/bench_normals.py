"""time the normals forward + backward at the 1M-vertex config: python tools/bench_normals.py"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic
from largesteps.normals import compute_face_normals, compute_vertex_normals
dev = torch.device("cuda:0")
v, f, _ = synthetic.config_mesh("cfg4_plane1m")
tv, tf = torch.from_numpy(v).to(dev).requires_grad_(True), torch.from_numpy(f).to(dev)
w = torch.randn_like(tv)
def step():
    fn = compute_face_normals(tv, tf); vn = compute_vertex_normals(tv, tf, fn)
    g, = torch.autograd.grad((vn * w).sum(), tv)
    return vn, g
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): fn = compute_face_normals(tv, tf); vn = compute_vertex_normals(tv, tf, fn)
torch.cuda.synchronize(); t_fwd = (time.perf_counter() - t0) / 20
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / 20
V, F = v.shape[0], f.shape[0]
print(f"normals @ V={V} F={F}: forward {t_fwd*1e3:.3f} ms, forward+backward {t_all*1e3:.3f} ms")
# the same ops through stock torch (the reference's formulation) for comparison
def ref_forward(verts, faces):
    fi = faces.t().long(); vt = verts.t()
    vv = [vt.index_select(1, fi[0]), vt.index_select(1, fi[1]), vt.index_select(1, fi[2])]
    c = torch.cross(vv[1] - vv[0], vv[2] - vv[0], dim=0); n = c / torch.norm(c, dim=0)
    normals = torch.zeros_like(vt)
    for i in range(3):
        d0 = vv[(i + 1) % 3] - vv[i]; d0 = d0 / torch.norm(d0)
        d1 = vv[(i + 2) % 3] - vv[i]; d1 = d1 / torch.norm(d1)
        ang = torch.acos(torch.sum(d0 * d1, 0).clamp(-1, 1))
        nn = n * ang
        for j in range(3): normals[j].index_add_(0, fi[i], nn[j])
    return (normals / torch.norm(normals, dim=0)).t()
for _ in range(3):
    g, = torch.autograd.grad((ref_forward(tv, tf) * w).sum(), tv)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    g, = torch.autograd.grad((ref_forward(tv, tf) * w).sum(), tv)
torch.cuda.synchronize(); t_ref = (time.perf_counter() - t0) / 10
print(f"stock torch formulation (same math, ~40 kernels + autograd): forward+backward {t_ref*1e3:.3f} ms -> x{t_ref / t_all:.1f}")
vn_t = ref_forward(tv, tf)
print("max |hip - torch| vertex normals:", float((vn_t - vn).abs().max()))
