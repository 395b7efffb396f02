"""B independent small meshes: one block-diagonal system through one set of launches (largesteps.batched) against a loop over
the meshes.      python tools/bench_batched.py [n_meshes] [icosphere frequency] [solves]"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic
from largesteps.batched import MeshBatch, compute_matrix_batched
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential, from_differential
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda:0")
v0, f0 = synthetic.icosphere(n)
vs = [torch.from_numpy(synthetic.perturb(v0, radial=0.03, seed=i)).to(dev) for i in range(B)]
fs = [torch.from_numpy(f0).to(dev) for _ in range(B)]
t0 = time.perf_counter()
batch = MeshBatch(vs, fs)
M = compute_matrix_batched(batch, 19.0)
u = to_differential(M, batch.verts)
x = from_differential(M, u, "Cholesky")
torch.cuda.synchronize(); t_build = time.perf_counter() - t0
for _ in range(3): from_differential(M, u, "Cholesky")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps): x = from_differential(M, u, "Cholesky")
torch.cuda.synchronize(); t_b = (time.perf_counter() - t0) / reps
err = float((x - batch.verts).abs().max())
t0 = time.perf_counter()
Ms = [compute_matrix(v, f, 19.0) for v, f in zip(vs, fs)]
us = [to_differential(Mi, v) for Mi, v in zip(Ms, vs)]
for Mi, ui in zip(Ms, us): from_differential(Mi, ui, "Cholesky")
torch.cuda.synchronize(); t_build_loop = time.perf_counter() - t0
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(max(1, reps // 10)):
    for Mi, ui in zip(Ms, us): from_differential(Mi, ui, "Cholesky")
torch.cuda.synchronize(); t_l = (time.perf_counter() - t0) / max(1, reps // 10)
print(f"{B} meshes x {v0.shape[0]} vertices: batched {t_b * 1e3:.3f} ms per solve of ALL meshes ({t_b / B * 1e6:.1f} us per mesh, err {err:.1e}), "
      f"loop {t_l * 1e3:.3f} ms ({t_l / B * 1e6:.1f} us per mesh) -> {t_l / t_b:.1f}x; setup (assemble + factorise + first solve): batched {t_build:.2f} s, loop {t_build_loop:.2f} s")
