"""to_differential's kernel (ls_spmv, csrc/spmv.hip) at the 1M-vertex config, HIP events: back to back (warm: matrix and vectors in the
Infinity Cache), and "cold" -- 600 MB of other data streamed through the caches before every timed call, which is what the one call per
(re)mesh of the reference's loop meets (parameterize.py:19-30 after compute_matrix).   python tools/time_spmv.py [workload...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic, _native
from largesteps.geometry import compute_matrix
dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["cfg4_plane1m", "cfg3_dragon250k"]:
    v, f, cfg = synthetic.config_mesh(name)
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
    csr = _native.csr_of(M)
    V, nnz = csr.V, csr.nnz
    bts = 8 * nnz + 4 * (V + 1) + 2 * 12 * V
    flush = torch.empty(150_000_000, dtype=torch.float32, device=dev)
    for variant in (0, 1):
        for _ in range(5): y = _native.spmv(csr, tv, variant)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(50): y = _native.spmv(csr, tv, variant)
        ev[1].record(); torch.cuda.synchronize()
        warm = ev[0].elapsed_time(ev[1]) / 50 * 1e3
        cold = []
        for _ in range(10):
            flush.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); y = _native.spmv(csr, tv, variant); e1.record(); torch.cuda.synchronize()
            cold.append(e0.elapsed_time(e1) * 1e3)
        cold.sort()
        print(f"{name} ls_spmv variant {variant}: warm {warm:.1f} us = {bts / warm * 1e-6:.2f} TB/s ({bts / warm * 1e-6 / 8:.2f} of 8), "
              f"cold median {cold[len(cold) // 2]:.1f} us = {bts / cold[len(cold) // 2] * 1e-6:.2f} TB/s; algorithmic {bts / 1e6:.1f} MB", flush=True)
