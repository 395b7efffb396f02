"""compute_matrix timing at the 1M-vertex config (uniform and cotangent): python tools/time_assembly.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
dev = torch.device("cuda:0")
v, f, cfg = synthetic.config_mesh("cfg4_plane1m")
tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
for cot in (False, True):
    for _ in range(3): M = compute_matrix(tv, tf, 50.0, cotan=cot)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): M = compute_matrix(tv, tf, 50.0, cotan=cot)
    torch.cuda.synchronize()
    print(f"compute_matrix(cotan={cot}) @1M: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms, nnz {M._nnz()}")
