import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential, from_differential
dev = torch.device("cuda:0")
for name in ("cfg2_bunny70k", "cfg4_plane1m"):
    v, f, cfg = synthetic.config_mesh(name)
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
    u = to_differential(M, tv)
    ref = from_differential(M, u, "Cholesky").clone()
    side = torch.cuda.Stream()
    bad = 0
    for i in range(1500):
        if i % 3 == 2:
            with torch.cuda.stream(side):
                side.wait_stream(torch.cuda.current_stream())
                x = from_differential(M, u, "Cholesky")
            torch.cuda.current_stream().wait_stream(side)
        else:
            x = from_differential(M, u, "Cholesky")
        if i % 50 == 0 and not torch.equal(x, ref):
            bad += 1
    torch.cuda.synchronize()
    print(name, "1500 solves (every third on a side stream), mismatches at 30 checkpoints:", bad, "final equal:", bool(torch.equal(x, ref)))
