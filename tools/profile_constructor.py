"""the constructor of the default solver alone (for rocprofv3 --kernel-trace --stats): assemble, factorise, one solve.
   python tools/profile_constructor.py [workload] [repetitions]"""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.solvers import NestedDissectionSolver
w = sys.argv[1] if len(sys.argv) > 1 else "cfg4_plane1m"
dev = torch.device("cuda:0")
v, f, cfg = synthetic.config_mesh(w)
tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
torch.cuda.synchronize()
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # > 1: the first construction of a process pays one-off costs
for r in range(reps):
    s = NestedDissectionSolver(M)
    x = s.solve(tv.contiguous())
    torch.cuda.synchronize()
    print(w, f"constructor (run {r})", round(s.build_seconds, 3), "s", {k: round(t, 3) for k, t in s.timings.items()})
    del s
