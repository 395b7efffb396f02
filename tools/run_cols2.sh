#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "patch" 2>&1 | tail -3
timeout 600 python - <<'PY'
import sys, time, os
sys.path[:0] = [os.path.join(os.getcwd(), "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential
from largesteps.solvers import CholeskySolver
dev = torch.device("cuda:0")
v, f, cfg = synthetic.config_mesh("cfg4_plane1m")
tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
M = compute_matrix(tv, tf, cfg["lambda_"])
u = to_differential(M, tv)
for pc, env in ((3, None), (2, None), (1, None), (1, "4096,10,20000,4"), (2, "4096,10,10000,4"), (1, "4096,8,20000,4")):
    if env: os.environ["LARGESTEPS_PATCH"] = env
    else: os.environ.pop("LARGESTEPS_PATCH", None)
    s = CholeskySolver(M, patch_columns=pc)
    b = u[:, :pc].contiguous()
    for _ in range(5): x = s.solve(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): x = s.solve(b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    p = s.patch_plan
    print(f"pc={pc} env={env}: {dt*1e3:.3f} ms/solve depth={p.depth if p else None} max_local={p.max_local if p else None} rows={p.max_rows if p else None} err={float((x - tv[:, :pc]).abs().max()):.2e}", flush=True)
PY
