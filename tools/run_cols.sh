#!/bin/bash
# column-sharding checks on the 1-GPU box: HIP test, loopback bench smoke (both modes), single-GPU solve time per column count
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_distributed_gpu.py -m gpu -q -k "column" 2>&1 | tail -3
for mode in columns vertex; do
LS_DIST_LOOPBACK=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
   bench.py --gpus 2 --steps 5 --warmup 2 --shard $mode 2>&1 | tail -2
done
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
for env in (None, "4096,12,20000,4", "4096,16,20000,4", "8192,16,20000,4"):
    if env: os.environ["LARGESTEPS_PATCH"] = env
    s = CholeskySolver(M)
    for k in (1, 2, 3):
        if env and k != 1: continue
        b = u[:, :k].contiguous()
        try:
            for _ in range(5): x = s.solve(b)
        except Exception as e:
            print("env", env, "k", k, "failed:", e); continue
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): x = s.solve(b)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print(f"env={env} k={k}: {dt*1e3:.3f} ms/solve, info={s.last_info}, err={float((x - tv[:, :k]).abs().max()):.2e}", flush=True)
PY
