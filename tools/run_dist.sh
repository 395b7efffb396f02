#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_distributed_gpu.py -m gpu -q --timeout 600 2>&1 | tail -30 ) > gpurun_out/pytest_dist.log 2>&1
tail -5 gpurun_out/pytest_dist.log
for N in 2 4; do
  ( LS_DIST_LOOPBACK=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $N --steps 3 --warmup 1 ) > gpurun_out/bench_loopback_$N.json 2> gpurun_out/bench_loopback_$N.err
  tail -c 1500 gpurun_out/bench_loopback_$N.json; echo; tail -5 gpurun_out/bench_loopback_$N.err
done
( LS_DIST_LOOPBACK=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 3 --warmup 1 --shard vertex ) > gpurun_out/bench_loopback_vertex2.json 2> gpurun_out/bench_loopback_vertex2.err
tail -c 900 gpurun_out/bench_loopback_vertex2.json; echo; tail -3 gpurun_out/bench_loopback_vertex2.err
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err
tail -c 600 gpurun_out/bench_torchrun1.json; tail -3 gpurun_out/bench_torchrun1.err
