#!/bin/bash
# full GPU pass (round 3): parity tests, smoke, bench (direct solver; other configs; persistent upper levels A/B), rocprofv3 kernel stats +
# per-level trace + FETCH_SIZE / WRITE_SIZE passes of the bench command, constructor profile, remesh / step / batched tools
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/full_r03; rm -rf $O; mkdir -p $O/pmc
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $O/pytest.log 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( timeout 600 python bench.py --steps 50 --warmup 3 ) > $O/bench.json 2> $O/bench.err
( LS_ND_PERSIST=1 timeout 400 python bench.py --steps 50 --warmup 3 --no-cpu-baseline ) > $O/bench_persistent.json 2> $O/bench_persistent.err
for w in cfg2_bunny70k cfg3_dragon250k cfg5_plane4m; do ( timeout 400 python bench.py --steps 50 --warmup 3 --workload $w --no-extra-baselines $( [ $w = cfg5_plane4m ] && echo --no-cpu-baseline ) ) > $O/bench_$w.json 2> $O/bench_$w.err; done
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline ) > $O/rocprof.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc/bench_$C -o out -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > $O/pmc/bench_$C.log 2>&1
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_ctor -o ctor -- python $GRAFT_REPO_ROOT/tools/profile_constructor.py cfg4_plane1m ) > $O/rocprof_ctor.log 2>&1
cp $(find $O/prof_ctor -name "*kernel_stats.csv" | head -1) $O/constructor_kernel_stats.csv; rm -rf $O/prof_ctor
for w in cfg4_plane1m cfg3_dragon250k cfg2_bunny70k; do timeout 300 python tools/profile_constructor.py $w 3 2>&1 | grep constructor; done > $O/constructor_times.txt
for w in cfg4_plane1m cfg3_dragon250k cfg2_bunny70k; do timeout 600 python tools/bench_remesh.py $w 100 3 2>&1 | grep -v amdgpu.ids; done > $O/remesh.txt
( timeout 600 python tools/bench_step.py cfg4_plane1m 30; timeout 600 python tools/bench_step.py cfg3_dragon250k 30; timeout 600 python tools/bench_step.py cfg2_bunny70k 30 ) 2>&1 | grep "^cfg" > $O/step.txt
for N in 2 4 8; do ( LS_DIST_LOOPBACK=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N bench.py --gpus $N --steps 3 --warmup 1 ) > $O/bench_loopback_$N.json 2> $O/bench_loopback_$N.err; done
python tools/pmc_summary.py $O/pmc/bench_FETCH_SIZE $O/pmc/bench_WRITE_SIZE cfg4_plane1m $O/pmc_traffic.json > $O/pmc_summary.log 2>&1
python tools/nd_trace.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/nd_levels.txt 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/prof; find $O/pmc -name "*.csv" -size +1M -delete
tail -4 $O/pytest.log; tail -2 $O/smoke.log; cut -c1-700 $O/bench.json; echo; cut -c1-300 $O/bench_persistent.json; echo; for w in cfg2_bunny70k cfg3_dragon250k cfg5_plane4m; do cut -c1-300 $O/bench_$w.json; echo; done
tail -3 $O/bench.err
head -14 $O/kernel_stats.csv | cut -c1-160; cat $O/constructor_times.txt $O/remesh.txt $O/step.txt; for N in 2 4 8; do tail -c 700 $O/bench_loopback_$N.json | head -c 400; echo; done; cat $O/pmc_summary.log | grep k_nd; cat $O/nd_levels.txt
