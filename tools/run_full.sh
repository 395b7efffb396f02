#!/bin/bash
# The full GPU pass behind the numbers DESIGN.md / docs/measurements.md / BASELINE.md quote:  gpurun --timeout 3000 -- 'bash tools/run_full.sh r05_run1'
# Everything lands in gpurun_out/<tag>/; copy what is to be cited into profiles/ as <tag>_*.
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-full}; O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $O/pytest.log 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( timeout 600 python bench.py ) > $O/bench_driver_style.json 2> $O/bench_driver_style.err
( timeout 600 python bench.py --steps 50 --warmup 3 ) > $O/bench.json 2> $O/bench.err
for w in cfg2_bunny70k cfg3_dragon250k cfg5_plane4m scroll250k cfg4b_sphere1m cfg4b_sphere1m_uniform scroll1m folded1m; do ( timeout 400 python bench.py --steps 50 --warmup 3 --workload $w --no-extra-baselines $( [ $w = cfg5_plane4m ] && echo --no-cpu-baseline ) ) > $O/bench_$w.json 2> $O/bench_$w.err; done
( timeout 600 python bench.py --workload cfg4b_sphere1m ) > $O/bench_driver_style_sphere.json 2> $O/bench_driver_style_sphere.err
for i in 1 2 3 4 5 6; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-baselines 2>/dev/null | cut -c1-170; done > $O/driver_style_repeats.txt
( timeout 900 python tools/irregular_1m.py 300 --table; timeout 900 python tools/irregular_1m.py 300 --sizes ) 2>&1 | grep -v amdgpu.ids > $O/tier16_irregular.txt
if [ -f tools/build/v_stamps/liblargesteps_hip.so ]; then for w in cfg4_plane1m cfg4b_sphere1m_uniform; do LARGESTEPS_HIP_LIB=$R/tools/build/v_stamps/liblargesteps_hip.so timeout 300 python tools/tier_stamps.py $w 2>&1 | grep -v amdgpu.ids; done > $O/tier_stamps.txt; fi
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_sphere -o bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra-baselines --workload cfg4b_sphere1m ) > $O/rocprof_sphere.log 2>&1
python tools/nd_trace.py $(find $O/prof_sphere -name "*kernel_trace.csv" | head -1) > $O/nd_levels_sphere.txt 2>&1; cp $(find $O/prof_sphere -name "*kernel_stats.csv" | head -1) $O/kernel_stats_sphere.csv; rm -rf $O/prof_sphere
for w in cfg4_plane1m cfg4b_sphere1m scroll1m cfg3_dragon250k; do timeout 600 python tools/ctor_in_loop.py $w 8 2>&1 | grep -v amdgpu.ids; done > $O/ctor_in_loop.txt
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline ) > $O/rocprof.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$O/pmc/bench_$C -o out -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > $O/pmc/bench_$C.log 2>&1
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_ctor -o ctor -- python $R/tools/profile_constructor.py cfg4_plane1m ) > $O/rocprof_ctor.log 2>&1
cp $(find $O/prof_ctor -name "*kernel_stats.csv" | head -1) $O/constructor_kernel_stats.csv; rm -rf $O/prof_ctor
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_asm -o asm -- python $R/tools/time_assembly.py ) > $O/time_assembly.txt 2>&1
cp $(find $O/prof_asm -name "*kernel_stats.csv" | head -1) $O/assembly_kernel_stats.csv; rm -rf $O/prof_asm
timeout 300 python tools/time_spmv.py 2>&1 | grep -v amdgpu > $O/spmv.txt
for w in cfg4_plane1m cfg4b_sphere1m cfg5_plane4m cfg3_dragon250k cfg2_bunny70k scroll250k; do LS_PLAN_TIMING=1 timeout 300 python tools/profile_constructor.py $w 3 2>&1 | grep -E "constructor|nd_plan|ls_direct_factor"; done > $O/constructor_times.txt
for w in cfg4_plane1m cfg4b_sphere1m cfg3_dragon250k cfg2_bunny70k; do timeout 600 python tools/bench_remesh.py $w 100 8 2>&1 | grep -v amdgpu.ids; done > $O/remesh.txt
( timeout 600 python tools/bench_step.py cfg4_plane1m 30; timeout 600 python tools/bench_step.py cfg3_dragon250k 30; timeout 600 python tools/bench_step.py cfg2_bunny70k 30 ) 2>&1 | grep "^cfg" > $O/step.txt
( timeout 600 python tools/shard_rank_time.py cfg4_plane1m 200; timeout 900 python tools/shard_rank_time.py cfg5_plane4m 100 ) 2>&1 | grep "^cfg" > $O/shard_rank_kernel_times.txt
for N in 2 4 8; do ( LS_DIST_LOOPBACK=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N bench.py --gpus $N --steps 3 --warmup 1 ) > $O/bench_loopback_$N.json 2> $O/bench_loopback_$N.err; done
( timeout 600 python bench.py --gpus 8 --steps 3 --warmup 1 ) > $O/bench_loopback_8_plain_start.json 2> $O/bench_loopback_8_plain_start.err
python tools/pmc_summary.py $O/pmc/bench_FETCH_SIZE $O/pmc/bench_WRITE_SIZE cfg4_plane1m $O/pmc_traffic.json > $O/pmc_summary.log 2>&1
python tools/nd_trace.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/nd_levels.txt 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/prof; find $O/pmc -name "*.csv" -size +1M -delete
tail -4 $O/pytest.log; tail -2 $O/smoke.log; cut -c1-500 $O/bench_driver_style.json; echo; echo
for w in cfg2_bunny70k cfg3_dragon250k cfg5_plane4m scroll250k; do cut -c1-200 $O/bench_$w.json; echo; done
head -14 $O/kernel_stats.csv | cut -c1-160; cat $O/remesh.txt $O/step.txt; grep k_nd $O/pmc_summary.log; cat $O/nd_levels.txt
