#!/bin/bash
# round 4, second GPU pass: the leaf loop of the tier kernels by LDS-DMA + software pipeline (default build) against round 3's loop
# (tools/build/v_nodma) and the nt cache policy (tools/build/v_dma_nt): C-ABI driver, bitwise comparison of the solutions, per-launch
# tables, per-wave clock stamps, the GPU suite on the new kernels, the headline line
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_c; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
D=tools/build/nd_drive
run() { lib=$1; n=$2; shift 2; if [ "$lib" = default ]; then env "$@" timeout 300 $D $n 300 3 -1 0; else env LD_LIBRARY_PATH=$PWD/tools/build/$lib "$@" timeout 300 $D $n 300 3 -1 0; fi 2>&1 | grep -E "persist 0|hash|levels [0-9]|sum of|error|HIP" | sed "s/^/[$lib n=$n] /"; }
( for rep in 1 2 3; do for lib in default v_nodma; do run $lib 1000 X=1; done; done
  for lib in default v_nodma; do run $lib 1000 ND_DRIVE_TABLE=1; done
  for rep in 1 2; do for lib in default v_nodma; do run $lib 2000 X=1; done; done
  for rep in 1 2; do for lib in default v_nodma; do run $lib 500 X=1; run $lib 250 X=1; run $lib 100 X=1; done; done ) > $O/variants.txt 2>&1
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest.log 2>&1
( LARGESTEPS_HIP_LIB=tools/build/liblargesteps_hip_exp.so timeout 300 python tools/tier_stamps.py ) > $O/tier_stamps_dma.txt 2>&1
( timeout 600 python bench.py --steps 50 --warmup 3 --no-extra-baselines ) > $O/bench.json 2> $O/bench.err
for w in cfg3_dragon250k cfg2_bunny70k cfg5_plane4m; do ( timeout 400 python bench.py --steps 50 --warmup 3 --workload $w --no-extra-baselines --no-cpu-baseline ) > $O/bench_$w.json 2> $O/bench_$w.err; done
for w in cfg3_dragon250k cfg2_bunny70k cfg5_plane4m cfg4_plane1m; do ( LARGESTEPS_HIP_LIB=tools/build/v_nodma/liblargesteps_hip.so timeout 400 python bench.py --steps 50 --warmup 3 --workload $w --no-extra-baselines --no-cpu-baseline ) > $O/bench_nodma_$w.json 2> $O/bench_nodma_$w.err; done
cat $O/variants.txt | grep -v "levels"; tail -3 $O/pytest.log
for f in $O/bench*.json; do echo -n "$f: "; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print("ms", round(d["ms_per_step"], 4), "err", d["config"].get("max_abs_err_vs_v"), "frac", round(d["roofline"]["frac"], 4))
except Exception as e:
    print("failed:", e)
PY
done
grep -E "^\s+\[|leaf|phase" $O/tier_stamps_dma.txt | tail -30
