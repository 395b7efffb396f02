#!/bin/bash
# round 4, GPU pass H: two-buffer ring in the lanes-along-the-reduction level kernels (default) against the same build without it (v_noring)
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_h; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
D=tools/build/nd_drive
run() { lib=$1; n=$2; shift 2; if [ "$lib" = default ]; then env "$@" timeout 300 $D $n 300 3 -1 0; else env LD_LIBRARY_PATH=$PWD/tools/build/$lib "$@" timeout 300 $D $n 300 3 -1 0; fi 2>&1 | grep -E "persist 0|hash|levels [0-9]|sum of|error|HIP" | sed "s/^/[$lib n=$n] /"; }
( for rep in 1 2 3 4; do for lib in default v_noring; do run $lib 1000 X=1; done; done; for lib in default v_noring; do run $lib 1000 ND_DRIVE_TABLE=1; done
  for rep in 1 2; do for lib in default v_noring; do run $lib 2000 X=1; done; done
  for rep in 1 2; do for lib in default v_noring; do run $lib 500 X=1; run $lib 250 X=1; run $lib 100 X=1; run $lib 40 X=1; done; done ) > $O/variants.txt 2>&1
grep -v levels $O/variants.txt
if grep -q "\[default n=1000\] solution hash 4aa54694b155829f" $O/variants.txt; then
  ( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest.log 2>&1
  tail -3 $O/pytest.log
  ( timeout 600 python bench.py --steps 50 --warmup 3 --no-extra-baselines --no-cpu-baseline ) > $O/bench.json 2> $O/bench.err
  for w in cfg3_dragon250k cfg2_bunny70k cfg5_plane4m; do ( timeout 400 python bench.py --steps 50 --warmup 3 --workload $w --no-extra-baselines --no-cpu-baseline ) > $O/bench_$w.json 2> $O/bench_$w.err; done
  for w in cfg4_plane1m cfg3_dragon250k cfg2_bunny70k cfg5_plane4m; do ( LARGESTEPS_HIP_LIB=tools/build/v_noring/liblargesteps_hip.so timeout 400 python bench.py --steps 50 --warmup 3 --workload $w --no-extra-baselines --no-cpu-baseline ) > $O/bench_noring_$w.json 2> $O/bench_noring_$w.err; done
  for f in $O/bench*.json; do echo -n "$f: "; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print("ms", round(d["ms_per_step"], 4), "err", d["config"].get("max_abs_err_vs_v"), "frac", round(d["roofline"]["frac"], 4))
except Exception as e:
    print("failed:", e)
PY
  done
else
  echo "DEFAULT BUILD GIVES A DIFFERENT SOLUTION: suite skipped"
fi
( LARGESTEPS_HIP_LIB=tools/build/liblargesteps_hip_exp.so timeout 300 python tools/tier_stamps.py ) > $O/tier_stamps.txt 2>&1
grep -E "^\s+\[|phase" $O/tier_stamps.txt
