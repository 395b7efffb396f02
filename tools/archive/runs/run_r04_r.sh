#!/bin/bash
# A/B through the C-ABI driver: default library against tools/build/v_base (the same tree without the change under test)
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_r; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
run() { lib=$1; n=$2; shift 2; if [ "$lib" = default ]; then env "$@" timeout 300 $D $n 300 3 -1 0; else env LD_LIBRARY_PATH=$PWD/tools/build/$lib "$@" timeout 300 $D $n 300 3 -1 0; fi 2>&1 | grep -E "persist 0|hash|levels [0-9]|sum of|error|HIP" | sed "s/^/[$lib n=$n] /"; }
( for rep in 1 2 3 4; do for lib in default v_base; do run $lib 1000 X=1; done; done
  for lib in default v_base; do run $lib 1000 ND_DRIVE_TABLE=1; done
  for rep in 1 2; do for lib in default v_base; do run $lib 2000 X=1; run $lib 500 X=1; run $lib 250 X=1; done; done ) > $O/ab.txt 2>&1
grep -v "levels" $O/ab.txt | grep -E "persist|hash" | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9}' | sort | uniq -c | sort -k2,3 | head -60
grep "levels" $O/ab.txt
for w in cfg3_dragon250k cfg2_bunny70k; do for lib in default v_base; do if [ $lib = default ]; then L=""; else L="LARGESTEPS_HIP_LIB=tools/build/v_base/liblargesteps_hip.so"; fi; env $L timeout 300 python bench.py --steps 100 --warmup 5 --workload $w --no-extra-baselines --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w','$lib',round(d['ms_per_step'],4))"; done; done
