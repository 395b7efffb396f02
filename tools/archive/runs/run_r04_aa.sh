#!/bin/bash
# whole rows requested at once at the top of the tree (default: levels of <= 320 tiles) against LS_ND_DEEP=0
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_aa; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
for rep in 1 2 3 4; do for T in 320 0; do echo -n "LS_ND_DEEP=$T n=1000: "; LS_ND_DEEP=$T timeout 300 $D 1000 300 3 -1 0 2>&1 | grep -E "persist 0" | cut -c1-70; done; done > $O/deep.txt 2>&1
for T in 320 0 600; do for n in 2000 700 500 250 100; do echo -n "LS_ND_DEEP=$T n=$n: "; LS_ND_DEEP=$T timeout 300 $D $n 200 3 -1 0 2>&1 | grep -E "persist 0" | cut -c1-70; done; done >> $O/deep.txt 2>&1
for T in 320 0; do echo "== LS_ND_DEEP=$T"; LS_ND_DEEP=$T ND_DRIVE_TABLE=1 timeout 300 $D 1000 300 3 -1 0 2>&1 | grep -E "levels [0-9]|hash"; done >> $O/deep.txt 2>&1
cat $O/deep.txt
for w in cfg3_dragon250k cfg2_bunny70k; do for T in 320 0; do LS_ND_DEEP=$T timeout 300 python bench.py --steps 100 --warmup 5 --workload $w --no-extra-baselines --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w','LS_ND_DEEP=$T',round(d['ms_per_step'],4), d['config'].get('max_abs_err_vs_v'), [round(l['us'],1) for l in d['config']['launches']])"; done; done | tee $O/bench.txt
