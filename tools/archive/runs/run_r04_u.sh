#!/bin/bash
# the up sweep on the down sweep's copy of W (k_nd_up_t, default) against LS_ND_UP_T=0 (k_nd_up_b on its own copy)
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_u; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
for rep in 1 2 3; do for T in 1 0; do for n in 1000; do echo -n "LS_ND_UP_T=$T n=$n: "; LS_ND_UP_T=$T timeout 300 $D $n 300 3 -1 0 2>&1 | grep -E "persist 0" | cut -c1-125; done; done; done > $O/up_t.txt 2>&1
for T in 1 0; do for n in 2000 700 500 350 250 150 100 40; do echo -n "LS_ND_UP_T=$T n=$n: "; LS_ND_UP_T=$T timeout 300 $D $n 200 3 -1 0 2>&1 | grep -E "persist 0" | cut -c1-125; done; done >> $O/up_t.txt 2>&1
for T in 1 0; do echo "== LS_ND_UP_T=$T"; LS_ND_UP_T=$T ND_DRIVE_TABLE=1 timeout 300 $D 1000 300 3 -1 0 2>&1 | grep -E "levels [0-9]|hash"; done >> $O/up_t.txt 2>&1
cat $O/up_t.txt
for w in cfg3_dragon250k cfg2_bunny70k; do for T in 1 0; do LS_ND_UP_T=$T timeout 300 python bench.py --steps 100 --warmup 5 --workload $w --no-extra-baselines --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w','LS_ND_UP_T=$T',round(d['ms_per_step'],4), d['config'].get('max_abs_err_vs_v'))"; done; done | tee $O/bench.txt
