#!/bin/bash
# arity-8 trees: slots named by the mask only (default) against all eight slots (tools/build/v_base)
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_v; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do for w in cfg3_dragon250k cfg2_bunny70k; do for lib in default v_base; do if [ $lib = default ]; then L=""; else L="LARGESTEPS_HIP_LIB=tools/build/v_base/liblargesteps_hip.so"; fi; env $L timeout 300 python bench.py --steps 100 --warmup 5 --workload $w --no-extra-baselines --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w','$lib',round(d['ms_per_step'],4), d['config'].get('max_abs_err_vs_v'), [round(l['us'],1) for l in d['config']['launches']])"; done; done; done | tee $O/bench.txt
