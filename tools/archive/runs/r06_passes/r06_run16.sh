mkdir -p gpurun_out/r16
for w in cfg2_bunny70k cfg3_dragon250k cfg4_plane1m; do python tools/bench_step.py $w 100 2>&1 | grep "^cfg"; done > gpurun_out/r16/step.txt
cat gpurun_out/r16/step.txt
