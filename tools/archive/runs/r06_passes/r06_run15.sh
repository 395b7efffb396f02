set -x
mkdir -p gpurun_out/r15
O=gpurun_out/r15
python tools/profile_step_host.py cfg2_bunny70k > $O/step_host_70k.txt 2>&1
for i in 1 2 3; do python tools/bench_step.py cfg2_bunny70k 100 2>&1 | grep "^cfg"; done > $O/step_repeat.txt
python tools/bench_step.py cfg3_dragon250k 100 2>&1 | grep "^cfg" >> $O/step_repeat.txt
head -40 $O/step_host_70k.txt; cat $O/step_repeat.txt
