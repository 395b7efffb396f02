set -x
mkdir -p gpurun_out/r13
O=gpurun_out/r13
for i in 1 2 3 4; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-baselines 2>/dev/null | cut -c1-160; done > $O/bench_repeat.txt
python tools/irregular_1m.py 300 --quick 2>&1 | grep plane >> $O/bench_repeat.txt
cat $O/bench_repeat.txt
