set -x
mkdir -p gpurun_out/r19
O=gpurun_out/r19
python tools/irregular_1m.py 300 --quick > $O/quick.txt 2>&1
LARGESTEPS_HIP_LIB=$PWD/tools/build/v_stamps/liblargesteps_hip.so python tools/tier_stamps.py cfg4_plane1m > $O/stamps_plane16.txt 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_nested_gpu.py -m gpu -x -q -k "irregular or sixteen_wave or one_million or cfg1 or cfg4 or columns or nested or direct" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-baselines 2>/dev/null | cut -c1-170; done > $O/driver_style_repeats.txt
tail -3 $O/pytest.log; grep -v amdgpu $O/quick.txt; grep -v "leaf [0-9] done\|amdgpu" $O/stamps_plane16.txt | head -8; cut -c100-170 $O/driver_style_repeats.txt
