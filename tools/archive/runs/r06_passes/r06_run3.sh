set -x
mkdir -p gpurun_out/r3
export LARGESTEPS_HIP_LIB=$PWD/tools/build/v_stamps/liblargesteps_hip.so
python tools/tier_stamps.py cfg4_plane1m > gpurun_out/r3/stamps_plane16.txt 2>&1
python tools/tier_stamps.py cfg4b_sphere1m_uniform > gpurun_out/r3/stamps_sphere16.txt 2>&1
python tools/tier_stamps.py cfg4_plane1m 4 > gpurun_out/r3/stamps_plane4.txt 2>&1
cat gpurun_out/r3/stamps_plane16.txt gpurun_out/r3/stamps_sphere16.txt gpurun_out/r3/stamps_plane4.txt
