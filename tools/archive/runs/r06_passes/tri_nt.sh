mkdir -p gpurun_out/trint
for v in trint base trint base; do
  if [ $v = base ]; then unset LARGESTEPS_HIP_LIB; else export LARGESTEPS_HIP_LIB=$PWD/tools/build/v_$v/liblargesteps_hip.so; fi
  echo "== variant $v"; python tools/irregular_1m.py 300 --quick 2>&1 | grep -v amdgpu | head -3
done > gpurun_out/trint/out.txt
cat gpurun_out/trint/out.txt
