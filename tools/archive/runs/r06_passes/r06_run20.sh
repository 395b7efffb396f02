mkdir -p gpurun_out/r20
O=gpurun_out/r20
for v in e4 e3 base; do
  if [ $v = base ]; then unset LARGESTEPS_HIP_LIB; else export LARGESTEPS_HIP_LIB=$PWD/tools/build/v_$v/liblargesteps_hip.so; fi
  echo "== variant $v"; python tools/irregular_1m.py 300 --quick --table 2>&1 | grep -v amdgpu
done > $O/nd_e.txt
cat $O/nd_e.txt
