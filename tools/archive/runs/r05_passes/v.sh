#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r05_v; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "normal or captured or optimisation_step" 2>&1 | tail -3 | tee $O/pytest_normals.txt
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/bitwise.txt
import os, sys
R = os.environ["GRAFT_REPO_ROOT"]; sys.path[:0] = [R, os.path.join(R, "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic, normals
dev = torch.device("cuda:0")
for name in ("cfg4_plane1m", "cfg3_dragon250k", "cfg2_bunny70k"):
    v, f, _ = synthetic.config_mesh(name)
    tf = torch.from_numpy(f).to(dev)
    w = torch.randn(v.shape[0], 3, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    w2 = torch.randn(3, f.shape[0], device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    res = []
    for vm in (False, True):
        normals._VERTEX_MAJOR = vm
        tv = torch.from_numpy(v).to(dev).requires_grad_(True)
        fn = normals.compute_face_normals(tv, tf); vn = normals.compute_vertex_normals(tv, tf, fn)
        g, = torch.autograd.grad((vn * w).sum() + (fn * w2).sum(), tv)
        res.append((vn.detach().clone(), g.clone()))
    print(name, "vertex normals bitwise equal:", bool(torch.equal(res[0][0], res[1][0])) or bool(((res[0][0] == res[1][0]) | (res[0][0].isnan() & res[1][0].isnan())).all()),
          "gradient bitwise equal:", bool(torch.equal(res[0][1], res[1][1])) or bool(((res[0][1] == res[1][1]) | (res[0][1].isnan() & res[1][1].isnan())).all()),
          "max |dg|", float((res[0][1] - res[1][1]).abs().nan_to_num().max()))
PY
for vm in 1 0; do
  ( cd /tmp && export TMPDIR=/tmp && LS_NORMALS_VERTEX_MAJOR=$vm timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof$vm -o n -- python $R/tools/bench_normals.py ) > $O/normals$vm.log 2>&1
  echo "== vertex-major $vm" | tee -a $O/kernels.txt
  grep "^normals" $O/normals$vm.log | tee -a $O/kernels.txt
  python - $(find $O/prof$vm -name "*kernel_stats.csv" | head -1) <<'PY' | tee -a $O/kernels.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'ls::' in r['Name'] and int(r['Calls']) > 20: print("   ", r['Name'].replace('void ls::','')[:44], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
  rm -rf $O/prof$vm
done
