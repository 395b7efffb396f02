#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python - <<'PY'
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "large-steps-pytorch_amd")]
import numpy as np, torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.solvers import NestedDissectionSolver
from oracle import solve as osv
dev = torch.device("cuda:0")
for name, (v, f), kw in [("plane40", synthetic.plane(40), dict(lambda_=30.0)), ("ico40", synthetic.icosphere(40), dict(lambda_=19.0)),
                         ("ico20cot", (synthetic.perturb(synthetic.icosphere(20)[0], radial=0.05, tangential=0.2, edge=0.1, seed=5), synthetic.icosphere(20)[1]), dict(lambda_=0.0, alpha=0.9, cotan=True))]:
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, **kw)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    for k in (1, 3, 4, 6):
        b = np.random.default_rng(k).standard_normal((v.shape[0], k)).astype(np.float32)
        x64 = osv.from_differential(idx[0], idx[1], val, b)
        s = NestedDissectionSolver(M)
        x = s.solve(torch.from_numpy(b).to(dev)); x2 = s.solve(torch.from_numpy(b).to(dev))
        print(name, "k", k, "rel err", float(np.abs(x.cpu().numpy() - x64).max() / np.abs(x64).max()), "deterministic", bool(torch.equal(x, x2)), flush=True)
PY
for leaf in 96 48; do python tools/nd_prof.py cfg4_plane1m $leaf 50 | head -1; done
python tools/nd_prof.py cfg3_dragon250k 96 50 | head -1
mkdir -p gpurun_out/ndprof; rm -rf gpurun_out/ndprof/*
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ndprof -- python tools/nd_prof.py cfg4_plane1m 96 5 > /dev/null 2>&1
python tools/nd_trace.py $(find gpurun_out/ndprof -name "*kernel_trace.csv" | head -1)
