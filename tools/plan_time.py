"""symbolic analysis only (host threads, no GPU): LS_PLAN_TIMING=1 LS_PLAN_THREADS=N python tools/plan_time.py [config]"""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(_R, "tests"), os.path.join(_R, "large-steps-pytorch_amd"), _R]
import numpy as np, scipy.sparse as sp
from largesteps import synthetic
from native_plan import native_plan
v, f, cfg = synthetic.config_mesh(sys.argv[1] if len(sys.argv) > 1 else "cfg4_plane1m")
V = v.shape[0]
e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
A = sp.coo_matrix((np.ones(len(e) * 2), (np.r_[e[:, 0], e[:, 1]], np.r_[e[:, 1], e[:, 0]])), shape=(V, V)).tocsr()
A = (A + sp.identity(V)).tocsr(); A.sort_indices()
for _ in range(2):
    t = time.perf_counter(); p = native_plan(A.indptr, A.indices, v, 64, 4); print("total incl. python wrap", time.perf_counter() - t, "plan seconds", p.seconds)
