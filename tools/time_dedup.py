"""remove_duplicates and the solver constructor, timed with the library LARGESTEPS_HIP_LIB points at: python tools/time_dedup.py [workload ...]"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import numpy as np, torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.meshops import remove_duplicates
from largesteps.solvers import NestedDissectionSolver
from largesteps.normals import compute_face_normals, compute_vertex_normals
dev = torch.device("cuda:0")
for w in sys.argv[1:] or ["cfg3_dragon250k", "cfg4_plane1m"]:
    v, f, cfg = synthetic.config_mesh(w)
    rng = np.random.default_rng(0)
    soup = v[f.reshape(-1)]
    order = rng.permutation(soup.shape[0])
    inv = np.empty_like(order); inv[order] = np.arange(order.shape[0])
    vs = torch.from_numpy(soup[order]).to(dev)
    fs = torch.from_numpy(inv.reshape(-1, 3)).to(dev)
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        vu, fu, dup = remove_duplicates(vs, fs)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
    cs = []
    for _ in range(4):
        s = NestedDissectionSolver(M); cs.append(s.build_seconds * 1e3); del s
    ns = []
    for _ in range(4):
        tf2 = tf.clone()                        # a new face tensor object: the corner ranking (a radix sort) is rebuilt
        torch.cuda.synchronize(); t0 = time.perf_counter()
        compute_vertex_normals(tv, tf2, compute_face_normals(tv, tf2))
        torch.cuda.synchronize(); ns.append((time.perf_counter() - t0) * 1e3)
    print(f"{w}: soup of {vs.shape[0]} rows -> {vu.shape[0]} vertices: remove_duplicates " + " ".join(f"{t:.2f}" for t in ts) + " ms | constructor "
          + " ".join(f"{t:.1f}" for t in cs) + " ms | normals incl. corner ranking " + " ".join(f"{t:.2f}" for t in ns) + " ms", flush=True)
