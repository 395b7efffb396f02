#!/usr/bin/env python3
"""Reduced-precision STORAGE of the direct solver's factor, emulated (csrc/experiments/round_study.h: the experiments library rounds
what k_convert writes; the solve kernels are the product's). For each workload and each (format, arrays) setting: factorise, solve
u = M v, report max |x - v| / max |v| (the round trip of SURVEY G7) and, where the fp64 oracle finishes in seconds, the error against
the oracle's solution of the same system with a white right-hand side.
    LARGESTEPS_HIP_LIB=tools/build/liblargesteps_hip_exp.so python tools/precision_study.py [workloads...]
Kill criterion: a format must stay <= 2e-5 on all workloads to be worth building."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "large-steps-pytorch_amd")]
import numpy as np
import torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential
from largesteps.solvers import NestedDissectionSolver

dev = torch.device("cuda:0")
# (label, mantissa bits, array mask: 1 leaf triangles | 2 dense tier | 4 upper levels, from level)
SETTINGS = [("fp32", 23, 0, 0),
            ("24-bit all", 15, 7, 0), ("24-bit leaves+tier", 15, 3, 0), ("24-bit leaves", 15, 1, 0),
            ("20-bit (m=11) all", 11, 7, 0), ("fp16 all", -16, 7, 0), ("fp16 leaves", -16, 1, 0), ("fp16 leaves+tier", -16, 3, 0),
            ("m=12 leaves+tier", 12, 3, 0), ("m=13 leaves+tier", 13, 3, 0), ("bf16 all", 7, 7, 0),
            # dense blocks only (the leaves' triangles stay fp32)
            ("24-bit tier+upper", 15, 6, 0), ("24-bit upper", 15, 4, 0), ("24-bit tier dense", 15, 2, 0), ("m=13 tier+upper", 13, 6, 0),
            ("m=11 tier+upper", 11, 6, 0), ("fp16 tier+upper", -16, 6, 0), ("fp16 upper", -16, 4, 0), ("bf16 tier+upper", 7, 6, 0)]
if os.environ.get("PS_ONLY"):
    SETTINGS = [t for t in SETTINGS if any(k in t[0] for k in os.environ["PS_ONLY"].split(","))]
names = sys.argv[1:] or ["cfg3_dragon250k", "cfg4_plane1m", "folded250k", "cfg5_plane4m"]
for name in names:
    v, f, cfg = synthetic.config_mesh(name)
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
    u = to_differential(M, tv)
    vmax = float(tv.abs().max())
    x64 = bw = None
    if v.shape[0] <= 1_100_000 and not os.environ.get("PS_NO_ORACLE"):
        from oracle import solve as osv
        idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
        bw_np = np.random.default_rng(0).standard_normal(v.shape).astype(np.float32)
        x64 = osv.from_differential(idx[0], idx[1], val, bw_np)
        bw = torch.from_numpy(bw_np).to(dev)
    for label, bits, mask, lv in SETTINGS:
        os.environ["LS_ND_ROUND_BITS"], os.environ["LS_ND_ROUND_MASK"], os.environ["LS_ND_ROUND_FROM"] = str(bits), str(mask), str(lv)
        s = NestedDissectionSolver(M)
        x = s.solve(u)
        e_v = float((x - tv).abs().max()) / vmax
        line = f"{name:18s} {label:22s} round trip {e_v:.2e}"
        if x64 is not None:
            xw = s.solve(bw).cpu().numpy()
            line += f"   vs fp64 oracle (white rhs) {np.abs(xw - x64).max() / np.abs(x64).max():.2e}"
        print(line, flush=True)
        s.close() if hasattr(s, "close") else None
        del s
