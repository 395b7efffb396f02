#!/usr/bin/env python3
"""Micro A/B of kernel structures (tools/ubench/experiments.hip -- NOT part of the product library) on the 1M-vertex plane.
Build the side library first:  make -C tools/ubench      Prints us/launch and GB/s."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "large-steps-pytorch_amd")]
import torch  # noqa: E402
from largesteps.geometry import compute_matrix  # noqa: E402
from largesteps.solvers import PCGSolver  # noqa: E402
from largesteps import synthetic, _native  # noqa: E402

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4_plane1m"
v, f, c = synthetic.config_mesh(name)
tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
M = compute_matrix(tv, tf, c["lambda_"] or 0.0, alpha=c["alpha"], cotan=c["cotan"])
s = PCGSolver(M)
lib = _native.lib()
xlib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "build", "libls_ubench.so"))
xlib.ls_experiment.restype = ctypes.c_int
xlib.ls_experiment.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 7
sp, cv, ne = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
_native.check(lib.ls_solver_sell(s._handle, ctypes.byref(sp), ctypes.byref(cv), ctypes.byref(ne)))
V, nnz = v.shape[0], M._nnz()
dinv = torch.rand(V, device=dev) + 0.5
r = torch.randn(V, 3, device=dev)
p = torch.randn(V, 3, device=dev)
part = torch.rand(6 * 1024, dtype=torch.float64, device=dev)
st = _native.stream_of(dev)


def run(which, bs, grid, reps=200):
    def launch():
        _native.check(xlib.ls_experiment(which, bs, grid, V, _native.ptr(dinv), _native.ptr(r), _native.ptr(p), _native.ptr(part), sp, cv, st))
    for _ in range(5):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{name}: V={V} nnz={nnz} sell_entries={ne.value}")
b3 = 40 * V
for label, which, combos in [
    ("copy4 (24V bytes)", 4, [(256, 2048), (256, 4096), (1024, 256), (1024, 512), (1024, 1024)]),
    ("k3_row", 0, [(256, 1024), (256, 2048), (256, 4096), (1024, 256), (1024, 512), (1024, 1024)]),
    ("k3_vec", 1, [(256, 977), (256, 2048), (1024, 245), (1024, 256), (1024, 512)]),
    ("k3_vec_prologue", 2, [(256, 977), (1024, 245), (1024, 256)]),
    ("k3_vec_hoist", 3, [(256, 977), (1024, 245)]),
]:
    for bs, grid in combos:
        us = run(which, bs, grid)
        nb = 24 * V if which == 4 else b3
        print(f"  {label:18s} bs={bs:4d} grid={grid:5d}: {us:7.2f} us  {nb / us / 1e3:7.0f} GB/s", flush=True)
b1 = 8 * nnz + 4 * (V + 1) + 24 * V
for label, which in [("spmv_cur", 10), ("spmv_prefetch", 11)]:
    for bs, grid in [(256, 1024), (256, 2048), (256, 3907), (1024, 256), (1024, 512), (1024, 977)]:
        us = run(which, bs, grid)
        print(f"  {label:18s} bs={bs:4d} grid={grid:5d}: {us:7.2f} us  {b1 / us / 1e3:7.0f} GB/s", flush=True)

# ---- the production kernels, one phase at a time (scalars frozen at their initial values) ----------------
b = torch.randn(V, 3, device=dev)
x = torch.empty_like(b)
for blk, grid in [(256, 512), (256, 1024)]:
    s.set_option("block", blk); s.set_option("grid", grid)
    def phase(ph, it=0):
        _native.check(lib.ls_solver_phase(s._handle, ph, _native.ptr(b), _native.ptr(x), 3, 1e-12, 0.0, it, st))
    phase(0); phase(1)
    for ph, label, nb in [(2, "K1 spmv_dot", b1), (3, "K2 update", 76 * V), (4, "K3 direction", 40 * V)]:
        for _ in range(5):
            phase(ph)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            phase(ph)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        print(f"  phase {label:14s} bs={blk:4d} grid={grid:5d}: {us:7.2f} us  {nb / us / 1e3:7.0f} GB/s (same kernel repeated)", flush=True)
    # the real sequence K1,K2,K3 repeated (scalars frozen): working set of a real iteration
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        phase(2); phase(3); phase(4)
    e0.record()
    for _ in range(100):
        phase(2); phase(3); phase(4)
    e1.record()
    torch.cuda.synchronize()
    print(f"  sequence K1+K2+K3 bs={blk} grid={grid}: {e0.elapsed_time(e1) / 100 * 1e3:7.2f} us/iteration", flush=True)
