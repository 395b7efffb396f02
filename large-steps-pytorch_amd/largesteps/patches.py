"""
Patch plan of the LDS-resident, s-step Chebyshev kernel (csrc/pcg.hip, k_patch_cheb): a thin owner of the arrays the native
analysis produces (csrc/patch_plan.cpp behind ls_patch_plan_*: recursive coordinate bisection into equal patches, scan-line order,
ghost layers, patch-local ELL ids -- host threads, no numpy arithmetic here). The layout is described in the header
(include/largesteps_hip.h) and restated, as a numpy statement the CPU tests check the native plan against, in
tests/patch_plan_statement.py.

Setup work of the iterative stand-in for the reference's default solver (largesteps/solvers.py:26-39); the direct solver does
not use it.
"""
import ctypes

import numpy as np

from . import _native

MAX_DEPTH = 12
TABLE_COLS = 8 + MAX_DEPTH


class PatchPlan:
    def __init__(self, V, perm, depth, patch_size, table, ghost_gid, cols16, diag, max_local, max_rows, max_width, seconds=0.0):
        self.V, self.perm, self.depth, self.patch_size = int(V), perm, int(depth), int(patch_size)
        self.table, self.ghost_gid, self.cols16, self.diag = table, ghost_gid, cols16, diag
        self.n_patches = int(table.shape[0])
        self.max_local, self.max_rows, self.max_width = int(max_local), int(max_rows), int(max_width)
        self.seconds = float(seconds)

    @property
    def redundancy(self):
        """computed rows / owned rows"""
        return float(self.table[:, 2].sum()) / max(1, int(self.table[:, 1].sum()))

    @staticmethod
    def build(rowptr, col, diag, positions, patch_size=4096, depth=8, cap_local=6500, min_depth=2, cap_rows=8192):
        """rowptr / col: CSR pattern of M (diagonal included); diag: (V,) diagonal of M; positions: (V, 3). Returns the deepest
        plan <= depth whose patches fit cap_local local vertices and cap_rows computed rows, or None (the caller keeps the
        one-step kernel)."""
        if not 1 <= depth <= MAX_DEPTH:
            raise ValueError(f"patch depth must be in [1, {MAX_DEPTH}]")
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        diag = np.ascontiguousarray(diag, dtype=np.float32)
        pos = np.ascontiguousarray(positions, dtype=np.float32)
        V = rowptr.shape[0] - 1
        if V <= 0:
            return None
        lib = _native.lib()
        as_p = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
        h = ctypes.c_void_p(None)
        _native.check(lib.ls_patch_plan_create(V, as_p(rowptr), as_p(col), as_p(diag), as_p(pos), int(patch_size), int(depth), int(cap_local),
                                               int(min_depth), int(cap_rows), ctypes.byref(h)))
        if not h.value:
            return None
        try:
            n, d, ml, mr, mw = (ctypes.c_int(0) for _ in range(5))
            ng, nc, nd = (ctypes.c_int64(0) for _ in range(3))
            sec = ctypes.c_double(0.0)
            _native.check(lib.ls_patch_plan_info(h, ctypes.byref(n), ctypes.byref(d), ctypes.byref(ml), ctypes.byref(mr), ctypes.byref(mw),
                                                 ctypes.byref(ng), ctypes.byref(nc), ctypes.byref(nd), ctypes.byref(sec)))
            table = np.empty((n.value, TABLE_COLS), dtype=np.int32)
            gid = np.empty(ng.value, dtype=np.int32)
            cols = np.empty(nc.value, dtype=np.uint16)
            dg = np.empty(nd.value, dtype=np.float32)
            perm = np.empty(V, dtype=np.int32)
            _native.check(lib.ls_patch_plan_arrays(h, as_p(table), as_p(gid), as_p(cols), as_p(dg), as_p(perm)))
        finally:
            lib.ls_patch_plan_destroy(h)
        return PatchPlan(V, perm.astype(np.int64), d.value, patch_size, table, gid, cols, dg, ml.value, mr.value, mw.value, sec.value)
