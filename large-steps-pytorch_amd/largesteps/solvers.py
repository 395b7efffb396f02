"""
Sparse SPD solvers of the parameterization path (reference: largesteps/solvers.py).

Same classes and protocol as the reference (Solver :6, CholeskySolver :26, ConjugateGradientSolver :41,
DifferentiableSolve :128, solve :148).
    CholeskySolver            factor once / re-solve: NestedDissectionSolver (csrc/direct.hip) for matrices built by
                              compute_matrix, IterativeCholeskySolver (Chebyshev-Jacobi / Jacobi-PCG, csrc/pcg.hip) otherwise
    ConjugateGradientSolver   the reference's stopping rule and warm start on the HIP Jacobi-PCG
Neither cholespy/CHOLMOD nor torch sparse ops are used; there is no CPU path.
"""
import ctypes
import os
import types
import warnings

import numpy as np
import torch
from torch.autograd import Function

from . import _native

_KMAX = 4   # right-hand-side columns one native solve handles; wider b is solved in column groups


class Solver:
    """
    What `solve()` below expects of a solver object (reference: solvers.py:6-24): a constructor that takes the matrix and a
    `solve(b, backward)` method. Subclass it to plug in any other linear solver.
    """
    def __init__(self, M):
        pass

    def solve(self, b, backward=False):
        """
        Return x with M x = b.

        Parameters
        ----------
        b : torch.Tensor
            Right-hand side(s), one column per system
        backward : bool (optional)
            True when called for the gradient (solvers that warm start keep the two directions apart)
        """
        raise NotImplementedError()


class PCGSolver(Solver):
    """
    Jacobi-preconditioned conjugate gradients on the MI355X (one native handle per matrix).

    Parameters
    ----------
    M : torch.sparse_coo_tensor
        SPD system matrix (coalesced, fp32, on a HIP device), normally from `compute_matrix`.
    rtol, atol : float
        A column is converged when ||r||_2 <= max(rtol * ||b||_2, atol).
    max_iter : int
        Iteration cap (the reference's CG has none and can spin forever).
    warm_start : bool
        Start from the previous solution of the same pass (forward / backward kept apart, like
        solvers.py:102-124) instead of zero.
    chebyshev : bool
        Use the Chebyshev-accelerated Jacobi iteration (one kernel per iteration, no dot products) when the
        matrix comes with a certified spectral enclosure (matrices built by `compute_matrix`); falls back to
        PCG otherwise, or if its final residual check fails. `rtol` is then the a-priori guaranteed reduction
        of the residual; `last_info['rnorm']` is the true fp32 residual of the returned solution.
    chebyshev_cap : int
        Largest a-priori Chebyshev iteration count (cold start, at `rtol`) for which Chebyshev is preferred to PCG.
    patch_min_vertices : int
        From this mesh size on (and for uniform-Laplacian matrices whose vertex positions are known) the Chebyshev
        steps run in the LDS-resident patch kernel, several iterations per launch (largesteps/patches.py); smaller
        meshes cannot give every CU a patch worth keeping resident and stay with the one-step kernel.
    patch_columns : int
        The largest number of right-hand-side columns the patch kernel has to serve (wider solves use the one-step
        kernel): fewer columns leave LDS room for larger patches and deeper plans (12 instead of 8 steps per launch).
    """

    def __init__(self, M, rtol=1e-6, atol=0.0, max_iter=10000, warm_start=False, chebyshev=False, chebyshev_cap=400,
                 patch_min_vertices=400000, patch_columns=3):
        if not 1 <= int(patch_columns) <= _KMAX:
            raise ValueError(f"patch_columns must be in [1, {_KMAX}]")
        self.patch_columns = int(patch_columns)
        csr = _native.csr_of(M)
        self._csr = csr                 # keeps rowptr/col/val alive; never M itself (cache eviction relies on it)
        self.rtol, self.atol, self.max_iter, self.warm_start = float(rtol), float(atol), int(max_iter), bool(warm_start)
        self.guess_fwd = None
        self.guess_bwd = None
        self.last_info = None
        self._handle = ctypes.c_void_p(None)
        dev = csr.device
        with torch.cuda.device(dev):
            _native.check(_native.lib().ls_solver_create(_native.ptr(csr.rowptr), _native.ptr(csr.col), _native.ptr(csr.val),
                                                         csr.V, csr.nnz, _KMAX, dev.index, _native.stream_of(dev),
                                                         ctypes.byref(self._handle)))
        self.chebyshev = False
        self.chebyshev_iterations = None
        self.implicit_values = False      # True: the Chebyshev kernel reads neighbour ids only (uniform Laplacian)
        self.patch_plan = None            # PatchPlan of the LDS-resident s-step kernel, if one is in use
        if csr.a_min is not None:
            _native.check(_native.lib().ls_solver_set_spectrum(self._handle, float(csr.a_min)))
            if chebyshev and (self.rtol > 0.0 or self.atol > 0.0):
                # The count is known a priori from the enclosure. A Chebyshev step costs ~0.4 of a PCG iteration, but
                # PCG adapts to the actual spectrum: with a loose enclosure (cotangent weights of sliver triangles, one
                # vertex of very high valence) the bound explodes and PCG wins -- keep Chebyshev only below the cap.
                n = ctypes.c_int(0)
                # (absolute tolerance only: the count depends on the starting residual -- judge the enclosure by a 1e-6 reduction)
                _native.check(_native.lib().ls_solver_chebyshev_iterations(self._handle, self.rtol if self.rtol > 0.0 else 1e-6,
                                                                           ctypes.byref(n)))
                self.chebyshev_iterations = n.value
                self.chebyshev = n.value <= int(chebyshev_cap)
                if self.chebyshev and csr.uniform is not None and not os.environ.get("LARGESTEPS_EXPLICIT_VALUES"):
                    # uniform Laplacian: every off-diagonal entry is -b, the solver reads neighbour ids only
                    with torch.cuda.device(dev):
                        _native.check(_native.lib().ls_solver_set_uniform(self._handle, float(csr.uniform[0]), float(csr.uniform[1]),
                                                                          _native.stream_of(dev)))
                    self.implicit_values = True
                    if csr.positions is not None and csr.V >= patch_min_vertices and not os.environ.get("LARGESTEPS_NO_PATCHES"):
                        self._set_patches(csr, dev)

    def _set_patches(self, csr, dev):
        """Host-side analysis for the patch kernel (csrc/pcg.hip k_patch_cheb): Morton-ordered patches + ghost layers.
        A mesh whose patches do not fit LDS even at depth 2 simply keeps the one-step kernel."""
        from .patches import PatchPlan
        rowptr, col = csr.rowptr.cpu().numpy(), csr.col.cpu().numpy()
        rows = torch.repeat_interleave(torch.arange(csr.V, device=dev), (csr.rowptr[1:] - csr.rowptr[:-1]).long())
        diag = torch.zeros(csr.V, dtype=torch.float32, device=dev)
        on = rows == csr.col.long()
        diag[rows[on]] = csr.val[on]
        # LDS budget: 2 buffers x (n_local+1) x 4k B <= 160 KiB -> n_local <= 6800 for k = 3 columns; a solver that is
        # only ever asked for k <= 2 columns (largesteps.distributed.ColumnSharded) fits larger patches -> deeper plans.
        # LARGESTEPS_PATCH="patch_size,depth,cap_local,min_depth" overrides the defaults (tuning / tests).
        kc = self.patch_columns
        default = "4096,8,6800,4" if kc >= 3 else f"4096,12,{min(160 * 1024 // (8 * kc) - 16, 60000)},4"
        ps, depth, cap_local, min_depth = (int(t) for t in (os.environ.get("LARGESTEPS_PATCH", default) + ",4").split(",")[:4])
        plan = PatchPlan.build(rowptr, col, diag.cpu().numpy(), csr.positions.cpu().numpy(), patch_size=ps, depth=depth,
                               cap_local=cap_local)
        if plan is None or plan.max_rows > 8192 or plan.depth < min_depth:
            return      # shallow plans (spread-out patches) are not worth the redundant work: keep the one-step kernel
        tab = np.ascontiguousarray(plan.table.reshape(-1))
        perm32 = plan.perm.astype(np.int32)
        as_p = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
        with torch.cuda.device(dev):
            _native.check(_native.lib().ls_solver_set_patches(self._handle, as_p(tab), plan.n_patches, as_p(plan.ghost_gid),
                                                              plan.ghost_gid.shape[0], as_p(plan.cols16), plan.cols16.shape[0],
                                                              as_p(plan.diag), plan.diag.shape[0], as_p(perm32), plan.depth,
                                                              plan.max_local, plan.max_rows, _native.stream_of(dev)))
        self.patch_plan = plan

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                _native.lib().ls_solver_destroy(h)
            except Exception:      # interpreter shutdown: module globals may already be gone
                pass
            self._handle = None

    def set_option(self, name, value):
        """Measurement knobs of the native solver: 'check_every', 'grid', 'block' (256 / 512 / 1024), 'graph', 'profile'."""
        _native.check(_native.lib().ls_solver_set(self._handle, name.encode(), int(value)))

    def kernel_profile(self):
        """(ms_K1, ms_K2, ms_K3, iterations) of the last solve run with set_option('profile', 1)."""
        ms = (ctypes.c_double * 3)()
        it = ctypes.c_int(0)
        _native.check(_native.lib().ls_solver_profile(self._handle, ctypes.byref(ms), ctypes.byref(it)))
        return ms[0], ms[1], ms[2], it.value

    def _solve_block(self, b, x0):
        csr = self._csr
        dev = csr.device
        x = torch.empty_like(b)
        info = _native.SolveInfo()
        lib = _native.lib()
        args = (self._handle, _native.ptr(b), _native.ptr(x0) if x0 is not None else None, _native.ptr(x), b.shape[1],
                self.rtol, self.atol, self.max_iter, ctypes.byref(info), _native.stream_of(dev))
        method = "pcg"
        with torch.cuda.device(dev):
            rc = None
            if self.chebyshev:
                rc = lib.ls_solver_solve_chebyshev(*args)
                method = "chebyshev"
                if rc in (_native.LS_E_NOT_CONVERGED, _native.LS_E_STATE):
                    warnings.warn(f"largesteps: {_native.last_error()}; falling back to PCG", RuntimeWarning, stacklevel=3)
                    rc = None
            if rc is None:
                rc = lib.ls_solver_solve(*args)
                method = "pcg"
        self.last_info = dict(iterations=info.iterations, converged=bool(info.converged), method=method,
                              rnorm=list(info.rnorm)[:b.shape[1]], bnorm=list(info.bnorm)[:b.shape[1]])
        if rc == _native.LS_E_NOT_CONVERGED:
            msg = _native.last_error()
            if "not converged" in msg:
                warnings.warn(f"largesteps: {msg}; returning the last iterate", RuntimeWarning, stacklevel=3)
            else:
                raise RuntimeError(f"largesteps: {msg}")
        else:
            _native.check(rc)
        return x

    @_native.retry_on_oom
    def solve(self, b, backward=False):
        _native.require_device(b, "b")
        csr = self._csr
        if b.device != csr.device:
            raise RuntimeError(f"matrix ({csr.device}) and b ({b.device}) must be on the same device")
        if b.dtype != torch.float32:
            raise TypeError(f"b must be float32, got {b.dtype}")
        if b.dim() not in (1, 2) or b.shape[0] != csr.V:
            raise ValueError(f"Invalid array shape {b.shape} for solve: expected ({csr.V}, k)")
        squeeze = b.dim() == 1
        b2 = (b.detach().unsqueeze(1) if squeeze else b.detach()).contiguous()
        k = b2.shape[1]
        x0 = None
        if self.warm_start:
            g = self.guess_bwd if backward else self.guess_fwd
            if g is not None and g.shape == b2.shape:
                x0 = g
        if k <= _KMAX:
            x = self._solve_block(b2, x0)
        else:
            x = torch.empty_like(b2)
            for c0 in range(0, k, _KMAX):
                c1 = min(c0 + _KMAX, k)
                g0 = x0[:, c0:c1].contiguous() if x0 is not None else None
                x[:, c0:c1] = self._solve_block(b2[:, c0:c1].contiguous(), g0)
        if self.warm_start:
            # like the reference (solvers.py:120-124) the guess aliases the returned tensor
            if backward:
                self.guess_bwd = x
            else:
                self.guess_fwd = x
        return x.squeeze(1) if squeeze else x


class IterativeCholeskySolver(PCGSolver):
    """
    The iterative stand-in for a direct solve: every call is a cold-started iteration run to a residual reduction of
    1e-6, i.e. the result is a function of b only. For matrices from `compute_matrix` (spectral enclosure known) the
    iteration is the Chebyshev-accelerated Jacobi method -- one HIP kernel per iteration (or per 8-12 iterations on
    LDS-resident patches), no reductions; any other matrix is solved by the Jacobi-PCG. Used by `CholeskySolver` when
    the matrix cannot be factorised by the nested-dissection solver (no vertex positions / fronts too large).
    """

    def __init__(self, M, rtol=1e-6, max_iter=10000, chebyshev=True, patch_columns=3):
        super().__init__(M, rtol=rtol, atol=0.0, max_iter=max_iter, warm_start=False, chebyshev=chebyshev,
                         patch_columns=patch_columns)


class _NativeDirect:
    """Owner of a native ls_direct handle built by ls_direct_factor (symbolic analysis + numeric factorisation behind the C ABI)."""

    _ORDERINGS = {None: -1, "auto": -1, "longest-axis": 0, "trial-cuts": 1}      # LS_ND_ORDER_AUTO / _LONGEST / _MINSEP

    def __init__(self, csr, leaf_size, arity, tier_levels, sparse_leaves, shard=(0, 1), ordering=None, tier_waves=0):
        self.device = csr.device
        self._h = ctypes.c_void_p(None)
        pos = csr.positions
        if pos is not None:
            pos = pos.detach().to(torch.float32).contiguous()
        dev = csr.device
        if ordering not in self._ORDERINGS:
            raise ValueError(f"ordering must be one of {sorted(k for k in self._ORDERINGS if k)} or None, got {ordering!r}")
        if tier_waves not in (0, 4, 8, 16):
            raise ValueError(f"tier_waves must be 0 (the library's rule), 4, 8 or 16, got {tier_waves!r}")
        # every choice travels as an argument of the C ABI (ls_direct_factor_ex); the process environment is not touched
        lib = _native.lib()
        opt = _native.DirectOptions()
        _native.check(lib.ls_direct_options_default(ctypes.byref(opt)))
        opt.leaf_size, opt.arity, opt.tier_levels, opt.sparse_leaves = int(leaf_size or 0), int(arity or 0), int(tier_levels), int(bool(sparse_leaves))
        opt.shard_rank, opt.shard_count = int(shard[0]), int(shard[1])
        opt.ordering, opt.tier_waves = self._ORDERINGS[ordering], int(tier_waves)
        with torch.cuda.device(dev):
            _native.check(lib.ls_direct_factor_ex(_native.ptr(csr.rowptr), _native.ptr(csr.col), _native.ptr(csr.val), csr.V, csr.nnz,
                                                  _native.ptr(pos), ctypes.byref(opt), dev.index, _native.stream_of(dev), ctypes.byref(self._h)))
        s3 = (ctypes.c_double * 3)()
        _native.check(_native.lib().ls_direct_factor_seconds(self._h, ctypes.byref(s3)))
        self.timings = dict(plan_seconds=s3[0], table_seconds=s3[1], factor_seconds=s3[2])
        self.tier_levels = int(tier_levels)
        self._solve = _native.lib().ls_direct_solve
        o, w, sp, wo = ctypes.c_int(0), ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
        _native.check(_native.lib().ls_direct_plan_quality(self._h, ctypes.byref(o), ctypes.byref(w), ctypes.byref(sp), ctypes.byref(wo)))
        # how the dissection's cutting directions were chosen and what it costs (include/largesteps_hip.h, ls_nd_plan_quality)
        self.plan_quality = dict(ordering="trial-cuts" if o.value == 1 else "longest-axis", words_per_vertex=w.value, spread=sp.value,
                                 words_per_vertex_other=wo.value)

    def close(self):
        """Destroy the native handle now (its factor arrays go to the library's buffer pool, where the next construction of a similar size
        finds them); idempotent. Solving with a closed solver raises."""
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _native.lib().ls_direct_destroy(h)
            except Exception:      # interpreter shutdown: module globals may already be gone
                pass
        self._h = ctypes.c_void_p(None)

    def __del__(self):
        self.close()

    def solve(self, b, x):
        if not self._h.value:
            raise RuntimeError("this solver was closed")
        # (no torch.cuda.device context here: ls_direct_solve selects the handle's device itself, and the stream is looked up for that
        #  device by index -- the context manager cost more host time than the call at the reference's mesh sizes)
        rc = self._solve(self._h, b.data_ptr(), x.data_ptr(), b.shape[1], _native.raw_stream(self.device))
        if rc:
            _native.check(rc)

    def set_option(self, name, value):
        _native.check(_native.lib().ls_direct_set(self._h, name.encode(), int(value)))

    def info(self):
        fe, nl = ctypes.c_int64(0), ctypes.c_int(0)
        ms = (ctypes.c_double * 3)()
        _native.check(_native.lib().ls_direct_info(self._h, ctypes.byref(fe), ctypes.byref(nl), ms))
        lv, ar, th, tw = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        wu, wd, nb = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        _native.check(_native.lib().ls_direct_shape(self._h, ctypes.byref(lv), ctypes.byref(ar), ctypes.byref(th), ctypes.byref(tw),
                                                    ctypes.byref(wu), ctypes.byref(wd), ctypes.byref(nb)))
        return dict(factor_entries=fe.value, launches=nl.value, up_ms=ms[0], down_ms=ms[1], mid_ms=ms[2], levels=lv.value, arity=ar.value,
                    tier_levels=th.value, tier_workgroups=tw.value, words_up=wu.value, words_down=wd.value, n_bnd=nb.value)


    def level_words(self):
        """fp32 words of factor data per tree level: (up sweep, down sweep), level 0 = root"""
        n = self.info()["levels"]
        up, down = (ctypes.c_int64 * n)(), (ctypes.c_int64 * n)()
        _native.check(_native.lib().ls_direct_level_words(self._h, n, up, down))
        return list(up), list(down)

    def level_rows(self):
        """own rows (vertices) and boundary entries per tree level, level 0 = root"""
        n = self.info()["levels"]
        rows, bnd = (ctypes.c_int64 * n)(), (ctypes.c_int64 * n)()
        _native.check(_native.lib().ls_direct_level_rows(self._h, n, rows, bnd))
        return list(rows), list(bnd)

    def level_index_bytes(self):
        """bytes of static index data per tree level: (up sweep, down sweep), level 0 = root (ls_direct_level_index_bytes)"""
        n = self.info()["levels"]
        up, down = (ctypes.c_int64 * n)(), (ctypes.c_int64 * n)()
        _native.check(_native.lib().ls_direct_level_index_bytes(self._h, n, up, down))
        return list(up), list(down)

    def tier_balance(self):
        """factor words of the tier's subtrees: dict(up_max, up_mean, down_max, down_mean, max_over_mean) (ls_direct_tier_balance)"""
        w = (ctypes.c_double * 4)()
        _native.check(_native.lib().ls_direct_tier_balance(self._h, ctypes.byref(w)))
        tot_max, tot_mean = w[0] + w[2], w[1] + w[3]
        return dict(up_max=w[0], up_mean=w[1], down_max=w[2], down_mean=w[3], max_over_mean=(tot_max / tot_mean if tot_mean else None))

    def launch_profile(self):
        """Launches of the last solve run with set_option("profile", 3): dicts of ms, factor words, levels (lo, hi), sweep."""
        n = ctypes.c_int(0)
        _native.check(_native.lib().ls_direct_launch_profile(self._h, 0, ctypes.byref(n), None, None, None, None, None))
        ms, words = (ctypes.c_double * n.value)(), (ctypes.c_int64 * n.value)()
        lo, hi, sw = (ctypes.c_int32 * n.value)(), (ctypes.c_int32 * n.value)(), (ctypes.c_int32 * n.value)()
        _native.check(_native.lib().ls_direct_launch_profile(self._h, n.value, ctypes.byref(n), ms, words, lo, hi, sw))
        return [dict(ms=ms[i], words=words[i], levels=(lo[i], hi[i]), sweep=("up", "down", "both")[sw[i]]) for i in range(n.value)]


def release_scratch(device=None):
    """Free the large device buffers the direct solver keeps between constructions (constructor scratch and the factor arrays of destroyed
    solvers: 4 GB at 1M vertices; per device at most LS_POOL_GB = 24 GB and a quarter of the device's memory; see ls_release_scratch in
    include/largesteps_hip.h). The library does this itself when one of ITS allocations fails; torch's caching allocator cannot see these
    buffers, so a caller that catches torch.cuda.OutOfMemoryError should call this (and torch.cuda.empty_cache()) before it retries.
    device: a torch device / index, or None for every device."""
    idx = -1 if device is None else (device.index if isinstance(device, torch.device) else int(device))
    _native.check(_native.lib().ls_release_scratch(-1 if idx is None else idx))


# What every direct solve reports. A direct solve has no stopping rule: `converged` only says that no iteration was cut short -- the
# accuracy statement of this path is the FORWARD error against an fp64 solution (<= 1e-4 relative; tests/test_gpu_parity.py,
# bench.py `config.tolerance`), not a residual. Read-only: one object is shared by every solver of the process.
_DIRECT_INFO = types.MappingProxyType(dict(iterations=0, converged=True, method="nested-dissection"))


class NestedDissectionSolver(Solver):
    """
    Factor-once / re-solve direct solver (what the reference's default method does through cholespy / CHOLMOD,
    solvers.py:26-39), MI355X-native and entirely behind the C ABI (`ls_direct_factor`): geometric nested dissection of
    the mesh (bisection rounds on the device, csrc/nd_bisect.hip; tree and index lists on host threads, csrc/nd_plan.cpp),
    multifrontal numeric factorisation in fp64 on the device with hand-written kernels (csrc/nd_factor.hip), and a re-solve of one launch per upper tree level and sweep plus one
    launch per sweep for the deepest levels (csrc/direct.hip, csrc/nd_tier.h). The result is a function of b only and
    bitwise reproducible.

    leaf_size=None / arity=None let the library pick the tree from the size of the system: one dense node (ONE launch per
    re-solve) up to 1280 vertices, shallow trees up to 32k vertices, three bisection rounds per level (arity 8) between 12k and
    300k vertices, the arity-4 tree with 64-vertex sparse leaves of the large-mesh kernels beyond.

    The dissection uses the vertex positions the matrix was assembled from (`compute_matrix`); a symmetric matrix built
    elsewhere gets graph-distance pseudo-positions instead. A surface that is folded or rolled up in space (cloth, a scroll, shells
    inside each other) is recognised by its thick separators and dissected again with graph distances among the cutting directions
    (`plan_quality['ordering'] == 'trial-cuts'`). ordering='trial-cuts' (or LS_ND_ORDER=1) asks for those trial cuts always: 5-10 % fewer
    factor numbers on rough closed scans (the 250k cotangent config: 0.102 -> 0.090 ms per solve) for 10-25 ms more constructor --
    worth it for a long captured run on one mesh; 'longest-axis' never tries them; 'auto' is the library's rule for THIS matrix alone;
    None (the default) is that rule plus what the process has learnt from its previous solvers of the same size class -- a surface the rule
    found suspect once gets the trial cuts at once at its next construction (one plan instead of two: a remesh of a closed scan at 1M
    vertices constructs in 78 ms instead of 150), and a size class whose previous solver served >= AUTO_TRIAL_CUTS_AFTER solves gets them too.
    tier_waves=4 / 8 / 16 picks the tier kernel's workgroup shape (0: the library's rule; A/B runs and tests). Raises ValueError when the matrix is not symmetric or not
    positive definite, RuntimeError when the mesh does not dissect into fronts that fit the kernels.
    """

    # A remesh loop tells the library its own period: when the previous solver of the same size class (same device, vertex count
    # within ~9 %) served at least this many solves, the next construction asks for the trial cuts (ordering='trial-cuts': 10-25 ms more
    # constructor, 5-12 % fewer factor numbers on rough closed scans -- the 250k cotangent config 0.102 -> 0.090 ms per solve, i.e.
    # ~1500 solves to earn the longer construction back; the better of the two plans is kept, so a flat sheet only loses the time).
    # Only where the trial cuts were measured to pay (100k .. 600k vertices) and only when the caller left `ordering` to the library.
    AUTO_TRIAL_CUTS_AFTER = 1500
    _served = {}                     # (device index, size class) -> solves the last closed solver of that class served
    _suspect = {}                    # (device index, size class) -> True: the automatic rule ended up with the trial cuts for the last mesh of that class

    @staticmethod
    def _size_class(V):
        import math
        return int(round(8.0 * math.log2(max(int(V), 1))))

    def _surface_key(self, csr):
        """the size class + the bounding box of the positions, each extent to ~19 %: 'the same surface, remeshed' (a folded sheet and a flat
        one of the same vertex count differ in it). One small reduction and a host read per construction."""
        key = getattr(self, "_surface_key_cached", None)
        if key is None:
            box = ()
            if csr.positions is not None and csr.V:
                import math
                p = csr.positions.detach()
                ext = (p.amax(0) - p.amin(0)).tolist()
                box = tuple(int(round(4.0 * math.log2(e))) if e > 0 and math.isfinite(e) else -999 for e in ext)
            key = self._surface_key_cached = self._class_key + box
        return key

    def __init__(self, M, leaf_size=None, arity=None, shard=(0, 1), ordering=None, tier_waves=0):
        import time
        csr = _native.csr_of(M)
        self._csr = csr
        self.last_info = None
        self.solves_served = 0
        self._class_key = (csr.device.index, self._size_class(csr.V))
        self.ordering_requested = ordering
        if ordering is None and not os.environ.get("LS_ND_ORDER"):
            if 100_000 <= csr.V <= 600_000 and self._served.get(self._class_key, 0) >= self.AUTO_TRIAL_CUTS_AFTER:
                ordering = "trial-cuts"
            elif self._suspect.get(self._surface_key(csr)):
                # the automatic rule found the previous mesh of this size class folded / rough (its longest-axis plan was built, found suspect and
                # replaced by the trial-cut plan): a remesh of the same surface will be too -- ask for the trial cuts at once, one plan instead of two
                ordering = "trial-cuts"
        self.ordering_used = ordering
        # a matrix that was not built by compute_matrix: the factorisation needs M = M^T (up to rounding: 1e-6 of the largest entry)
        if not _native.is_symmetric(csr):
            raise ValueError("NestedDissectionSolver: the matrix is not symmetric")
        t0 = time.perf_counter()
        tier = max(-1, min(6, int(os.environ.get("LS_ND_TIER_H", "-1"))))      # -1: the library picks (and never picks one that does not fit)
        sparse = not os.environ.get("LS_ND_DENSE_LEAVES")
        self._direct = _NativeDirect(csr, leaf_size, arity, tier, sparse, shard=shard, ordering=ordering, tier_waves=tier_waves)
        torch.cuda.synchronize(csr.device)
        self.build_seconds = time.perf_counter() - t0
        self.timings = self._direct.timings
        self.plan_quality = self._direct.plan_quality
        if self.ordering_requested is None and self.ordering_used is None:
            if len(self._suspect) > 64:
                self._suspect.clear()
            self._suspect[self._surface_key(csr)] = self.plan_quality["ordering"] == "trial-cuts"

    @_native.retry_on_oom
    def solve(self, b, backward=False):
        _native.require_device(b, "b")
        if b.device != self._csr.device:
            raise RuntimeError(f"matrix ({self._csr.device}) and b ({b.device}) must be on the same device")
        if b.dtype != torch.float32:
            raise TypeError(f"b must be float32, got {b.dtype}")
        if b.dim() not in (1, 2) or b.shape[0] != self._csr.V:
            raise ValueError(f"Invalid right-hand side shape {tuple(b.shape)}: expected ({self._csr.V}, k)")
        squeeze = b.dim() == 1
        self.solves_served += 1
        if not squeeze and b.is_contiguous():            # the common case: (V, k) contiguous as it is -- no view, no copy (the pointer is all the call takes)
            b32 = b
            x = torch.empty(b.shape, dtype=torch.float32, device=b.device)
        else:
            b32 = (b.detach().unsqueeze(1) if squeeze else b.detach()).contiguous()
            x = torch.empty_like(b32)
        if b32.shape[1] <= _KMAX:                        # k = 3 coordinates: one native call, no column loop
            self._direct.solve(b32, x)
        else:
            for c0 in range(0, b32.shape[1], _KMAX):
                c1 = min(b32.shape[1], c0 + _KMAX)
                xb = torch.empty((b32.shape[0], c1 - c0), dtype=torch.float32, device=b32.device)
                self._direct.solve(b32[:, c0:c1].contiguous(), xb)
                x[:, c0:c1] = xb
        self.last_info = _DIRECT_INFO
        return x.squeeze(1) if squeeze else x

    def set_option(self, name, value):
        self._direct.set_option(name, value)

    def close(self):
        """Free the factor now instead of when the object dies (a remesh loop that keeps the old solver alive while it builds the next one
        pays for both at once)."""
        self._note_served()
        self._direct.close()

    def _note_served(self):
        key = getattr(self, "_class_key", None)
        if key is not None and self.solves_served:
            served = NestedDissectionSolver._served
            if len(served) > 64:
                served.clear()
            served[key] = self.solves_served
            self._class_key = None

    def __del__(self):
        try:
            self._note_served()
        except Exception:           # interpreter shutdown
            pass

    def info(self):
        return self._direct.info()

    def level_words(self):
        return self._direct.level_words()

    def level_rows(self):
        return self._direct.level_rows()

    def launch_profile(self):
        return self._direct.launch_profile()

    def level_index_bytes(self):
        return self._direct.level_index_bytes()

    def tier_balance(self):
        return self._direct.tier_balance()


class CholeskySolver(Solver):
    """
    Drop-in for the reference's default solver (solvers.py:26-39: cholespy / CHOLMOD factor in the constructor, two
    triangular solves per call). Same contract: the constructor factorises, `solve` is a re-solve whose result depends
    on b only.

    * matrices built by `compute_matrix` (vertex positions known): `NestedDissectionSolver` -- factor once on the
      device, then one HIP launch per upper tree level and sweep plus one per sweep for the deepest levels (9 launches at 1M vertices);
    * anything else, or a mesh whose fronts exceed that solver's limits: `IterativeCholeskySolver` (Chebyshev-Jacobi /
      Jacobi-PCG run to a residual reduction `rtol`).
    `direct=False` (or LARGESTEPS_NO_DIRECT=1) forces the iterative path; `method` says which one is in use and every
    other attribute is the chosen solver's.
    """

    def __init__(self, M, rtol=1e-6, max_iter=10000, chebyshev=True, patch_columns=3, direct=None, leaf_size=None, arity=None):
        if direct is None:
            direct = not os.environ.get("LARGESTEPS_NO_DIRECT")
        self._impl = None
        self.direct_error = None
        devices = [t for t in os.environ.get("LARGESTEPS_DEVICES", "").replace(" ", "").split(",") if t != ""]
        if direct and len(devices) > 1:
            # one process, several devices: the subtree-sharded solver behind the unchanged call sites
            try:
                from .distributed import MultiDeviceDirect
                self._impl = MultiDeviceDirect(M, devices, leaf_size=leaf_size, arity=arity)      # None: the tree ls_direct_pick_tree picks, as on one device
            except (ValueError, RuntimeError) as e:
                self.direct_error = str(e)
                warnings.warn(f"CholeskySolver: LARGESTEPS_DEVICES={','.join(devices)} could not be used ({e}); one device instead",
                              RuntimeWarning, stacklevel=2)
        if direct and self._impl is None:
            try:
                self._impl = NestedDissectionSolver(M, leaf_size=leaf_size, arity=arity)
            except (ValueError, RuntimeError) as e:      # no positions / fronts too large / numerically not SPD
                self.direct_error = str(e)
                if _native.csr_of(M).positions is not None:      # unexpected for a compute_matrix matrix: say so
                    warnings.warn(f"CholeskySolver: the direct solver is not usable for this matrix ({e}); iterating instead",
                                  RuntimeWarning, stacklevel=2)
        if self._impl is None:
            self._impl = IterativeCholeskySolver(M, rtol=rtol, max_iter=max_iter, chebyshev=chebyshev, patch_columns=patch_columns)
        self.method = "iterative" if isinstance(self._impl, IterativeCholeskySolver) else "nested-dissection"
        # the facade adds nothing to a solve: the instance attribute shadows the method below, one Python frame and one retry wrapper less
        # per call (an eager optimisation step at the reference's mesh sizes is host-bound)
        self.solve = self._impl.solve

    def solve(self, b, backward=False):                  # (documentation and the class-level contract; instances call _impl.solve directly)
        return self._impl.solve(b, backward=backward)

    @property
    def last_info(self):
        return self._impl.last_info

    def __getattr__(self, name):                     # only reached for attributes this facade does not define
        if name == "_impl":
            raise AttributeError(name)
        return getattr(self._impl, name)


class ConjugateGradientSolver(PCGSolver):
    """
    Conjugate gradients solver with the reference's stopping rule and warm start (solvers.py:41-126):
    every column iterates until ||r||_2 <= 1e-5 (absolute), starting from the previous forward /
    backward solution. Differences: Jacobi preconditioning, all columns share each matrix pass, an iteration
    cap, and no strong reference to M. For matrices with a certified spectral enclosure (built by `compute_matrix`)
    the same stopping rule is served by the Chebyshev-accelerated Jacobi iteration (a-priori count from the starting
    residual, true residual verified at the end, PCG if that check fails): same answer to the same tolerance, no dot
    products, and on large uniform meshes the LDS-resident patch kernel; `chebyshev=False` forces the PCG.
    """

    def __init__(self, M, atol=1e-5, max_iter=10000, chebyshev=True):
        super().__init__(M, rtol=0.0, atol=atol, max_iter=max_iter, warm_start=True, chebyshev=chebyshev)

    @_native.retry_on_oom
    def solve(self, b, backward=False):
        if len(b.shape) != 2:
            raise ValueError(f"Invalid array shape {b.shape} for ConjugateGradientSolver.solve: expected shape (a, b)")
        return super().solve(b, backward=backward)


class DifferentiableSolve(Function):
    """
    x = M^-1 b as an autograd node (reference: solvers.py:128-145). M is symmetric, so the gradient with respect to b is the
    same solve applied to the incoming gradient; the solver object itself gets no gradient.
    """
    @staticmethod
    def forward(ctx, solver, b):
        ctx.solver = solver
        return solver.solve(b, backward=False)

    @staticmethod
    def backward(ctx, grad_output):
        grad_b = ctx.solver.solve(grad_output.contiguous(), backward=True) if ctx.needs_input_grad[1] else None
        return None, grad_b


# functional form, as in the reference (solvers.py:148)
solve = DifferentiableSolve.apply
