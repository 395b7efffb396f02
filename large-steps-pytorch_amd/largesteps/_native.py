"""
ctypes binding of liblargesteps_hip.so (C ABI: include/largesteps_hip.h).

This is the only module that touches the native library. There is NO fallback: if the shared library is
missing, or a tensor is not on a HIP device, the call fails loudly.

Tensors cross the boundary as raw device pointers + sizes + (device ordinal, hipStream_t). The stream is
always torch's current stream of the tensor's device *on the calling thread* -- autograd runs backward()
on its own worker thread (reference: largesteps/solvers.py:139-145), so nothing is cached per thread.
"""
import ctypes
import os
import threading
import weakref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("LARGESTEPS_HIP_LIB", os.path.join(_HERE, "..", "lib", "liblargesteps_hip.so"))

LS_E_INVALID, LS_E_INDEX, LS_E_WORKSPACE, LS_E_OVERFLOW, LS_E_STATE, LS_E_NOT_CONVERGED = -1, -2, -3, -4, -5, -6

_lib = None
_lib_lock = threading.Lock()

c_void_p, c_int, c_i64, c_float, c_double, c_size_t = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float,
                                                        ctypes.c_double, ctypes.c_size_t)


class SolveInfo(ctypes.Structure):
    _fields_ = [("iterations", ctypes.c_int32), ("converged", ctypes.c_int32),
                ("rnorm", ctypes.c_double * 4), ("bnorm", ctypes.c_double * 4)]


class DirectOptions(ctypes.Structure):
    """ls_direct_options of include/largesteps_hip.h (filled by ls_direct_options_default)"""
    _fields_ = [("struct_bytes", ctypes.c_int32), ("leaf_size", ctypes.c_int32), ("arity", ctypes.c_int32), ("tier_levels", ctypes.c_int32),
                ("sparse_leaves", ctypes.c_int32), ("shard_rank", ctypes.c_int32), ("shard_count", ctypes.c_int32), ("ordering", ctypes.c_int32),
                ("tier_waves", ctypes.c_int32)]


_SIGNATURES = {
    "ls_version": (c_int, []),
    "ls_last_error": (ctypes.c_char_p, []),
    "ls_assemble_workspace_bytes": (c_int, [c_i64, c_i64, ctypes.POINTER(c_size_t)]),
    "ls_assemble_pattern": (c_int, [c_void_p, c_int, c_i64, c_i64, c_void_p, c_int, c_float, c_float, c_void_p, c_size_t,
                                    c_void_p, ctypes.POINTER(c_i64), c_int, c_void_p]),
    "ls_assemble_fill": (c_int, [c_void_p, c_size_t, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p,
                                 c_int, c_void_p]),
    "ls_csr_from_coo": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                c_int, c_void_p]),
    "ls_spmv": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ls_solver_create": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_int, c_int, c_void_p, ctypes.POINTER(c_void_p)]),
    "ls_solver_create_ext": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_int, c_int, c_void_p,
                                     ctypes.POINTER(c_void_p)]),
    "ls_solver_phase": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_double, c_double, c_int, c_void_p]),
    "ls_solver_buffers": (c_int, [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_int),
                                  ctypes.POINTER(c_int)]),
    "ls_shard_cheb_steps": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_i64, c_void_p]),
    "ls_shard_resnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_void_p]),
    "ls_solver_bind": (c_int, [c_void_p, c_void_p, c_void_p]),
    "ls_solver_poll": (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(SolveInfo), c_void_p]),
    "ls_gather_rows": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_int, c_void_p]),
    "ls_solver_destroy": (c_int, [c_void_p]),
    "ls_direct_create": (c_int, [c_void_p, c_int, c_void_p, ctypes.POINTER(c_void_p)]),
    "ls_direct_destroy": (c_int, [c_void_p]),
    "ls_direct_solve": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ls_direct_set": (c_int, [c_void_p, ctypes.c_char_p, c_int]),
    "ls_remove_duplicates_workspace_bytes": (c_int, [c_i64, ctypes.POINTER(c_size_t)]),
    "ls_remove_duplicates": (c_int, [c_void_p, c_i64, c_void_p, c_int, c_i64, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_i64), c_void_p,
                                     c_size_t, c_int, c_void_p]),
    "ls_csr_transpose_workspace_bytes": (c_int, [c_i64, c_i64, ctypes.POINTER(c_size_t)]),
    "ls_csr_transpose": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "ls_corner_ranks_workspace_bytes": (c_int, [c_i64, c_i64, ctypes.POINTER(c_size_t)]),
    "ls_corner_ranks": (c_int, [c_void_p, c_int, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "ls_normals_workspace_bytes": (c_int, [c_i64, c_i64, ctypes.POINTER(c_size_t)]),
    "ls_face_normals": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_i64, c_void_p, c_int, c_void_p]),
    "ls_face_normals_backward": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_size_t, c_int, c_void_p]),
    "ls_vertex_normals": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_size_t, c_int, c_void_p]),
    "ls_vertex_normals_backward": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "ls_face_normals_with_norms": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "ls_vertex_normals_from_norms": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_size_t, c_int, c_void_p]),
    "ls_normals_pair_backward_faces": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "ls_normals_pair_backward_verts": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "ls_vertex_normals_gathered": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ls_shard_plan_create": (c_int, [c_i64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "ls_shard_plan_destroy": (c_int, [c_void_p]),
    "ls_shard_plan_info": (c_int, [c_void_p] + [ctypes.POINTER(c_i64)] * 5 + [ctypes.POINTER(c_int)] * 2 + [ctypes.POINTER(c_i64)]),
    "ls_shard_plan_arrays": (c_int, [c_void_p] + [c_void_p] * 8),
    "ls_shard_layer_sizes": (c_int, [c_i64, c_void_p, c_void_p, c_i64, c_i64, c_int, c_void_p]),
    "ls_patch_plan_create": (c_int, [c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "ls_patch_plan_destroy": (c_int, [c_void_p]),
    "ls_patch_plan_info": (c_int, [c_void_p] + [ctypes.POINTER(c_int)] * 5 + [ctypes.POINTER(c_i64)] * 3 + [ctypes.POINTER(ctypes.c_double)]),
    "ls_patch_plan_arrays": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ls_direct_level_words": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "ls_direct_tier_lds_bytes": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_size_t)]),
    "ls_direct_level_index_bytes": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "ls_direct_tier_balance": (c_int, [c_void_p, ctypes.POINTER(c_double * 4)]),
    "ls_direct_level_rows": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "ls_direct_exchange_region": (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_i64)]),
    "ls_dist_unique_id": (c_int, [c_void_p]),
    "ls_dist_create": (c_int, [c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "ls_dist_destroy": (c_int, [c_void_p]),
    "ls_dist_allreduce_sum": (c_int, [c_void_p, c_void_p, c_i64, c_void_p]),
    "ls_dist_info": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "ls_dist_direct_solve": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ls_direct_launch_profile": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ls_direct_factor": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_void_p, ctypes.POINTER(c_void_p)]),
    "ls_direct_factor_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_int, c_void_p, ctypes.POINTER(c_void_p)]),
    "ls_direct_options_default": (c_int, [c_void_p]),
    "ls_direct_solve_part": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ls_direct_shard_info": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_i64),
                                     c_void_p]),
    "ls_direct_shape": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "ls_direct_factor_seconds": (c_int, [c_void_p, ctypes.POINTER(c_double * 3)]),
    "ls_csr_is_symmetric": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_float, ctypes.POINTER(c_int), c_int, c_void_p]),
    "ls_direct_pick_tree": (c_int, [c_i64, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "ls_release_scratch": (c_int, [c_int]),
    "ls_nd_plan_create": (c_int, [c_i64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "ls_nd_plan_create_device": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_int, c_int, c_int, c_int, c_void_p, ctypes.POINTER(c_void_p)]),
    "ls_nd_plan_create_ordered": (c_int, [c_i64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "ls_nd_plan_create_device_ordered": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_int, c_int, c_int, c_int, c_int, c_void_p, ctypes.POINTER(c_void_p)]),
    "ls_nd_plan_quality": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_double), ctypes.POINTER(c_double), ctypes.POINTER(c_double)]),
    "ls_direct_plan_quality": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_double), ctypes.POINTER(c_double), ctypes.POINTER(c_double)]),
    "ls_nd_plan_destroy": (c_int, [c_void_p]),
    "ls_nd_plan_info": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_i64),
                                ctypes.POINTER(c_i64), ctypes.POINTER(c_double)]),
    "ls_nd_plan_arrays": (c_int, [c_void_p] + [c_void_p] * 9),
    "ls_direct_info": (c_int, [c_void_p, ctypes.POINTER(c_i64), ctypes.POINTER(c_int), ctypes.POINTER(c_double * 3)]),
    "ls_solver_solve": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_double, c_double, c_int,
                                ctypes.POINTER(SolveInfo), c_void_p]),
    "ls_solver_set": (c_int, [c_void_p, ctypes.c_char_p, c_int]),
    "ls_solver_set_spectrum": (c_int, [c_void_p, c_double]),
    "ls_solver_chebyshev_iterations": (c_int, [c_void_p, c_double, ctypes.POINTER(c_int)]),
    "ls_solver_set_uniform": (c_int, [c_void_p, c_float, c_float, c_void_p]),
    "ls_solver_set_patches": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_int,
                                      c_int, c_int, c_void_p]),
    "ls_solver_spectrum": (c_int, [c_void_p, ctypes.POINTER(c_double), ctypes.POINTER(c_double)]),
    "ls_solver_solve_chebyshev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_double, c_double, c_int,
                                          ctypes.POINTER(SolveInfo), c_void_p]),
    "ls_solver_profile": (c_int, [c_void_p, ctypes.POINTER(c_double * 3), ctypes.POINTER(c_int)]),
    "ls_solver_sell": (c_int, [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_i64)]),
    "ls_solver_workspace_bytes": (c_int, [c_void_p, ctypes.POINTER(c_size_t)]),
    "ls_adam_uniform_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_float, c_float, c_float, c_int,
                                     c_void_p, c_int, c_void_p]),
    "ls_adam_uniform_step_device": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_float, c_float, c_float, c_void_p,
                                            c_void_p, c_int, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib_path():
    return os.path.abspath(_LIB_PATH)


def lib():
    """Load liblargesteps_hip.so once. Raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        with _lib_lock:
            if _lib is None:
                path = lib_path()
                if not os.path.exists(path):
                    raise RuntimeError(
                        f"largesteps: native library not found at {path}. Build it with "
                        f"`make -C {os.path.join(_HERE, '..', 'csrc')}` (hipcc, gfx950) or `python -c "
                        f"'import __graft_entry__ as g; g.build()'` from the repository root. There is no CPU/PyTorch fallback.")
                handle = ctypes.CDLL(path)
                for name, (res, args) in _SIGNATURES.items():
                    fn = getattr(handle, name)   # AttributeError if the ABI drifted
                    fn.restype = res
                    fn.argtypes = args
                _lib = handle
    return _lib


class NotConverged(RuntimeError):
    pass


def last_error():
    msg = lib().ls_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    """Map a C-ABI status to the Python exception the reference's code path would raise."""
    if rc == 0:
        return
    msg = last_error()
    if rc == LS_E_INVALID:
        raise ValueError(msg)
    if rc == LS_E_INDEX:
        raise IndexError(msg)
    if rc == LS_E_OVERFLOW:
        raise OverflowError(msg)
    if rc == LS_E_NOT_CONVERGED:
        raise NotConverged(msg)
    raise RuntimeError(f"largesteps native error {rc}: {msg}")


def require_device(t, what):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{what} must be a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"largesteps (MI355X build): {what} must live on a HIP device ('cuda'), got device '{t.device}'. "
                           "There is no CPU path in this package.")
    return t


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # the handle without building a torch.cuda.Stream object (~5 us per call)


_oom_state = threading.local()


def retry_on_oom(fn):
    """Decorator of the package's allocating entry points (compute_matrix, the solvers' solve, the normals): the direct solver keeps up
    to LS_POOL_GB of freed device buffers in a pool that torch's caching allocator cannot see (ls_release_scratch in the header). When
    torch runs out of memory inside such a call the pool is emptied, torch's cache too, and the call is repeated ONCE; a second failure
    is the caller's. Decorated calls nest (CholeskySolver.solve -> NestedDissectionSolver.solve): only the OUTERMOST one on a thread
    retries, so one failure costs one repetition, and nothing is repeated while a stream capture is in progress (emptying the caches
    would invalidate the capture: the error goes to the caller as it is)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        if getattr(_oom_state, "inside", False):
            return fn(*args, **kwargs)
        _oom_state.inside = True
        try:
            try:
                return fn(*args, **kwargs)
            except torch.cuda.OutOfMemoryError:
                if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                    raise
                check(lib().ls_release_scratch(-1))
                torch.cuda.empty_cache()
                return fn(*args, **kwargs)
        finally:
            _oom_state.inside = False
    return wrapped


def stream_of(device):
    if _raw_stream is not None and device.index is not None:
        return ctypes.c_void_p(_raw_stream(device.index))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def raw_stream(device):
    """the current stream of `device` as a plain integer (ctypes converts it for a c_void_p argument)"""
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t):
    """Raw device pointer of a tensor (NULL for None / empty tensors)."""
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


# ---------------------------------------------------------------------------------------------------
# CSR side car of a torch sparse COO matrix
# ---------------------------------------------------------------------------------------------------
class CsrMatrix:
    """int32 CSR (rowptr, col) + fp32 val of a (V,V) matrix on one HIP device. `val` is the very tensor
    that backs M.values() (same order: row-major sorted COO == CSR order), so no value copy exists."""
    __slots__ = ("V", "nnz", "rowptr", "col", "val", "device", "symmetric", "exact_symmetric", "a_min", "uniform", "positions", "_transposed",
                 "_transposed_version", "__weakref__")

    def __init__(self, V, rowptr, col, val, symmetric, a_min=None, uniform=None, positions=None):
        self.V, self.nnz = int(V), int(col.shape[0])
        self.rowptr, self.col, self.val = rowptr, col, val
        self.device = val.device
        # two different facts (None: not determined yet, see is_symmetric):
        #   symmetric        M = M^T up to 1e-6 max|M_ij| -- what the factorisation needs
        #   exact_symmetric  M = M^T entry for entry -- what lets a backward pass apply M instead of M^T
        self.symmetric = symmetric
        self.exact_symmetric = symmetric
        # certified lower bound of lambda_min(M): M = a I + b L with L positive semi-definite => a (None: unknown)
        self.a_min = a_min
        # (a, b) if M = a I + b L_uniform (all off-diagonal values equal -b): lets the solver drop the value array
        self.uniform = uniform
        # (V,3) vertex positions the matrix was assembled from (detached; only their spatial order is used, to cut
        # the mesh into compact patches for the LDS-resident solver kernel)
        self.positions = positions
        self._transposed = None
        self._transposed_version = None


def is_symmetric(csr, exact=False):
    """M = M^T? exact: entry for entry (tolerance 0); otherwise up to 1e-6 of the largest entry. Determined once per side
    car and tolerance by a native kernel (ls_csr_is_symmetric); matrices built by compute_matrix carry the answer."""
    have = csr.exact_symmetric if exact else csr.symmetric
    if have is not None:
        return have
    if not exact and csr.exact_symmetric:
        csr.symmetric = True
        return True
    ok = ctypes.c_int(0)
    tol = 0.0 if exact else (1e-6 * float(csr.val.abs().max()) if csr.nnz else 0.0)
    with torch.cuda.device(csr.device):
        check(lib().ls_csr_is_symmetric(ptr(csr.rowptr), ptr(csr.col), ptr(csr.val), csr.V, csr.nnz, tol, ctypes.byref(ok), csr.device.index,
                                        stream_of(csr.device)))
    if exact:
        csr.exact_symmetric = bool(ok.value)
        if ok.value:
            csr.symmetric = True
    else:
        csr.symmetric = bool(ok.value)
        if not ok.value:
            csr.exact_symmetric = False
    return bool(ok.value)


# (id(M)) -> (CsrMatrix, weakref(M)). Mirrors the reference's solver cache (parameterize.py:5-17): keyed by
# object identity, dropped by a weakref callback when M is garbage collected, never holds M itself.
_csr_cache = {}
_csr_lock = threading.Lock()


def register_csr(M, csr):
    key = id(M)

    def _drop(_wr, key=key):
        with _csr_lock:
            _csr_cache.pop(key, None)

    with _csr_lock:
        _csr_cache[key] = (csr, weakref.ref(M, _drop))


def csr_of(M):
    """CSR side car of M: the one built by compute_matrix, or (foreign matrix) converted on first use."""
    key = id(M)
    with _csr_lock:
        hit = _csr_cache.get(key)
    if hit is not None and hit[1]() is M:
        return hit[0]
    csr = csr_from_coo(M)
    register_csr(M, csr)
    return csr


def csr_from_coo(M):
    if not isinstance(M, torch.Tensor) or M.layout != torch.sparse_coo:
        raise TypeError("expected a torch sparse COO matrix (as returned by largesteps.geometry.compute_matrix)")
    require_device(M, "the system matrix")
    if M.dim() != 2 or M.shape[0] != M.shape[1]:
        raise ValueError(f"expected a square sparse matrix, got shape {tuple(M.shape)}")
    if M.dtype != torch.float32:
        raise TypeError(f"expected a float32 matrix, got {M.dtype}")
    if not M.is_coalesced():
        raise ValueError("the system matrix must be coalesced (call .coalesce(); the reference's solvers need it too)")
    V = M.shape[0]
    idx = M.indices()
    rows, cols = idx[0].contiguous(), idx[1].contiguous()
    val = M.values().contiguous()
    nnz = val.shape[0]
    dev = M.device
    rowptr = torch.empty(V + 1, dtype=torch.int32, device=dev)
    col = torch.empty(nnz, dtype=torch.int32, device=dev)
    scratch = torch.empty(4 * (V + 1) + 4 * ((V + 1) // 2048 + 4) + 1024, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib().ls_csr_from_coo(ptr(rows), ptr(cols), ptr(val), nnz, V, ptr(rowptr), ptr(col), None, ptr(scratch),
                                    scratch.numel(), dev.index, stream_of(dev)))
    return CsrMatrix(V, rowptr, col, val, symmetric=None)


def spmv(csr, x, variant=0):
    """y = M x for x of shape (V,k), float32, contiguous."""
    y = torch.empty_like(x)
    dev = csr.device
    check(lib().ls_spmv(ptr(csr.rowptr), ptr(csr.col), ptr(csr.val), csr.V, csr.nnz, ptr(x), ptr(y), x.shape[1], variant,
                        dev.index, stream_of(dev)))
    return y


def csr_transposed(csr):
    """CSR side car of M^T (cached on the side car of M): native radix sort of the entries by column, no torch sort."""
    t = getattr(csr, "_transposed", None)
    if t is not None and csr._transposed_version == csr.val._version:     # (the values may have been updated in place since)
        return t
    dev = csr.device
    n = c_size_t(0)
    check(lib().ls_csr_transpose_workspace_bytes(csr.V, csr.nnz, ctypes.byref(n)))
    ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
    rowptr = torch.empty(csr.V + 1, dtype=torch.int32, device=dev)
    col = torch.empty(max(csr.nnz, 1), dtype=torch.int32, device=dev)[: csr.nnz]
    val = torch.empty(max(csr.nnz, 1), dtype=torch.float32, device=dev)[: csr.nnz]
    with torch.cuda.device(dev):
        check(lib().ls_csr_transpose(ptr(csr.rowptr), ptr(csr.col), ptr(csr.val), csr.V, csr.nnz, ptr(rowptr), ptr(col), ptr(val), ptr(ws),
                                     ws.numel(), dev.index, stream_of(dev)))
    t = CsrMatrix(csr.V, rowptr, col, val, symmetric=csr.symmetric)
    t.exact_symmetric = csr.exact_symmetric
    csr._transposed = t
    csr._transposed_version = csr.val._version
    return t
