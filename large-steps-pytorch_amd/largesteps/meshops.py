"""
Mesh bookkeeping the optimisation loop needs around a remesh, on the MI355X.

`remove_duplicates` is the reference's scripts/geometry.py:3-11 with the same name, argument order and return values
(unique vertices in the order torch.unique(dim=0) gives them, re-indexed faces, inverse map) -- one native call
(`ls_remove_duplicates`: hand-written radix sort + compaction) instead of torch.unique's library sort. There is no CPU path.
"""
import ctypes

import torch

from . import _native


def remove_duplicates(v, f):
    """
    Generate a mesh representation with no duplicates and
    return it along with the mapping to the original mesh layout.

    Returns (unique_verts (U, 3), new_faces (F, 3) int64, inverse (V,) int64) with v == unique_verts[inverse].
    """
    _native.require_device(v, "v")
    _native.require_device(f, "f")
    if v.dim() != 2 or v.shape[1] != 3:
        raise ValueError(f"v must be (V, 3), got {tuple(v.shape)}")
    if f.dim() != 2 or f.shape[1] != 3:
        raise ValueError(f"f must be (F, 3), got {tuple(f.shape)}")
    if v.dtype != torch.float32:
        raise TypeError(f"v must be float32, got {v.dtype}")
    if f.dtype not in (torch.int32, torch.int64):
        raise TypeError(f"f must be int32 or int64, got {f.dtype}")
    if f.device != v.device:
        raise RuntimeError(f"v ({v.device}) and f ({f.device}) must be on the same device")
    vc, fc = v.detach().contiguous(), f.contiguous()
    V, F, dev = vc.shape[0], fc.shape[0], vc.device
    lib = _native.lib()
    n = ctypes.c_size_t(0)
    _native.check(lib.ls_remove_duplicates_workspace_bytes(V, ctypes.byref(n)))
    ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
    unique = torch.empty((V, 3), dtype=torch.float32, device=dev)
    inverse = torch.empty(V, dtype=torch.int64, device=dev)
    new_faces = torch.empty((F, 3), dtype=torch.int64, device=dev)
    nu = ctypes.c_int64(0)
    with torch.cuda.device(dev):
        _native.check(lib.ls_remove_duplicates(_native.ptr(vc), V, _native.ptr(fc), fc.element_size(), F, _native.ptr(unique), _native.ptr(inverse),
                                               _native.ptr(new_faces), ctypes.byref(nu), _native.ptr(ws), ws.numel(), dev.index,
                                               _native.stream_of(dev)))
    return unique[: nu.value], new_faces, inverse


def reorder(v, f, matrix=None):
    """
    The mesh renumbered in the order the direct solver dissects it (vertices of one elimination-tree node contiguous, deepest level first).

    A remesher's vertex order is arbitrary; the solver works in its own order and reaches b and x through a permutation -- 12-byte rows
    gathered and scattered at random, the one part of a re-solve that moves more bytes than it needs (1.15x in the leaf launch at 1M
    vertices). A mesh handed over in THIS order makes that permutation (nearly) the identity: the same solve streams. Call it once after
    `remove_duplicates` / a remesh, before `compute_matrix`; nothing else changes (scripts/main.py:137-169 builds everything from v, f).

    Returns (v2, f2, perm): v2 = v[perm], f2 = the faces in the new numbering (int64), perm (V,) int64 -- row i of the new mesh is row
    perm[i] of the old one (take results back with x_old[perm] = x_new). matrix: a matrix `compute_matrix(v, f, ...)` already built for
    this mesh (its pattern is all that is used), else one is assembled here. Symbolic analysis only (no factorisation): ~10 ms at 1M.
    """
    from .geometry import compute_matrix
    _native.require_device(v, "v")
    _native.require_device(f, "f")
    if v.dim() != 2 or v.shape[1] != 3 or v.dtype != torch.float32:
        raise ValueError(f"v must be float32 (V, 3), got {tuple(v.shape)} {v.dtype}")
    if f.dim() != 2 or f.shape[1] != 3 or f.dtype not in (torch.int32, torch.int64):
        raise ValueError(f"f must be int32 / int64 (F, 3), got {tuple(f.shape)} {f.dtype}")
    M = matrix if matrix is not None else compute_matrix(v, f, 1.0)
    csr = _native.csr_of(M)
    if csr.V != v.shape[0]:
        raise ValueError(f"the matrix has {csr.V} rows, the mesh {v.shape[0]} vertices")
    lib, dev, V = _native.lib(), v.device, csr.V
    leaf, arity = ctypes.c_int(0), ctypes.c_int(0)
    _native.check(lib.ls_direct_pick_tree(V, ctypes.byref(leaf), ctypes.byref(arity)))
    pos = v.detach().contiguous()
    plan = ctypes.c_void_p(None)
    with torch.cuda.device(dev):
        _native.check(lib.ls_nd_plan_create_device(_native.ptr(csr.rowptr), _native.ptr(csr.col), _native.ptr(pos), V, csr.nnz, leaf.value, arity.value, 4,
                                                   dev.index, _native.stream_of(dev), ctypes.byref(plan)))
    try:
        perm32 = torch.empty(V, dtype=torch.int32)
        _native.check(lib.ls_nd_plan_arrays(plan, perm32.data_ptr(), None, None, None, None, None, None, None, None))
    finally:
        lib.ls_nd_plan_destroy(plan)
    perm = perm32.to(dev, dtype=torch.int64)
    inverse = torch.empty_like(perm)
    inverse[perm] = torch.arange(V, dtype=torch.int64, device=dev)
    return v[perm].contiguous(), inverse[f.long()].contiguous(), perm
