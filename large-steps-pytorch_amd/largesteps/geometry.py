"""
Sparse Laplacian assembly and the parameterization matrix M = I + lambda L  /  (1-alpha) I + alpha L.

API of the reference's largesteps/geometry.py (laplacian_cot :3, laplacian_uniform :65, compute_matrix :96);
the work is done by the sort-free HIP CSR assembler (csrc/assemble.hip) instead of torch.unique / sparse add /
coalesce. Returned matrices are real, coalesced `torch.sparse_coo_tensor`s (fp32 values, int64 indices in
torch's coalesce order), so everything the reference's callers do with them keeps working (`M @ v`,
`.indices()`, `.values()`, identity-keyed solver caching); the int32 CSR the kernels use rides along in a side
cache keyed by id(M) and is dropped when M is garbage collected.
"""
import ctypes

import numpy as np
import torch

from . import _native

LS_LAPLACIAN_UNIFORM, LS_LAPLACIAN_COT = 0, 1


def _assemble(verts, faces, kind, a, b):
    _native.require_device(verts, "verts")
    _native.require_device(faces, "faces")
    if faces.device != verts.device:
        raise RuntimeError(f"verts ({verts.device}) and faces ({faces.device}) must be on the same device")
    if faces.dim() != 2 or faces.shape[1] != 3:
        raise ValueError(f"faces must have shape (F, 3), got {tuple(faces.shape)}")
    if faces.dtype not in (torch.int32, torch.int64):
        raise TypeError(f"faces must be int32 or int64, got {faces.dtype}")
    if verts.dim() != 2:
        raise ValueError(f"verts must have shape (V, 3), got {tuple(verts.shape)}")
    V, F = verts.shape[0], faces.shape[0]
    dev = verts.device
    faces_c = faces.contiguous()
    verts_c = None
    if kind == LS_LAPLACIAN_COT:
        if verts.shape[1] != 3:
            raise ValueError(f"verts must have shape (V, 3), got {tuple(verts.shape)}")
        verts_c = verts.detach().to(torch.float32).contiguous()
    lib = _native.lib()
    nbytes = ctypes.c_size_t(0)
    _native.check(lib.ls_assemble_workspace_bytes(V, F, ctypes.byref(nbytes)))
    with torch.cuda.device(dev):
        ws = torch.empty(max(nbytes.value, 1), dtype=torch.uint8, device=dev)
        rowptr = torch.empty(V + 1, dtype=torch.int32, device=dev)
        nnz = ctypes.c_int64(0)
        st = _native.stream_of(dev)
        _native.check(lib.ls_assemble_pattern(_native.ptr(faces_c), faces_c.element_size(), F, V, _native.ptr(verts_c), kind,
                                              float(a), float(b), _native.ptr(ws), ws.numel(), _native.ptr(rowptr),
                                              ctypes.byref(nnz), dev.index, st))
        n = nnz.value
        col = torch.empty(n, dtype=torch.int32, device=dev)
        val = torch.empty(n, dtype=torch.float32, device=dev)
        idx = torch.empty((2, n), dtype=torch.int64, device=dev)
        _native.check(lib.ls_assemble_fill(_native.ptr(ws), ws.numel(), V, F, _native.ptr(rowptr), _native.ptr(col),
                                           _native.ptr(val), _native.ptr(idx), n, None, dev.index, st))
    return V, rowptr, col, val, idx


def _wrap(V, rowptr, col, val, idx, a_min=None, uniform=None, positions=None):
    M = torch.sparse_coo_tensor(idx, val, (V, V), is_coalesced=True)
    _native.register_csr(M, _native.CsrMatrix(V, rowptr, col, M._values(), symmetric=True, a_min=a_min, uniform=uniform,
                                              positions=positions))
    return M


@_native.retry_on_oom
def laplacian_uniform(verts, faces):
    """
    Compute the uniform laplacian  L = D - A  (reference: geometry.py:65-94).

    Parameters
    ----------
    verts : torch.Tensor
        Vertex positions (only the vertex count is used).
    faces : torch.Tensor
        array of triangle faces (int32 or int64).
    """
    V, rowptr, col, val, idx = _assemble(verts, faces, LS_LAPLACIAN_UNIFORM, 0.0, 1.0)
    # The assembler always emits the diagonal; the reference has no entry at all in the row of a vertex that
    # no face references (geometry.py:82-94). Drop those explicit zeros (rare path, plain index arithmetic).
    lonely = (rowptr[1:] - rowptr[:-1]) == 1
    if bool(lonely.any()):
        keep = ~lonely[idx[0]]
        idx, val = idx[:, keep].contiguous(), val[keep].contiguous()
        return torch.sparse_coo_tensor(idx, val, (V, V), is_coalesced=True)   # CSR side car rebuilt on first use
    return _wrap(V, rowptr, col, val, idx)


@_native.retry_on_oom
def laplacian_cot(verts, faces):
    """
    Compute the cotangent laplacian (reference: geometry.py:3-63; weights cot a + cot b, no 1/2 factor).

    The reference returns this matrix uncoalesced; here it is returned coalesced (same values after
    `.coalesce()`), with an explicit diagonal entry for every vertex as in the reference.
    """
    return _wrap(*_assemble(verts, faces, LS_LAPLACIAN_COT, 0.0, 1.0))


@_native.retry_on_oom
def compute_matrix(verts, faces, lambda_, alpha=None, cotan=False):
    """
    Build the parameterization matrix (reference: geometry.py:96-133).

    If alpha is defined, then we compute it as (1-alpha)*I + alpha*L otherwise
    as I + lambda*L as in the paper.

    Parameters
    ----------
    verts : torch.Tensor
        Vertex positions
    faces : torch.Tensor
        Triangle faces
    lambda_ : float
        Hyperparameter lambda of the method: M = I + lambda_ * L
    alpha : float in [0, 1[
        Alternative hyperparameter: M = (1-alpha) * I + alpha * L  (lambda_ is then ignored)
    cotan : bool
        Compute the cotangent laplacian. Otherwise, compute the combinatorial one
    """
    if alpha is None:
        a, b = 1.0, float(lambda_)
    else:
        if alpha < 0.0 or alpha >= 1.0:
            raise ValueError(f"Invalid value for alpha: {alpha} : it should take values between 0 (included) and 1 (excluded)")
        a, b = 1.0 - float(alpha), float(alpha)
    # python doubles become fp32 scalars exactly where torch rounds them
    a32, b32 = float(np.float32(a)), float(np.float32(b))
    kind = LS_LAPLACIAN_COT if cotan else LS_LAPLACIAN_UNIFORM
    # M = a I + b L with L positive semi-definite (graph Laplacian: always; cotangent stiffness matrix: as a
    # quadratic form) and b >= 0  =>  lambda_min(M) >= a: the enclosure the Chebyshev solver needs.
    a_min = a32 if (a32 > 0.0 and b32 >= 0.0) else None
    positions = verts.detach() if (verts.dim() == 2 and verts.shape[1] == 3) else None
    return _wrap(*_assemble(verts, faces, kind, a32, b32), a_min=a_min, uniform=None if cotan else (a32, b32), positions=positions)
