"""
The differential parameterization (reference: largesteps/parameterize.py:5-61), same function names, arguments and caching
behaviour, on the MI355X:  u = L v  is one LDS-staged CSR SpMV (csrc/spmv.hip),  v = L^-1 u  goes through a solver object that
is built on first use per (matrix, method) and lives exactly as long as the matrix does.
"""
import weakref

import torch

from . import _native
from .solvers import CholeskySolver, ConjugateGradientSolver, solve

# (id(L), method) -> (solver, weak reference to L). The key is the matrix OBJECT, as in the reference (parameterize.py:5-17);
# the weak reference's callback removes the entry when the matrix dies. Unlike the reference's 'CG' solver, nothing stored here
# holds a strong reference to L, so the entry -- and the native handle behind it -- really goes away with the matrix.
_cache = {}


def cache_put(key, value, A):
    def forget(_ref, key=key):          # runs when A is garbage collected
        _cache.pop(key, None)

    _cache[key] = (value, weakref.ref(A, forget))


class _SpMV(torch.autograd.Function):
    """u = L v with the HIP CSR SpMV; the gradient with respect to v is L^T g: the same kernel on the transposed side car
    (identical to L's for the symmetric matrices compute_matrix builds)."""

    @staticmethod
    def forward(ctx, csr, v):
        ctx.csr = csr
        return _native.spmv(csr, v)

    @staticmethod
    def backward(ctx, g):
        if not ctx.needs_input_grad[1]:
            return None, None
        csr = ctx.csr
        # a foreign matrix: find out once whether L^T = L entry for entry (a matrix that is symmetric only up to rounding gets its
        # transpose applied -- the gradient is exact either way)
        return None, _native.spmv(csr if _native.is_symmetric(csr, exact=True) else _native.csr_transposed(csr), g.contiguous())


def to_differential(L, v):
    """
    Differential coordinates of v:  u = L @ v.

    Parameters
    ----------
    L : torch.sparse.Tensor
        The system matrix (I + lambda * Laplacian), as returned by `compute_matrix`
    v : torch.Tensor
        Vertex coordinates, (V, k) or (V,), float32 on the matrix's device
    """
    _native.require_device(v, "v")
    csr = _native.csr_of(L)
    if v.device != csr.device:
        raise RuntimeError(f"matrix ({csr.device}) and v ({v.device}) must be on the same device")
    if v.dtype != torch.float32:
        raise TypeError(f"v must be float32, got {v.dtype}")
    if v.shape[0] != csr.V or v.dim() not in (1, 2):
        raise ValueError(f"v has shape {tuple(v.shape)}, expected ({csr.V}, k)")
    squeeze = v.dim() == 1
    v2 = (v.unsqueeze(1) if squeeze else v).contiguous()
    if v2.shape[1] == 0 or v2.shape[1] > 64:
        raise ValueError(f"to_differential supports 1..64 columns, got {v2.shape[1]}")
    u = _SpMV.apply(csr, v2)
    return u.squeeze(1) if squeeze else u


def from_differential(L, u, method='Cholesky'):
    """
    Vertex coordinates from differential coordinates:  the solution v of  L v = u.

    The solver for (L, method) is constructed on the first call and reused afterwards; it is dropped when L is garbage
    collected.

    Parameters
    ----------
    L : torch.sparse.Tensor
        The system matrix (I + lambda * Laplacian)
    u : torch.Tensor
        Differential coordinates
    method : {'Cholesky', 'CG'}
        'Cholesky' (default): factor once / re-solve -- the nested-dissection direct solver (the constructor factorises,
        every call is a re-solve whose result depends on u only); matrices it cannot factorise fall back to a cold-started
        Chebyshev / PCG iteration run to a residual reduction of 1e-6. 'CG': the reference's conjugate-gradient contract
        (stop at ||r||_2 <= 1e-5, warm start from the previous forward / backward solution) on the HIP iteration kernels.
    """
    key = (id(L), method)
    hit = _cache.get(key)
    if hit is None:
        if method == 'Cholesky':
            solver = CholeskySolver(L)
        elif method == 'CG':
            solver = ConjugateGradientSolver(L)
        else:
            raise ValueError(f"Unknown solver type '{method}'.")
        cache_put(key, solver, L)
    else:
        solver = hit[0]
    return solve(solver, u)
