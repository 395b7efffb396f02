"""
to_differential / from_differential with the reference's API and cache semantics
(reference: largesteps/parameterize.py:5-61).
"""
import weakref

import torch

from . import _native
from .solvers import CholeskySolver, ConjugateGradientSolver, solve

# Cache for the system solvers: (id(L), method) -> (solver, weakref(L)); same keying and eviction as the
# reference (parameterize.py:5-17). No solver of this package keeps a strong reference to L, so the entry
# (and the native handle) really goes away with the matrix -- the reference's 'CG' entry never did.
_cache = {}


def cache_put(key, value, A):
    # Called when 'A' is garbage collected
    def cleanup_callback(wr):
        _cache.pop(key, None)

    wr = weakref.ref(A, cleanup_callback)
    _cache[key] = (value, wr)


class _SpMV(torch.autograd.Function):
    """u = L v with the HIP CSR SpMV; d/dv = L^T g."""

    @staticmethod
    def forward(ctx, L, csr, v):
        ctx.L, ctx.csr = L, csr
        return _native.spmv(csr, v)

    @staticmethod
    def backward(ctx, g):
        if not ctx.needs_input_grad[2]:
            return None, None, None
        g = g.contiguous()
        if ctx.csr.symmetric:
            return None, None, _native.spmv(ctx.csr, g)
        return None, None, torch.sparse.mm(ctx.L.t(), g)   # foreign, possibly unsymmetric matrix


def to_differential(L, v):
    """
    Convert vertex coordinates to the differential parameterization:  u = L @ v.

    Parameters
    ----------
    L : torch.sparse.Tensor
        (I + l*L) matrix
    v : torch.Tensor
        Vertex coordinates
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
    u = _SpMV.apply(L, csr, v2)
    return u.squeeze(1) if squeeze else u


def from_differential(L, u, method='Cholesky'):
    """
    Convert differential coordinates back to Cartesian:  solve L v = u.

    If this is the first time we call this function on a given matrix L, the
    solver is cached. It will be destroyed once the matrix is garbage collected.

    Parameters
    ----------
    L : torch.sparse.Tensor
        (I + l*L) matrix
    u : torch.Tensor
        Differential coordinates
    method : {'Cholesky', 'CG'}
        Solver to use. Both run the HIP Jacobi-PCG: 'Cholesky' to a relative residual of 1e-6 from a cold
        start (the accuracy class of the reference's fp32 Cholesky solve), 'CG' with the reference's own
        stopping rule (||r|| <= 1e-5, warm-started from the previous solution).
    """
    key = (id(L), method)
    if key not in _cache.keys():
        if method == 'Cholesky':
            solver = CholeskySolver(L)
        elif method == 'CG':
            solver = ConjugateGradientSolver(L)
        else:
            raise ValueError(f"Unknown solver type '{method}'.")

        cache_put(key, solver, L)
    else:
        solver = _cache[key][0]

    return solve(solver, u)
