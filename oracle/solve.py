"""
oracle.solve -- CPU restatement of largesteps/parameterize.py + solvers.py (TEST INFRASTRUCTURE,
see oracle/__init__.py).
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def coo_to_scipy(rows, cols, vals, V, dtype=np.float64):
    return sp.csr_matrix((np.asarray(vals).astype(dtype), (rows, cols)), shape=(V, V))


def to_differential(rows, cols, vals, v, dtype=np.float64):
    """parameterize.py:19-30: u = M @ v. fp64 by default (ground truth for the fp32 SpMV)."""
    V = v.shape[0]
    return coo_to_scipy(rows, cols, vals, V, dtype) @ np.asarray(v).astype(dtype)


class DirectSolver:
    """Stand-in for solvers.py:26-39 (CholeskySolver -> cholespy.CholeskySolverF -> CHOLMOD):
    factor once in fp64, re-solve per right-hand side. SuperLU in symmetric mode on the SPD M
    (diag_pivot_thresh=0 keeps the diagonal pivots, i.e. an LDL^T-like elimination)."""

    def __init__(self, rows, cols, vals, V):
        A = coo_to_scipy(rows, cols, vals, V, np.float64).tocsc()
        self.A = A
        self.lu = spla.splu(A, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0,
                            options=dict(SymmetricMode=True))

    def solve(self, b):
        b = np.asarray(b).astype(np.float64)
        x = self.lu.solve(b)
        return x

    def residual(self, x, b):
        return np.linalg.norm(self.A @ x - b, axis=0)


def from_differential(rows, cols, vals, u):
    """parameterize.py:32-61 with method='Cholesky': x = M^-1 u (fp64)."""
    return DirectSolver(rows, cols, vals, u.shape[0]).solve(u)


def reference_cg(rows, cols, vals, b, x0=None, tol=1e-5, max_iter=100000):
    """solvers.py:58-84 column by column (solvers.py:115-118), in fp32 like the reference:
    r = Mx - b; p = -r; while ||r|| > 1e-5: alpha = ||r||^2 / (p.Ap); x += alpha p; r += alpha Ap;
    beta = ||r'||^2/||r||^2; p = -r + beta p.   Returns (x, iterations per column).
    `max_iter` is a safety cap the reference does not have (it would spin forever)."""
    V = b.shape[0]
    M = coo_to_scipy(rows, cols, vals, V, np.float32)
    b = np.asarray(b, dtype=np.float32)
    x_out = np.zeros_like(b)
    its = []
    for ax in range(b.shape[1]):
        x = np.zeros(V, np.float32) if x0 is None else np.asarray(x0[:, ax], dtype=np.float32).copy()
        r = (M @ x - b[:, ax]).astype(np.float32)
        p = -r
        r_norm = np.float32(np.linalg.norm(r))
        k = 0
        while r_norm > tol and k < max_iter:
            Ap = (M @ p).astype(np.float32)
            r2 = np.float32(r_norm * r_norm)
            alpha = np.float32(r2 / np.float32((p * Ap).sum(dtype=np.float32)))
            x = (x + alpha * p).astype(np.float32)
            r = (r + alpha * Ap).astype(np.float32)
            r_norm = np.float32(np.linalg.norm(r))
            beta = np.float32(np.float32(r_norm * r_norm) / r2)
            p = (-r + beta * p).astype(np.float32)
            k += 1
        x_out[:, ax] = x
        its.append(k)
    return x_out, its


def jacobi_pcg(rows, cols, vals, b, x0=None, rtol=1e-6, atol=0.0, max_iter=10000, dtype=np.float64):
    """Plain statement of the algorithm the HIP solver runs (Jacobi-preconditioned CG, every
    column its own alpha/beta, stop when ||r||_2 <= max(rtol*||b||_2, atol) for every column;
    a converged column is frozen). Used to sanity-check iteration counts, not as ground truth."""
    V = b.shape[0]
    M = coo_to_scipy(rows, cols, vals, V, dtype)
    dinv = 1.0 / M.diagonal()
    b = np.asarray(b).astype(dtype)
    x = np.zeros_like(b) if x0 is None else np.asarray(x0).astype(dtype).copy()
    r = b - M @ x
    z = dinv[:, None] * r
    p = z.copy()
    rz = (r * z).sum(0)
    thr = np.maximum(rtol * np.linalg.norm(b, axis=0), atol)
    active = np.linalg.norm(r, axis=0) > thr
    it = 0
    while active.any() and it < max_iter:
        Ap = M @ p
        pAp = (p * Ap).sum(0)
        alpha = np.where(active, rz / np.where(pAp != 0, pAp, 1), 0.0)
        x += alpha * p
        r -= alpha * Ap
        z = dinv[:, None] * r
        rz_new = (r * z).sum(0)
        beta = np.where(active, rz_new / np.where(rz != 0, rz, 1), 0.0)
        p = z + beta * p
        rz = rz_new
        active = active & (np.linalg.norm(r, axis=0) > thr)
        it += 1
    return x, it
