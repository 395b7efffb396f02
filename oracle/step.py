"""
CPU restatement (numpy, fp64) of one whole optimisation step the way the reference chains the path's pieces
(scripts/main.py:172-208 without the renderer). TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

    v = from_differential(M, u, 'Cholesky')                       largesteps/parameterize.py:32-61, solvers.py:26-39
    n = compute_vertex_normals(v, f, compute_face_normals(v, f))  scripts/geometry.py:91-147
    loss = mean((v - v*)^2) + mean((n - n*)^2) + reg * mean((L v)^2)             (regulariser: main.py:193)
    u.grad = M^-1 dloss/dv                                         solvers.py:134-145 (backward = the same solve, M symmetric)
    AdamUniform.step                                               largesteps/optimize.py:18-41
Pinned against tests/golden/reference_step.npz (tests/golden/make_golden_step.py executes the reference itself) in
tests/test_oracle.py.
"""
import numpy as np

from . import laplacian, normals, solve


class AdamUniform:
    """optimize.py:18-41: m, v EMA; bias correction; p -= lr * m_hat / (1e-8 + max sqrt(v_hat))"""

    def __init__(self, lr=0.1, betas=(0.9, 0.999)):
        self.lr, self.b1, self.b2 = lr, betas[0], betas[1]
        self.t, self.g1, self.g2 = 0, None, None

    def step(self, p, grad):
        if self.g1 is None:
            self.g1, self.g2 = np.zeros_like(p), np.zeros_like(p)
        self.t += 1
        self.g1 = self.b1 * self.g1 + (1 - self.b1) * grad
        self.g2 = self.b2 * self.g2 + (1 - self.b2) * grad * grad
        m1 = self.g1 / (1 - self.b1 ** self.t)
        m2 = self.g2 / (1 - self.b2 ** self.t)
        return p - self.lr * m1 / (1e-8 + np.sqrt(m2).max())


def run(v, f, lambda_, alpha, cotan, target_v, target_n, steps, lr, reg):
    """Returns (u after every step, v of every step, loss of every step), all fp64."""
    rows, cols, vals = laplacian.compute_matrix(v, f, lambda_, alpha=alpha, cotan=cotan)
    V = v.shape[0]
    lr_, lc_, lv_ = laplacian.uniform_laplacian(V, f)
    L = solve.coo_to_scipy(lr_, lc_, lv_, V)
    direct = solve.DirectSolver(rows, cols, vals, V)
    u = solve.to_differential(rows, cols, vals, v).astype(np.float64)
    opt = AdamUniform(lr)
    us, vs, losses = [], [], []
    n_el = float(V * 3)
    for _ in range(steps):
        x = direct.solve(u)
        fn = normals.face_normals(x, f)
        n = normals.vertex_normals(x, f, fn)
        Lx = L @ x
        losses.append(float(((x - target_v) ** 2).mean() + ((n - target_n) ** 2).mean() + reg * (Lx ** 2).mean()))
        g_n = 2.0 * (n - target_n) / n_el
        g_x, g_fn = normals.vertex_normals_backward(x, f, fn, g_n)
        g_x = g_x + normals.face_normals_backward(x, f, g_fn) + 2.0 * (x - target_v) / n_el + reg * 2.0 * (L.T @ Lx) / n_el
        g_u = direct.solve(g_x)
        u = opt.step(u, g_u)
        us.append(u.copy()); vs.append(x.copy())
    return np.stack(us), np.stack(vs), np.array(losses)
