"""
Host-side patch plan of the LDS-resident s-step Chebyshev kernel, CPU only: the NATIVE analysis (csrc/patch_plan.cpp behind
ls_patch_plan_*, what largesteps.patches.PatchPlan.build calls) and its numpy statement (tests/patch_plan_statement.py) --
structure invariants, and the numpy statement of the kernel: s steps on overlapping patches must reproduce s global steps
exactly on every owned vertex.
"""
import math

import numpy as np
import pytest
import scipy.sparse as sp

from largesteps import synthetic
from largesteps.patches import PatchPlan
import patch_plan_statement as pps
from statements import patch_steps
from oracle import laplacian as ol


def _system(name):
    if name == "plane":
        v, f = synthetic.plane(90)
        lam = 40.0
    else:
        v, f = synthetic.icosphere(24)
        v = synthetic.perturb(v, radial=0.05, seed=1)
        lam = 19.0
    r, c, val = ol.compute_matrix(v, f, lam)
    V = v.shape[0]
    rp = np.zeros(V + 1, np.int64)
    np.add.at(rp, r + 1, 1)
    rp = np.cumsum(rp)
    A = sp.csr_matrix((val.astype(np.float64), c, rp), shape=(V, V))
    return v, lam, rp, c, A


def _schedule(A, a_min, reduction):
    d = A.diagonal()
    lmax = (abs(A).sum(axis=1).A1 / d).max() * (1 + 1e-5)
    lmin = 0.98 * a_min / d.max()
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma1 = theta / delta
    sk = math.sqrt(lmax / lmin)
    n = int(math.ceil(math.log(2 / reduction) / -math.log((sk - 1) / (sk + 1))))
    c1, c2, rho = [], [], 1 / sigma1
    for it in range(n):
        if it == 0:
            c1.append(0.0), c2.append(1 / theta)
        else:
            rn = 1 / (2 * sigma1 - rho)
            c1.append(rn * rho), c2.append(2 * rn / delta)
            rho = rn
    return n, c1, c2


@pytest.mark.parametrize("impl", ["native", "statement"])
@pytest.mark.parametrize("name", ["plane", "sphere"])
@pytest.mark.parametrize("patch_size,depth", [(400, 3), (900, 5)])
def test_patch_plan_reproduces_global_iteration(name, patch_size, depth, impl):
    v, lam, rp, c, A = _system(name)
    V = v.shape[0]
    d = A.diagonal()
    build = PatchPlan.build if impl == "native" else pps.PatchPlan.build
    plan = build(rp, c, d, v, patch_size=patch_size, depth=depth, cap_local=6000)
    assert plan is not None and plan.depth == depth
    T = plan.table
    # patches tile the new numbering; sizes bounded; local ids fit uint16 with the zero slot
    assert T[0, 0] == 0 and (T[1:, 0] == T[:-1, 0] + T[:-1, 1]).all() and T[-1, 0] + T[-1, 1] == V
    assert T[:, 1].max() <= patch_size and plan.max_local < 65535
    assert sorted(plan.perm.tolist()) == list(range(V))
    assert (plan.cols16.astype(np.int64).reshape(-1) <= plan.max_local).all()
    # compactness: a patch is a grid cell, so its ghost layers stay small (no elongated / scattered patches)
    assert plan.redundancy < 4.0
    b = A @ v.astype(np.float64)
    n, c1, c2 = _schedule(A, 1.0, 1e-6)
    xc, xp = np.zeros_like(b), np.zeros_like(b)
    for it in range(n):
        xc, xp = xc + c1[it] * (xc - xp) + c2[it] * (b - A @ xc) / d[:, None], xc
    bn = b[plan.perm]
    cur, prev = np.zeros_like(b), np.zeros_like(b)
    for it0 in range(0, n, plan.depth):
        cur, prev = patch_steps(plan, -lam, bn, cur, prev, c1[it0:it0 + plan.depth], c2[it0:it0 + plan.depth])
    x = np.empty_like(cur)
    x[plan.perm] = cur
    assert np.abs(x - xc).max() <= 1e-12 * np.abs(xc).max(), "s steps on overlapping patches == s global steps"
    assert np.abs(x - v).max() <= 1e-5


@pytest.mark.parametrize("impl", ["native", "statement"])
def test_cell_patches_and_refusal(impl):
    v, f = synthetic.plane(60)
    r, c, val = ol.compute_matrix(v, f, 5.0)
    rp = np.zeros(v.shape[0] + 1, np.int64)
    np.add.at(rp, r + 1, 1)
    rp = np.cumsum(rp)
    d = np.ones(v.shape[0], np.float32)
    build = PatchPlan.build if impl == "native" else pps.PatchPlan.build
    if impl == "statement":
        perm, starts = pps.cell_patches(v, 500)
    else:
        plan0 = build(rp, c, d, v, patch_size=500, depth=2, cap_local=60000)
        perm, starts = plan0.perm, np.concatenate([plan0.table[:, 0], [v.shape[0]]]).astype(np.int64)
    assert starts[0] == 0 and starts[-1] == v.shape[0] and (np.diff(starts) > 0).all() and np.diff(starts).max() <= 500
    # every patch is spatially compact: its bounding box is small compared with the mesh
    for a, b in zip(starts[:-1], starts[1:]):
        p = v[perm[a:b]]
        assert (p.max(0) - p.min(0))[:2].max() <= 0.6
    assert build(rp, c, d, v, patch_size=500, depth=4, cap_local=100) is None, "patches that cannot fit are refused"
    plan = build(rp, c, d, v, patch_size=500, depth=8, cap_local=700)
    assert plan is not None and 2 <= plan.depth < 8 and plan.max_local <= 700, "depth is reduced until the patches fit"


def test_native_patch_plan_is_independent_of_the_thread_count(monkeypatch):
    v, f = synthetic.icosphere(24)
    r, c, val = ol.compute_matrix(v, f, 5.0)
    rp = np.zeros(v.shape[0] + 1, np.int64)
    np.add.at(rp, r + 1, 1)
    rp = np.cumsum(rp)
    d = np.ones(v.shape[0], np.float32)
    plans = []
    for t in ("1", "3", "8"):
        monkeypatch.setenv("LS_PLAN_THREADS", t)
        plans.append(PatchPlan.build(rp, c, d, v, patch_size=700, depth=4, cap_local=6000))
    for q in plans[1:]:
        for name in ("perm", "table", "ghost_gid", "cols16", "diag"):
            assert np.array_equal(getattr(plans[0], name), getattr(q, name)), name
