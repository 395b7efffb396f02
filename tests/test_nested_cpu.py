"""
Nested-dissection plan of the direct solver (largesteps/nested.py) on the CPU: structural invariants of the plan, the
numpy statement of factorisation + sweeps against the fp64 oracle, and the torch factorisation code of
largesteps/direct.py (device agnostic; here on CPU tensors) against that statement. The HIP sweeps themselves are
covered by tests/test_gpu_parity.py (-m gpu).
"""
import numpy as np
import pytest
import torch

from largesteps import synthetic
from nd_plan_statement import NDPlan
from statements import nd_factor, nd_solve
from oracle import laplacian as ol
from oracle import solve as osv


def csr_of(v, f, **kw):
    r, c, val = ol.compute_matrix(v, f, **kw)
    V = v.shape[0]
    rowptr = np.zeros(V + 1, np.int64)
    np.add.at(rowptr, r + 1, 1)
    return r, np.cumsum(rowptr), c, val


def soup(seed):
    rng = np.random.default_rng(seed)
    V = int(rng.integers(40, 400))
    F = int(rng.integers(V, 3 * V))
    f = rng.integers(0, int(V * 0.9), size=(F, 3)).astype(np.int64)
    f[: F // 5, 0] = int(rng.integers(0, V))           # a hub vertex
    return rng.standard_normal((V, 3)).astype(np.float32), f


MESHES = {
    "plane30": lambda: synthetic.plane(30),
    "ico10": lambda: synthetic.icosphere(10),
    "tiny": lambda: synthetic.icosphere(1),
    "soup0": lambda: soup(0),
    "soup1": lambda: soup(1),
}


def ancestor_at(p, i, lv):
    while p.level_of[i] > lv:
        i = p.parent[i]
    return i


@pytest.mark.parametrize("arity", [2, 4, 8])
@pytest.mark.parametrize("leaf", [4, 16, 64])
@pytest.mark.parametrize("name", list(MESHES))
def test_plan_invariants(name, leaf, arity):
    v, f = MESHES[name]()
    r, rowptr, c, val = csr_of(v, f, lambda_=5.0)
    V = v.shape[0]
    p = NDPlan.build(rowptr, c, v, leaf_size=leaf, arity=arity)
    assert sorted(p.perm.tolist()) == list(range(V)) and np.array_equal(p.inv[p.perm], np.arange(V))
    assert int(p.s[1:].sum()) == V and p.b[1] == 0 and p.s[0] == 0 and p.b[0] == 0
    assert p.n_nodes == sum(arity ** l for l in range(p.levels))
    nn = p.node_of_new
    # contiguous own ranges, deepest level first
    for i in range(1, p.n_nodes + 1):
        assert (nn[p.own_start[i]:p.own_start[i] + p.s[i]] == i).all()
        assert [int(p.parent[ch]) for ch in p.children(i)] == [i] * len(p.children(i))
    # separator property: an entry only links a vertex to its own node, an ancestor or a descendant
    pr, pc = p.inv[r], p.inv[c]
    for a, d in set(zip(nn[pr].tolist(), nn[pc].tolist())):
        lo, hi = (a, d) if p.level_of[a] >= p.level_of[d] else (d, a)
        assert ancestor_at(p, lo, p.level_of[hi]) == hi
    # boundary sets: sorted, strictly after the own range, inside the ancestors; children's boundaries nest
    for i in range(1, p.n_nodes + 1):
        bi = p.bnd[p.bnd_off[i]:p.bnd_off[i] + p.b[i]]
        assert (np.diff(bi) > 0).all() and (bi >= p.own_start[i] + p.s[i]).all()
        for a in set(nn[bi].tolist()):
            assert p.level_of[a] < p.level_of[i] and ancestor_at(p, i, p.level_of[a]) == a, "boundary vertices live in ancestors"
        if i > 1:
            par = int(p.parent[i])
            front = np.concatenate([np.arange(p.own_start[par], p.own_start[par] + p.s[par]),
                                    p.bnd[p.bnd_off[par]:p.bnd_off[par] + p.b[par]]])
            pp = p.ppos[p.bnd_off[i]:p.bnd_off[i] + p.b[i]]
            assert np.array_equal(front[pp], bi)
    # push lists: front position -> exactly the children's boundary entries that are the same vertex
    n_front = int((p.s + p.b).sum())
    assert p.push_ptr.shape[0] == n_front + 1 and p.push_ptr[-1] == p.bnd.shape[0]
    assert sorted(p.push_tgt.tolist()) == list(range(p.bnd.shape[0]))
    for i in range(1, p.n_nodes + 1):
        front = np.concatenate([np.arange(p.own_start[i], p.own_start[i] + p.s[i]), p.bnd[p.bnd_off[i]:p.bnd_off[i] + p.b[i]]])
        for q_, vert in enumerate(front):
            tg = p.push_tgt[p.push_ptr[p.front_off[i] + q_]:p.push_ptr[p.front_off[i] + q_ + 1]]
            assert (p.bnd[tg] == vert).all()
            owners = np.searchsorted(p.bnd_off, tg, side="right") - 1       # node of every target entry (skips empty nodes)
            assert all(int(p.parent[o]) == i for o in owners)
    assert p.factor_entries == int((p.s * p.s + 2 * p.s * p.b).sum())
    with pytest.raises(ValueError):
        NDPlan.build(rowptr, c, v[:-1], leaf_size=leaf)
    with pytest.raises(ValueError):
        NDPlan.build(rowptr, c, v, arity=3)


@pytest.mark.parametrize("name,kw", [("plane30", dict(lambda_=30.0)), ("ico10", dict(lambda_=0.0, alpha=0.9, cotan=True)),
                                      ("soup0", dict(lambda_=3.0)), ("soup1", dict(lambda_=0.0, alpha=0.5)), ("tiny", dict(lambda_=1.0))])
@pytest.mark.parametrize("arity", [2, 4, 8])
def test_numpy_statement_vs_oracle(name, kw, arity):
    v, f = MESHES[name]()
    if kw.get("cotan"):
        v = synthetic.perturb(v, radial=0.05, tangential=0.1, edge=0.1, seed=1)
    r, rowptr, c, val = csr_of(v, f, **kw)
    p = NDPlan.build(rowptr, c, v, leaf_size=12, arity=arity)
    finv, w = nd_factor(p, rowptr, c, val)
    b = np.random.default_rng(0).standard_normal((v.shape[0], 3))
    x = nd_solve(p, finv, w, b)
    x64 = osv.from_differential(r, c, val, b)
    assert np.abs(x - x64).max() <= 1e-10 * np.abs(x64).max()


@pytest.mark.parametrize("name,kw", [("plane30", dict(lambda_=30.0)), ("ico10", dict(lambda_=0.0, alpha=0.9, cotan=True)), ("soup0", dict(lambda_=3.0))])
@pytest.mark.parametrize("arity", [2, 4, 8])
def test_torch_factorisation_matches_statement(name, kw, arity):
    """largesteps.direct.factorize (what runs on the MI355X, here on CPU tensors): padded level batches, extend-add by
    index arithmetic, packing into the three flat fp32 arrays of the C ABI."""
    from nd_factor_statement import factorize
    v, f = MESHES[name]()
    r, rowptr, c, val = csr_of(v, f, **kw)
    p = NDPlan.build(rowptr, c, v, leaf_size=10, arity=arity)
    finv, w = nd_factor(p, rowptr, c, val)
    fac = factorize(p, rowptr, c, torch.from_numpy(val), torch.device("cpu"), sparse_leaves=False)
    finv_t, wf_t, wb_t = fac.finv, fac.wf, fac.wb
    assert np.array_equal(fac.finv_off, p.finv_off) and np.array_equal(fac.w_off, p.w_off) and not fac.sparse.any()
    scale = np.abs(finv).max()
    assert np.abs(finv_t.numpy()[:p.finv_size] - finv).max() <= 2e-7 * scale
    assert np.abs(wb_t.numpy()[:p.w_size] - w).max() <= 2e-7 * max(np.abs(w).max(), 1e-30)
    for i in range(1, p.n_nodes + 1):
        s, b = int(p.s[i]), int(p.b[i])
        if s and b:
            W = wb_t.numpy()[p.w_off[i]:p.w_off[i] + s * b].reshape(b, s)
            Wf = wf_t.numpy()[p.w_off[i]:p.w_off[i] + s * b].reshape(s, b)
            assert np.array_equal(Wf, W.T)
    # and the fp32 factor solves the system to fp32 accuracy through the numpy sweeps
    b = np.random.default_rng(1).standard_normal((v.shape[0], 2))
    x = nd_solve(p, finv_t.numpy()[:p.finv_size].astype(np.float64), wb_t.numpy()[:p.w_size].astype(np.float64), b)
    x64 = osv.from_differential(r, c, val, b)
    assert np.abs(x - x64).max() <= 1e-5 * np.abs(x64).max()


@pytest.mark.parametrize("name,kw", [("plane30", dict(lambda_=30.0)), ("ico10", dict(lambda_=0.0, alpha=0.9, cotan=True)), ("soup0", dict(lambda_=3.0))])
@pytest.mark.parametrize("arity", [2, 4])
def test_sparse_leaf_format(name, kw, arity):
    """Leaves in the tier kernel's format (csrc/nd_tier.h): one packed triangle of F_ss^-1 and the matrix block A_bs as
    two CSR lists. The tables must reproduce the dense statement: tri == tril(Finv), A_bs Finv == W, and the sweeps
    written with them (y = Finv b, upd = A_bs y; x = y - Finv A_sb x_bnd) give the dense sweeps' result."""
    from nd_factor_statement import factorize
    v, f = MESHES[name]()
    r, rowptr, c, val = csr_of(v, f, **kw)
    p = NDPlan.build(rowptr, c, v, leaf_size=10, arity=arity)
    finv, w = nd_factor(p, rowptr, c, val)
    fac = factorize(p, rowptr, c, torch.from_numpy(val), torch.device("cpu"), sparse_leaves=True, tier_levels=2)
    leaves = p.level_nodes(p.levels - 1)
    assert fac.sparse[leaves].sum() == ((p.s[leaves] >= 1) & (p.s[leaves] <= 64)).sum() > 0 and not fac.sparse[:leaves[0]].any()
    tri, ptr = fac.tri.numpy(), fac.sp_ptr.numpy()
    ent = fac.sp_ent.numpy().view(np.dtype([("val", np.float32), ("idx", np.int32)]))
    dense_words = 0
    for i in range(1, p.n_nodes + 1):
        s, b = int(p.s[i]), int(p.b[i])
        Fi = finv[p.finv_off[i]:p.finv_off[i] + s * s].reshape(s, s)
        W = w[p.w_off[i]:p.w_off[i] + b * s].reshape(b, s)
        if fac.quad[i]:                       # quad-interleaved streams of a dense tier node
            assert fac.tri_off[i] < 0 and p.level_of[i] >= p.levels - 2
            s4, b4 = (s + 3) & ~3, (b + 3) & ~3
            u4 = fac.u4.numpy()[fac.w_off_all[i]:fac.w_off_all[i] + s4 * b].reshape(s4 // 4, b, 4)
            got_w = u4.transpose(1, 0, 2).reshape(b, s4)
            assert np.abs(got_w[:, :s] - W).max(initial=0) <= 2e-7 * max(np.abs(W).max(initial=0), 1e-30) and not got_w[:, s:].any()
            d4 = fac.d4.numpy()[fac.finv_off_all[i]:fac.finv_off_all[i] + (s4 + b4) * s].reshape((s4 + b4) // 4, s, 4)
            got_d = d4.transpose(1, 0, 2).reshape(s, s4 + b4)
            assert np.abs(got_d[:, :s] - Fi).max(initial=0) <= 2e-7 * max(np.abs(Fi).max(initial=0), 1e-30)
            assert np.abs(got_d[:, s4:s4 + b] - W.T).max(initial=0) <= 2e-7 * max(np.abs(W).max(initial=0), 1e-30)
            assert not got_d[:, s:s4].any() and not got_d[:, s4 + b:].any()
            dense_words += 1
            continue
        if not fac.sparse[i]:
            assert fac.tri_off[i] < 0
            got = fac.finv.numpy()[fac.finv_off[i]:fac.finv_off[i] + s * s].reshape(s, s)
            assert np.abs(got - Fi).max(initial=0) <= 2e-7 * max(np.abs(Fi).max(initial=0), 1e-30)
            dense_words += s * s + 2 * s * b
            continue
        o = int(fac.tri_off[i])
        assert o % 4 == 0
        for rr in range(s):
            np.testing.assert_allclose(tri[o + rr * (rr + 1) // 2:o + rr * (rr + 1) // 2 + rr + 1], Fi[rr, :rr + 1], rtol=0, atol=2e-7 * np.abs(Fi).max())
        A_bs = np.zeros((b, s))
        for ib in range(b):
            e = ent[ptr[fac.spb_off[i] + ib]:ptr[fac.spb_off[i] + ib + 1]]
            assert (np.diff(e["idx"]) > 0).all()
            A_bs[ib, e["idx"]] = e["val"]
        A_sb = np.zeros((s, b))
        for js in range(s):
            e = ent[ptr[fac.sps_off[i] + js]:ptr[fac.sps_off[i] + js + 1]]
            A_sb[js, e["idx"]] = e["val"]
        assert np.array_equal(A_sb, A_bs.T)
        np.testing.assert_allclose(A_bs @ Fi, W, rtol=0, atol=1e-6 * max(np.abs(W).max(initial=0), 1e-30))
    assert dense_words > 0 or p.levels == 1


def test_smoothed_positions_give_thin_separators():
    """A rough surface (radial noise far above the edge length, SURVEY's bunny / dragon stand-ins): bisecting the raw
    positions leaves separators that do not shrink with the domains; a few neighbour-averaging passes restore them."""
    v, f = synthetic.icosphere(70)                     # 49k vertices, edge ~0.015 against 0.05 of radial noise
    v = synthetic.perturb(v, radial=0.05, seed=0)
    r, rowptr, c, val = csr_of(v, f, lambda_=10.0)
    raw = NDPlan.build(rowptr, c, v, leaf_size=64, arity=4, smooth=0)
    smooth = NDPlan.build(rowptr, c, v, leaf_size=64, arity=4, smooth=4)
    assert smooth.factor_entries < 0.5 * raw.factor_entries
    # and a smoothed plan is as valid as any other: the numpy statement still solves the system
    v, f = synthetic.icosphere(12)
    v = synthetic.perturb(v, radial=0.05, seed=1)
    r, rowptr, c, val = csr_of(v, f, lambda_=10.0)
    plan = NDPlan.build(rowptr, c, v, leaf_size=16, arity=4, smooth=4)
    finv, w = nd_factor(plan, rowptr, c, val)
    b = np.random.default_rng(0).standard_normal((v.shape[0], 2))
    x64 = osv.from_differential(r, c, val, b)
    assert np.abs(nd_solve(plan, finv, w, b) - x64).max() <= 1e-10 * np.abs(x64).max()


def test_graph_embedding_replaces_positions():
    """No positions (a matrix built elsewhere): graph distances from three far landmarks order the vertices well enough
    for thin separators; disconnected components and isolated vertices are laid out side by side."""
    from nd_plan_statement import graph_embedding
    v, f = synthetic.plane(60)
    r, rowptr, c, val = csr_of(v, f, lambda_=5.0)
    pos = graph_embedding(rowptr, c, v.shape[0])
    assert pos.shape == v.shape and np.isfinite(pos).all()
    with_pos = NDPlan.build(rowptr, c, v, leaf_size=32)
    without = NDPlan.build(rowptr, c, pos, leaf_size=32)
    assert without.factor_entries < 2.0 * with_pos.factor_entries
    finv, w = nd_factor(without, rowptr, c, val)
    b = np.random.default_rng(0).standard_normal((v.shape[0], 2))
    x64 = osv.from_differential(r, c, val, b)
    assert np.abs(nd_solve(without, finv, w, b) - x64).max() <= 1e-10 * np.abs(x64).max()
    # two components + unreferenced vertices
    v2, f2 = synthetic.icosphere(4)
    vv = np.concatenate([v2, v2 + 5.0, np.zeros((3, 3), np.float32)])
    ff = np.concatenate([f2, f2 + v2.shape[0]])
    r, rowptr, c, val = csr_of(vv, ff, lambda_=2.0)
    pos = graph_embedding(rowptr, c, vv.shape[0])
    assert np.isfinite(pos).all() and pos[:v2.shape[0], 0].max() < pos[v2.shape[0]:2 * v2.shape[0], 0].min()
    plan = NDPlan.build(rowptr, c, pos, leaf_size=16)
    finv, w = nd_factor(plan, rowptr, c, val)
    b = np.random.default_rng(1).standard_normal((vv.shape[0], 1))
    x64 = osv.from_differential(r, c, val, b)
    assert np.abs(nd_solve(plan, finv, w, b) - x64).max() <= 1e-10 * np.abs(x64).max()


@pytest.mark.parametrize("shape", ["fan", "strip"])
def test_extreme_meshes(shape):
    """A hub of valence 3000 and a 2 x 6000 strip: the separators stay tiny (the hub itself / two vertices), with the
    real positions and with graph-distance pseudo-positions."""
    from nd_plan_statement import graph_embedding
    if shape == "fan":
        n = 3000
        ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
        v = np.concatenate([[[0, 0, 0]], np.stack([np.cos(ang), np.sin(ang), 0 * ang], 1)]).astype(np.float32)
        f = np.stack([np.zeros(n, int), 1 + np.arange(n), 1 + (np.arange(n) + 1) % n], 1).astype(np.int64)
    else:
        m = 6000
        v = np.stack([np.tile(np.arange(m), 2), np.repeat([0, 1], m), np.zeros(2 * m)], 1).astype(np.float32)
        a = np.arange(m - 1)
        f = np.concatenate([np.stack([a, a + 1, m + a], 1), np.stack([a + 1, m + a + 1, m + a], 1)]).astype(np.int64)
    r, rowptr, c, val = csr_of(v, f, lambda_=10.0)
    b = np.random.default_rng(0).standard_normal((v.shape[0], 2))
    x64 = osv.from_differential(r, c, val, b)
    for pos in (v, graph_embedding(rowptr, c, v.shape[0])):
        p = NDPlan.build(rowptr, c, pos, leaf_size=64, arity=4)
        assert int((p.s + p.b).max()) <= 80 and p.factor_entries <= 80 * v.shape[0]
        finv, w = nd_factor(p, rowptr, c, val)
        assert np.abs(nd_solve(p, finv, w, b) - x64).max() <= 1e-10 * np.abs(x64).max()


@pytest.mark.parametrize("name,kw", [("plane30", dict(lambda_=30.0)), ("ico10", dict(lambda_=0.0, alpha=0.9, cotan=True)),
                                      ("soup0", dict(lambda_=3.0)), ("soup1", dict(lambda_=0.0, alpha=0.5)), ("tiny", dict(lambda_=1.0))])
@pytest.mark.parametrize("arity,leaf", [(2, 12), (4, 12), (8, 12), (4, 3)])
@pytest.mark.parametrize("with_pos", [True, False])
def test_native_plan_solves_the_system(name, kw, arity, leaf, with_pos):
    """The C++ symbolic analysis (csrc/nd_plan.cpp, through the host-only C entry points ls_nd_plan_*): its plan, fed to the
    numpy statements of the factorisation and of the two sweeps, solves the system -- which exercises every array of it
    (ordering, fronts, parent positions, push lists). With and without vertex positions (graph-distance embedding)."""
    from native_plan import native_plan
    v, f = MESHES[name]()
    if kw.get("cotan"):
        v = synthetic.perturb(v, radial=0.05, tangential=0.1, edge=0.1, seed=1)
    r, rowptr, c, val = csr_of(v, f, **kw)
    p = native_plan(rowptr, c, v if with_pos else None, leaf_size=leaf, arity=arity)
    V = v.shape[0]
    assert sorted(p.perm.tolist()) == list(range(V)) and p.b[1] == 0 and int(p.s.sum()) == V
    # ordering: deepest level first, every node one contiguous range; boundary lists ascending and above the own block
    for i in range(1, p.n_nodes + 1):
        bi = p.bnd[p.bnd_off[i]:p.bnd_off[i] + p.b[i]]
        assert (np.diff(bi) > 0).all() and (bi >= p.own_start[i] + p.s[i]).all()
    finv, w = nd_factor(p, rowptr, c, val)
    b = np.random.default_rng(0).standard_normal((V, 3))
    x = nd_solve(p, finv, w, b)
    x64 = osv.from_differential(r, c, val, b)
    assert np.abs(x - x64).max() <= 1e-10 * np.abs(x64).max()
    if with_pos and name in ("plane30", "ico10"):            # same quality as the numpy statement's plan
        q = NDPlan.build(rowptr, c, v, leaf_size=leaf, arity=arity)
        assert p.levels == q.levels and abs(p.factor_entries - q.factor_entries) <= 0.02 * q.factor_entries


@pytest.mark.parametrize("with_pos", [True, False])
def test_native_plan_is_independent_of_the_thread_count(with_pos, monkeypatch):
    """Rounds with fewer domains than threads split a domain with a parallel bucket selection (csrc/nd_plan.cpp); the plan must
    be the one a single thread computes -- every array identical -- on a mesh large enough to take that path (> 65536 vertices,
    degenerate keys: a plane has 300 distinct x values)."""
    from native_plan import native_plan
    v, f = synthetic.plane(300)
    v = v.copy()
    v[:, 2] = 0.05 * np.sin(7.0 * v[:, 0]) * np.cos(5.0 * v[:, 1])
    r, rowptr, c, val = csr_of(v, f, lambda_=5.0)
    plans = []
    for threads in ("1", "8", "5"):
        monkeypatch.setenv("LS_PLAN_THREADS", threads)
        plans.append(native_plan(rowptr, c, v if with_pos else None, leaf_size=64, arity=4))
    a = plans[0]
    assert sorted(a.perm.tolist()) == list(range(v.shape[0]))
    for q in plans[1:]:
        for name in ("perm", "s", "b", "own_start", "bnd", "ppos", "push_ptr", "push_tgt"):
            assert np.array_equal(getattr(a, name), getattr(q, name)), name


def test_native_plans_built_at_the_same_moment():
    """The analysis keeps ONE pool of host threads for the life of the process (csrc/nd_plan.cpp, PoolLease): a second analysis that
    starts while the first one holds it (a multi-device solver builds one plan per device from its own thread; ctypes releases the
    GIL) gets a pool of its own. Both must produce the plan a lone call produces, call after call."""
    import threading
    from native_plan import native_plan
    meshes = []
    for n in (120, 90):
        v, f = synthetic.plane(n)
        r, rowptr, c, val = csr_of(v, f, lambda_=5.0)
        meshes.append((rowptr, c, v))
    alone = [native_plan(rp, c, v, leaf_size=32, arity=4) for rp, c, v in meshes]
    for rep in range(3):
        got = [None, None]

        def work(i):
            rp, c, v = meshes[i]
            got[i] = native_plan(rp, c, v, leaf_size=32, arity=4)

        ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for a, q in zip(alone, got):
            for name in ("perm", "s", "b", "own_start", "bnd", "ppos", "push_ptr", "push_tgt"):
                assert np.array_equal(getattr(a, name), getattr(q, name)), name


def test_native_plan_in_a_forked_child():
    """A forked child has none of the parent's threads: the process-wide pool of the parent must not be waited for there."""
    import os
    from native_plan import native_plan
    v, f = synthetic.plane(100)
    r, rowptr, c, val = csr_of(v, f, lambda_=5.0)
    a = native_plan(rowptr, c, v, leaf_size=32, arity=4)           # the parent's pool exists now
    rd, wr = os.pipe()
    pid = os.fork()
    if pid == 0:
        ok = b"0"
        try:
            q = native_plan(rowptr, c, v, leaf_size=32, arity=4)
            ok = b"1" if np.array_equal(a.perm, q.perm) and np.array_equal(a.bnd, q.bnd) else b"0"
        finally:
            os.write(wr, ok)
            os._exit(0)
    os.close(wr)
    import select
    ready, _, _ = select.select([rd], [], [], 60.0)
    if not ready:
        os.kill(pid, 9)
    os.waitpid(pid, 0)
    assert ready, "the child hung in the analysis"
    assert os.read(rd, 1) == b"1"
