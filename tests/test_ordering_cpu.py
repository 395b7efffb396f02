"""
The dissection must not depend on how the surface lies in space (what CHOLMOD's graph-based ordering gives the reference for free,
largesteps/solvers.py:34): a sheet that is folded, rolled up, or two shells 1e-3 apart have layers that are neighbours in space and
far apart on the surface, and a cutting plane crosses all of them. Host only (ls_nd_plan_create_ordered); the GPU side of the same
meshes is tests/test_gpu_parity.py::test_folded_surfaces_vs_oracle.
"""
import ctypes

import numpy as np
import pytest

from largesteps import _native, synthetic
from native_plan import native_plan
from statements import nd_factor, nd_solve
from oracle import laplacian as ol
from oracle import solve as osv


def pattern(f, V):
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    e = np.concatenate([e, e[:, ::-1], np.stack([np.arange(V), np.arange(V)], 1)])
    key = np.unique(e[:, 0].astype(np.int64) * V + e[:, 1])
    rowptr = np.zeros(V + 1, np.int64)
    np.add.at(rowptr, key // V + 1, 1)
    return np.cumsum(rowptr), key % V


N = 160            # 25 600 vertices per sheet
FOLDED = {
    "folded": lambda: synthetic.folded_sheet(N),
    "scroll3": lambda: synthetic.scroll(N, 3),
    "scroll10": lambda: synthetic.scroll(N, 10),
    "shells": lambda: synthetic.shells(36),         # 2 x 12 962 vertices
}


@pytest.fixture(scope="module")
def flat_words():
    v, f = synthetic.plane(N)
    rowptr, col = pattern(f, v.shape[0])
    p = native_plan(rowptr, col, v, 64, 4, ordering=-1)
    assert p.ordering == 0 and p.words_other == 0.0 and 0.6 < p.spread < 0.9
    return p.words_per_vertex


@pytest.mark.parametrize("name", list(FOLDED))
def test_folded_surfaces_dissect_like_the_flat_sheet(name, flat_words):
    v, f = FOLDED[name]()
    rowptr, col = pattern(f, v.shape[0])
    plain = native_plan(rowptr, col, v, 64, 4, ordering=0)
    auto = native_plan(rowptr, col, v, 64, 4, ordering=-1)
    # the cutting planes cross several layers: thick separators, recognised by the scale-free spread
    assert plain.spread > 1.3 and plain.words_per_vertex > 1.35 * flat_words
    # ... and the automatic choice dissects again with graph distances among the directions, and takes that plan
    assert auto.ordering == 1 and auto.words_other == pytest.approx(plain.words_per_vertex)
    assert auto.words_per_vertex <= 1.3 * flat_words, (auto.words_per_vertex, flat_words)
    assert auto.spread < 1.0
    assert sorted(auto.perm.tolist()) == list(range(v.shape[0]))


def test_ordinary_surfaces_keep_the_fast_rounds():
    """A flat sheet, a rough closed surface (the bunny / dragon stand-in) and a strip stay below the threshold: ONE plan is built,
    by the rule the device rounds run, array for array what ls_nd_plan_create returns."""
    cv, cf, _ = synthetic.config_mesh("cfg2_bunny70k")
    m = 4000
    sv = np.stack([np.tile(np.arange(m), 8), np.repeat(np.arange(8), m), np.zeros(8 * m)], 1).astype(np.float32)
    i = (np.arange(7)[:, None] * m + np.arange(m - 1)[None, :]).reshape(-1)
    sf = np.concatenate([np.stack([i, i + 1, i + m + 1], 1), np.stack([i, i + m + 1, i + m], 1)])
    for v, f in (synthetic.plane(120), (cv, cf), (sv, sf)):
        rowptr, col = pattern(f, v.shape[0])
        a = native_plan(rowptr, col, v, 64, 4, ordering=-1)
        b = native_plan(rowptr, col, v, 64, 4, ordering=0)
        assert a.ordering == 0 and a.words_other == 0.0 and a.spread <= 1.3
        for name in ("perm", "s", "b", "own_start", "bnd", "ppos", "push_ptr", "push_tgt"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), name


def test_trial_cuts_never_lose_much_and_help_rough_scans():
    """Six trial cuts per domain on meshes that do not need them: within 3 % of the longest-axis rule on a flat sheet, and fewer
    factor numbers on a rough closed surface (noise in space does not move a graph distance's level sets)."""
    v, f = synthetic.plane(200)
    rowptr, col = pattern(f, v.shape[0])
    assert native_plan(rowptr, col, v, 64, 4, ordering=1).words_per_vertex <= 1.03 * native_plan(rowptr, col, v, 64, 4).words_per_vertex
    cv, cf, _ = synthetic.config_mesh("cfg2_bunny70k")
    rowptr, col = pattern(cf, cv.shape[0])
    assert native_plan(rowptr, col, cv, 64, 4, ordering=1).words_per_vertex <= 0.97 * native_plan(rowptr, col, cv, 64, 4).words_per_vertex


@pytest.mark.parametrize("arity,leaf", [(2, 12), (4, 12), (8, 12), (4, 3)])
@pytest.mark.parametrize("mesh", ["scroll", "shells", "soup", "no_positions"])
def test_trial_cut_plans_solve_the_system(mesh, arity, leaf):
    """A plan built from trial cuts is a plan like any other: fed to the numpy statements of the factorisation and of the sweeps
    it solves the system (every array of it is exercised); isolated vertices, several components, no positions."""
    if mesh == "scroll":
        v, f = synthetic.scroll(26, 3)
    elif mesh == "shells":
        v, f = synthetic.shells(3)
        v = np.concatenate([v, np.zeros((3, 3), np.float32)])        # + unreferenced vertices
    elif mesh == "soup":
        rng = np.random.default_rng(5)
        v = rng.standard_normal((300, 3)).astype(np.float32)
        f = rng.integers(0, 270, size=(500, 3)).astype(np.int64)
    else:
        v, f = synthetic.folded_sheet(24)
    kw = dict(lambda_=7.0)
    r, c, val = ol.compute_matrix(v, f, **kw)
    V = v.shape[0]
    rowptr = np.zeros(V + 1, np.int64)
    np.add.at(rowptr, r + 1, 1)
    rowptr = np.cumsum(rowptr)
    p = native_plan(rowptr, c, None if mesh == "no_positions" else v, leaf_size=leaf, arity=arity, ordering=1)
    assert p.ordering == 1 and sorted(p.perm.tolist()) == list(range(V)) and p.b[1] == 0 and int(p.s.sum()) == V
    finv, w = nd_factor(p, rowptr, c, val)
    b = np.random.default_rng(0).standard_normal((V, 3))
    x64 = osv.from_differential(r, c, val, b)
    assert np.abs(nd_solve(p, finv, w, b) - x64).max() <= 1e-10 * np.abs(x64).max()


def test_trial_cut_plan_is_independent_of_the_thread_count(monkeypatch):
    v, f = synthetic.scroll(280, 5)                                   # 78 400 vertices: the parallel in-domain split is taken
    rowptr, col = pattern(f, v.shape[0])
    plans = []
    for threads in ("1", "8", "5"):
        monkeypatch.setenv("LS_PLAN_THREADS", threads)
        plans.append(native_plan(rowptr, col, v, 64, 4, ordering=1))
    for q in plans[1:]:
        for name in ("perm", "s", "b", "own_start", "bnd", "ppos", "push_ptr", "push_tgt"):
            assert np.array_equal(getattr(plans[0], name), getattr(q, name)), name


def test_threshold_and_bad_arguments(monkeypatch):
    v, f = synthetic.folded_sheet(60)
    rowptr, col = pattern(f, v.shape[0])
    monkeypatch.setenv("LS_ND_SUSPECT", "100")                        # nothing is suspect: the folded sheet keeps its thick separators
    assert native_plan(rowptr, col, v, 64, 4, ordering=-1).ordering == 0
    monkeypatch.setenv("LS_ND_SUSPECT", "0.1")                        # everything is: both plans are built, the cheaper one is kept
    v2, f2 = synthetic.plane(60)
    rp2, c2 = pattern(f2, v2.shape[0])
    p = native_plan(rp2, c2, v2, 64, 4, ordering=-1)
    assert p.words_other > 0.0 and p.words_per_vertex <= p.words_other
    h = ctypes.c_void_p()
    rp32, c32 = rowptr.astype(np.int32), col.astype(np.int32)
    rc = _native.lib().ls_nd_plan_create_ordered(v.shape[0], rp32.ctypes.data_as(ctypes.c_void_p), c32.ctypes.data_as(ctypes.c_void_p), None,
                                                 64, 4, 4, 7, ctypes.byref(h))
    assert rc == _native.LS_E_INVALID and "ordering" in _native.last_error()


def _grid_faces(nx, ny, wrapx=False, wrapy=False):
    X, Y = (nx if wrapx else nx - 1), (ny if wrapy else ny - 1)
    x, y = np.meshgrid(np.arange(X), np.arange(Y), indexing="xy")
    x, y = x.ravel(), y.ravel()
    i00, i10, i01, i11 = y * nx + x, y * nx + (x + 1) % nx, ((y + 1) % ny) * nx + x, ((y + 1) % ny) * nx + (x + 1) % nx
    return np.concatenate([np.stack([i00, i10, i11], 1), np.stack([i00, i11, i01], 1)])


def _hard_mesh(name, n=160):
    u, v = np.meshgrid(np.arange(n) / n, np.arange(n) / n, indexing="xy")
    u, v = u.ravel(), v.ravel()
    if name == "torus":                  # a cutting plane crosses the tube twice
        return np.stack([(1 + 0.3 * np.cos(2 * np.pi * v)) * np.cos(2 * np.pi * u), (1 + 0.3 * np.cos(2 * np.pi * v)) * np.sin(2 * np.pi * u),
                         0.3 * np.sin(2 * np.pi * v)], 1), _grid_faces(n, n, True, True)
    if name == "helicoid":               # a spiral staircase of 8 turns, 0.05 high: every plane through the axis crosses 8 layers
        return np.stack([(0.2 + 0.8 * v) * np.cos(16 * np.pi * u), (0.2 + 0.8 * v) * np.sin(16 * np.pi * u), 0.05 * u], 1), _grid_faces(n, n)
    if name == "collapsed":              # every vertex at the origin (a mesh before its first step)
        return np.zeros((n * n, 3)), _grid_faces(n, n)
    if name == "graded":                 # vertex density varying by orders of magnitude: bounding boxes say nothing about vertex counts
        return np.stack([u ** 4, v ** 4, 0 * u], 1), _grid_faces(n, n)
    if name == "swarm":                  # 60 closed surfaces scattered inside each other
        sv, sf = synthetic.icosphere(6)
        rng = np.random.default_rng(0)
        return (np.concatenate([sv * rng.uniform(0.5, 1.5) + rng.normal(0, 0.3, 3) for _ in range(60)]),
                np.concatenate([sf + k * sv.shape[0] for k in range(60)]))
    sv, sf = synthetic.icosphere(50)     # "spike": a sphere with a long thin spike pulled out of a cap
    sp = sv.astype(np.float64).copy()
    cap = sp[:, 2] > 0.95
    sp[cap, 2] += (sp[cap, 2] - 0.95) * 400.0
    return sp, sf


@pytest.mark.parametrize("name", ["torus", "helicoid", "collapsed", "graded", "swarm", "spike"])
def test_automatic_choice_on_meshes_that_mislead_cutting_planes(name):
    """Surfaces on which the positions mislead a cutting plane for OTHER reasons than folds: a tube cut twice, a spiral staircase, a
    mesh whose vertices all sit in one point, graded density, closed surfaces scattered inside each other, a long spike. Whatever the
    reason, the automatic choice must end within 1.25x of the better of the two pure embeddings and with ordinary separators."""
    v, f = _hard_mesh(name)
    v = np.asarray(v, dtype=np.float32)
    rowptr, col = pattern(f, v.shape[0])
    axis = native_plan(rowptr, col, v, 64, 4, ordering=0)
    graph = native_plan(rowptr, col, None, 64, 4, ordering=0)
    auto = native_plan(rowptr, col, v, 64, 4, ordering=-1)
    assert auto.words_per_vertex <= 1.25 * min(axis.words_per_vertex, graph.words_per_vertex), (auto.words_per_vertex, axis.words_per_vertex, graph.words_per_vertex)
    assert auto.words_per_vertex <= axis.words_per_vertex and auto.spread <= 1.3
    assert int((auto.s + auto.b).max()) <= 8000            # the direct solver's front limit


def _tier_lds(p, tier_levels, waves):
    n = ctypes.c_size_t(0)
    s32, b32, o32 = (np.ascontiguousarray(a, dtype=np.int32) for a in (p.s, p.b, p.own_start))
    as_p = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
    _native.check(_native.lib().ls_direct_tier_lds_bytes(int(p.levels), int(p.arity), as_p(s32), as_p(b32), as_p(o32), tier_levels, 1, waves, ctypes.byref(n)))
    return n.value


def test_one_million_vertex_closed_scan_fits_the_sixteen_wave_tier():
    """Round 6's finding, held on the CPU: every mesh >= 800k vertices of rounds 1-5 was a plane (leaves of <= 37 boundary rows). The shipped
    planner on cfg3's recipe at the headline size -- a closed noisy sphere, 998 562 vertices -- gives leaves of up to 83 boundary rows, a root
    separator of ~3400 rows and subtrees 1.2-1.3x apart; with the down sweep's x_bnd BEHIND the leaf triangle in LDS its 16-wave tier needed
    2668 floats per wave (limit 2560) and the solver fell back, silently, to 11 launches on three workgroups per CU (315 us per solve). With
    x_bnd staged IN the triangle area the need is triangle + one vector whatever the boundary: the 4-level tier on 16 waves must fit 160 KB
    (and the 3-level tier on 4 waves four workgroups' worth: 40 KB each)."""
    v, f, _ = synthetic.config_mesh("cfg4b_sphere1m")
    rowptr, col = pattern(f, v.shape[0])
    p = native_plan(rowptr, col, v, 64, 4, ordering=-1)
    assert p.levels == 8 and p.ordering == 1 and 200.0 < p.words_per_vertex < 235.0, (p.levels, p.ordering, p.words_per_vertex)
    leaves = np.arange(p.level_off[p.levels - 1], p.level_off[p.levels])
    assert p.s[leaves].max() <= 64 and p.b[leaves].max() > 40       # (the plane's leaves: <= 37 boundary rows)
    assert 3000 < p.s[1] < 4000
    lds16 = _tier_lds(p, 4, 16)
    assert 0 < lds16 <= 160 * 1024, lds16
    lds4 = _tier_lds(p, 3, 4)
    assert 0 < lds4 <= 40 * 1024, lds4                     # four workgroups per CU, as on the plane
    # the property itself: a leaf's LDS need does not grow with its boundary (the dissection of the first measurement had leaves of 83
    # boundary rows; whatever a dissection produces, up to the triangle's own size, must cost nothing)
    import copy
    q = copy.copy(p)
    q.b = p.b.copy()
    q.b[leaves[:500]] = 90
    assert _tier_lds(q, 4, 16) == lds16 and _tier_lds(q, 3, 4) == lds4
    # the plane of the same size, for scale: same budget, smaller boundaries
    vp, fp, _ = synthetic.config_mesh("cfg4_plane1m")
    rp, cp = pattern(fp, vp.shape[0])
    pp = native_plan(rp, cp, vp, 64, 4, ordering=-1)
    assert pp.b[np.arange(pp.level_off[7], pp.level_off[8])].max() < 64 and _tier_lds(pp, 4, 16) <= lds16
    with pytest.raises(ValueError, match="tier_levels"):
        _tier_lds(p, 9, 16)
    with pytest.raises(ValueError, match="waves"):
        _tier_lds(p, 3, 5)
