"""The bisection rounds of the direct solver's analysis run on the device (csrc/nd_bisect.hip); the host rounds (csrc/nd_plan.cpp,
ls_nd_plan_create: what the CPU tests pin to the numpy statement) are what they are checked against -- the two must produce the
SAME plan, array for array: after every single round (arity 2, leaf size V >> r stops after r rounds), at the solver's own
settings, on closed surfaces, open ones, rough ones, matrices without positions, matrices with empty rows."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu


def _pattern(v, f):
    V = v.shape[0]
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    A = sp.coo_matrix((np.ones(len(e) * 2), (np.r_[e[:, 0], e[:, 1]], np.r_[e[:, 1], e[:, 0]])), shape=(V, V)).tocsr()
    A = (A + sp.identity(V)).tocsr()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32)


def _grid(nx, ny):
    """an open nx x ny grid of unit squares cut into triangles: whole rows / columns of equal coordinates (ties in every sort)"""
    x, y = np.meshgrid(np.arange(nx, dtype=np.float32), np.arange(ny, dtype=np.float32), indexing="xy")
    v = np.stack([x.ravel(), y.ravel(), np.zeros(nx * ny, np.float32)], axis=1)
    i = (np.arange(ny - 1)[:, None] * nx + np.arange(nx - 1)[None, :]).ravel()
    f = np.concatenate([np.stack([i, i + 1, i + nx], axis=1), np.stack([i + 1, i + nx + 1, i + nx], axis=1)]).astype(np.int64)
    return v, f


def _arrays(lib, h, V):
    from largesteps import _native
    lv, ar, nn = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    nb, nf, sec = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double()
    _native.check(lib.ls_nd_plan_info(h, lv, ar, nn, nb, nf, sec))
    n = nn.value
    out = {"perm": np.zeros(V, np.int32)}
    for k in ("s", "b", "own_start", "parent"):
        out[k] = np.zeros(n + 1, np.int32)
    for k in ("bnd", "ppos", "push_tgt"):
        out[k] = np.zeros(nb.value, np.int32)
    out["push_ptr"] = np.zeros(nf.value + 1, np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
    _native.check(lib.ls_nd_plan_arrays(h, p(out["perm"]), p(out["s"]), p(out["b"]), p(out["own_start"]), p(out["parent"]), p(out["bnd"]),
                                        p(out["ppos"]), p(out["push_ptr"]), p(out["push_tgt"])))
    out["levels"], out["seconds"] = lv.value, sec.value
    return out


def _both(rowptr, col, pos, leaf, arity, smooth=4, ordering=0):
    from largesteps import _native
    lib = _native.lib()
    dev = torch.device("cuda:0")
    V = rowptr.shape[0] - 1
    as_p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
    pos32 = None if pos is None else np.ascontiguousarray(pos, dtype=np.float32)
    h = ctypes.c_void_p()
    if ordering:
        _native.check(lib.ls_nd_plan_create_ordered(V, as_p(rowptr), as_p(col), as_p(pos32), leaf, arity, smooth, ordering, ctypes.byref(h)))
    else:
        _native.check(lib.ls_nd_plan_create(V, as_p(rowptr), as_p(col), as_p(pos32), leaf, arity, smooth, ctypes.byref(h)))
    try:
        host = _arrays(lib, h, V)
    finally:
        lib.ls_nd_plan_destroy(h)
    d_rp, d_col = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
    d_pos = None if pos32 is None else torch.from_numpy(pos32).to(dev)
    h = ctypes.c_void_p()
    if ordering:
        _native.check(lib.ls_nd_plan_create_device_ordered(_native.ptr(d_rp), _native.ptr(d_col), _native.ptr(d_pos), V, col.shape[0], leaf, arity, smooth,
                                                           ordering, 0, _native.stream_of(dev), ctypes.byref(h)))
    else:
        _native.check(lib.ls_nd_plan_create_device(_native.ptr(d_rp), _native.ptr(d_col), _native.ptr(d_pos), V, col.shape[0], leaf, arity, smooth,
                                                   0, _native.stream_of(dev), ctypes.byref(h)))
    try:
        device = _arrays(lib, h, V)
    finally:
        lib.ls_nd_plan_destroy(h)
    return host, device


def _same(host, device, what):
    assert host["levels"] == device["levels"], what
    for k in ("perm", "s", "b", "own_start", "parent", "bnd", "ppos", "push_ptr", "push_tgt"):
        assert np.array_equal(host[k], device[k]), f"{what}: {k} differs (first at {np.flatnonzero(host[k] != device[k])[:5]})"


def _meshes():
    from largesteps import synthetic
    rng = np.random.default_rng(5)
    v, f = synthetic.icosphere(5)
    yield "icosphere 10k", v, f
    v2 = v + 0.02 * rng.standard_normal(v.shape).astype(np.float32)
    yield "rough sphere", v2, f
    v, f, _ = synthetic.config_mesh("cfg2_bunny70k")
    yield "bunny stand-in 70k", v, f
    v, f = _grid(150, 97)
    yield "open grid, ties along both axes", v, f


def test_device_rounds_equal_host_rounds_after_every_round():
    v, f = _grid(120, 90)
    rowptr, col = _pattern(v, f)
    V = v.shape[0]
    for r in range(1, 9):
        host, device = _both(rowptr, col, v, V >> r, 2)
        _same(host, device, f"after {r} rounds")


@pytest.mark.parametrize("arity", [2, 4, 8])
def test_device_plan_equals_host_plan(arity):
    for name, v, f in _meshes():
        rowptr, col = _pattern(v, f)
        host, device = _both(rowptr, col, v, 64, arity)
        _same(host, device, f"{name}, arity {arity}")


@pytest.mark.parametrize("arity", [2, 4, 8])
def test_device_trial_cuts_equal_host_trial_cuts(arity):
    """ND_ORDER_MINSEP (every domain tries the three position axes and three graph distances and takes the thinnest separator) on the
    device -- six sorted lists, a side bit and a cut bit per vertex and direction -- against the host rounds of round 4: every array of the
    finished plan, on smooth, rough and open meshes with ties, on a surface rolled up in space, without positions (three directions) and
    with empty rows (no averaging)."""
    from largesteps import synthetic
    for name, v, f in _meshes():
        rowptr, col = _pattern(v, f)
        host, device = _both(rowptr, col, v, 64, arity, ordering=1)
        _same(host, device, f"trial cuts: {name}, arity {arity}")
    v, f = synthetic.scroll(120, 3)
    rowptr, col = _pattern(v, f)
    host, device = _both(rowptr, col, v, 64, arity, ordering=1)
    _same(host, device, f"trial cuts: scroll, arity {arity}")
    v, f = synthetic.icosphere(4)
    rowptr, col = _pattern(v, f)
    host, device = _both(rowptr, col, None, 32, arity, ordering=1)
    _same(host, device, "trial cuts without positions")
    rp2 = np.concatenate([rowptr, [rowptr[-1], rowptr[-1]]]).astype(np.int32)
    v2 = np.concatenate([v, [[3.0, 0, 0], [0, 3.0, 0]]]).astype(np.float32)
    host, device = _both(rp2, col, v2, 32, arity, ordering=1)
    _same(host, device, "trial cuts with empty rows")


def test_device_trial_cuts_after_every_round():
    v, f = _grid(120, 90)
    rowptr, col = _pattern(v, f)
    V = v.shape[0]
    for r in range(1, 9):
        host, device = _both(rowptr, col, v, V >> r, 2, ordering=1)
        _same(host, device, f"trial cuts after {r} rounds")


def test_device_plan_without_positions_and_with_empty_rows():
    from largesteps import synthetic
    v, f = synthetic.icosphere(4)
    rowptr, col = _pattern(v, f)
    host, device = _both(rowptr, col, None, 32, 4)
    _same(host, device, "graph embedding")
    # two isolated vertices without any entry (not even a diagonal): no smoothing on either side
    V = v.shape[0]
    rp2 = np.concatenate([rowptr, [rowptr[-1], rowptr[-1]]]).astype(np.int32)
    v2 = np.concatenate([v, [[3.0, 0, 0], [0, 3.0, 0]]]).astype(np.float32)
    host, device = _both(rp2, col, v2, 32, 4)
    _same(host, device, "empty rows")
    assert V + 2 == host["perm"].shape[0]


def test_device_plan_at_one_million_vertices_and_constructor_time():
    from largesteps import synthetic
    v, f, _ = synthetic.config_mesh("cfg4_plane1m")
    rowptr, col = _pattern(v, f)
    host, device = _both(rowptr, col, v, 64, 4)
    _same(host, device, "1M plane")
    print(f"analysis seconds: host rounds {host['seconds']:.3f}, device rounds {device['seconds']:.3f}")
    assert device["seconds"] < host["seconds"]
