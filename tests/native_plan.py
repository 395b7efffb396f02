"""TEST helper: run the native (C++) symbolic analysis through the C ABI (host only, no GPU) and wrap its arrays in the
attribute layout of the numpy statement's plan object (largesteps.nested.NDPlan), so that tests/statements.py can
factorise and solve with it."""
import ctypes

import numpy as np

from largesteps import _native
from nd_plan_statement import NDPlan


def native_plan(rowptr, col, positions, leaf_size=64, arity=4, smooth=4, ordering=0):
    """ordering: 0 longest axis of the embedding (ls_nd_plan_create), 1 thinnest of six trial separators, -1 the automatic choice
    (ls_nd_plan_create_ordered); the plan carries .ordering / .words_per_vertex / .spread / .words_other (ls_nd_plan_quality)."""
    lib = _native.lib()
    rowptr32 = np.ascontiguousarray(rowptr, dtype=np.int32)
    col32 = np.ascontiguousarray(col, dtype=np.int32)
    V = rowptr32.shape[0] - 1
    pos = None if positions is None else np.ascontiguousarray(positions, dtype=np.float32)
    as_p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
    h = ctypes.c_void_p()
    if ordering == 0:
        _native.check(lib.ls_nd_plan_create(V, as_p(rowptr32), as_p(col32), as_p(pos), leaf_size, arity, smooth, ctypes.byref(h)))
    else:
        _native.check(lib.ls_nd_plan_create_ordered(V, as_p(rowptr32), as_p(col32), as_p(pos), leaf_size, arity, smooth, ordering, ctypes.byref(h)))
    try:
        lv, ar, nn = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        nb, nf, sec = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double()
        _native.check(lib.ls_nd_plan_info(h, lv, ar, nn, nb, nf, sec))
        n = nn.value
        q_ord, q_w, q_s, q_o = ctypes.c_int(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _native.check(lib.ls_nd_plan_quality(h, q_ord, q_w, q_s, q_o))
        perm = np.zeros(V, np.int32)
        s, b, own_start, parent = (np.zeros(n + 1, np.int32) for _ in range(4))
        bnd, ppos, push_tgt = (np.zeros(nb.value, np.int32) for _ in range(3))
        push_ptr = np.zeros(nf.value + 1, np.int32)
        _native.check(lib.ls_nd_plan_arrays(h, as_p(perm), as_p(s), as_p(b), as_p(own_start), as_p(parent), as_p(bnd), as_p(ppos),
                                            as_p(push_ptr), as_p(push_tgt)))
    finally:
        lib.ls_nd_plan_destroy(h)
    p = NDPlan()
    p.V, p.levels, p.arity, p.n_nodes = V, lv.value, ar.value, n
    p.D = p.levels - 1
    p.seconds = sec.value
    p.ordering, p.words_per_vertex, p.spread, p.words_other = q_ord.value, q_w.value, q_s.value, q_o.value
    p.level_off = np.array([1 + (p.arity ** l - 1) // (p.arity - 1) for l in range(p.levels + 1)], dtype=np.int64)
    p.level_of = np.zeros(n + 1, np.int64)
    p.child_ix = np.zeros(n + 1, np.int64)
    for l in range(p.levels):
        ids = np.arange(p.level_off[l], p.level_off[l + 1])
        p.level_of[ids] = l
        if l:
            p.child_ix[ids] = (ids - p.level_off[l]) % p.arity
            assert np.array_equal(parent[ids], p.level_off[l - 1] + (ids - p.level_off[l]) // p.arity)
    p.parent = parent.astype(np.int64)
    p.perm = perm.astype(np.int64)
    p.inv = np.empty(V, np.int64)
    p.inv[p.perm] = np.arange(V)
    p.s, p.b, p.own_start = s.astype(np.int64), b.astype(np.int64), own_start.astype(np.int64)
    p.bnd, p.ppos = bnd.astype(np.int64), ppos.astype(np.int64)
    p.bnd_off = np.concatenate([[0], np.cumsum(p.b)])[:-1]
    p.front_off = np.concatenate([[0], np.cumsum(p.s + p.b)])[:-1]
    p.push_ptr, p.push_tgt = push_ptr.astype(np.int64), push_tgt.astype(np.int64)
    p.finv_off = np.concatenate([[0], np.cumsum(p.s * p.s)])[:-1]
    p.w_off = np.concatenate([[0], np.cumsum(p.s * p.b)])[:-1]
    p.finv_size, p.w_size = int((p.s * p.s).sum()), int((p.s * p.b).sum())
    node_of_new = np.zeros(V, np.int64)
    order = np.concatenate([np.arange(p.level_off[l], p.level_off[l + 1]) for l in range(p.levels - 1, -1, -1)])
    node_of_new[:] = np.repeat(order, p.s[order])
    p.node_of_new = node_of_new
    return p
