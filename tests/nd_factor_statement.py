"""
TEST CODE -- torch statement of the numeric factorisation the product does with hand-written kernels
(csrc/nd_factor.hip), and a handle built from PYTHON arrays through ls_direct_create (the array-level entry of the C ABI):
numeric multifrontal factorisation of a nested-dissection plan (tests/nd_plan_statement.py) in the layouts of the native
re-solve kernels (csrc/direct.hip, csrc/nd_tier.h).

Replaces: largesteps/solvers.py:26-39 of the reference (CholeskySolver: cholespy / CHOLMOD factorisation in the
constructor, two triangular solves per call). Here the constructor
  1. builds the elimination tree from the vertex positions (host, numpy, integer work only),
  2. factorises level by level ON THE DEVICE: every tree level is one batch of dense fronts (padded to the level's
     largest front), assembled by index arithmetic and factorised in fp64 with torch.linalg (rocSOLVER / rocBLAS):
     Finv = F_ss^-1, W = F_bs Finv, U = F_bb - W F_sb -> parent,
  3. packs Finv and W (twice: both sweep layouts) in fp32 and hands the device pointers to ls_direct_create.
Every solve afterwards is a handful of hand-written HIP launches (one per upper tree level and sweep, one per sweep for the tier) that read the factor once (W twice).
"""
import ctypes
import os

import numpy as np
import torch

from largesteps import _native
from nd_plan_statement import NDPlan, _row_index, graph_embedding


def _level_tables(plan, lv):
    nodes = plan.level_nodes(lv)
    s, b = plan.s[nodes], plan.b[nodes]
    return nodes, s, b, int(s.max(initial=0)), int(b.max(initial=0))


class Factor:
    """Device arrays of one factorisation in the layout of include/largesteps_hip.h (ls_direct_arrays) + per-node offsets."""
    pass


def _sparse_leaf_tables(plan, sparse, q_nodes, rr, bb, v32):
    """CSR lists of the sparse leaves' off-diagonal blocks A_bs, in both row orders (host, numpy). q_nodes / rr / bb / v32:
    node id, own-row index, boundary index and value of every matrix entry (own row, ancestor column) of a leaf."""
    keep = sparse[q_nodes]
    q_nodes, rr, bb, v32 = q_nodes[keep], rr[keep], bb[keep], v32[keep]
    leaves = np.flatnonzero(sparse)
    nb, ns = plan.b[leaves], plan.s[leaves]
    slot = np.zeros(plan.n_nodes + 1, dtype=np.int64)
    slot[leaves] = np.arange(leaves.shape[0])
    lq = slot[q_nodes]
    out_ptr, out_idx, out_val, offs = [], [], [], []
    ent_base, ptr_base = 0, 0
    for rows_of, r_idx, c_idx in ((nb, bb, rr), (ns, rr, bb)):          # boundary rows (up sweep), own rows (down sweep)
        rowbase = np.concatenate([[0], np.cumsum(rows_of)])
        grow = rowbase[lq] + r_idx
        order = np.lexsort((c_idx, grow))
        counts = np.bincount(grow, minlength=int(rowbase[-1]))
        E = np.concatenate([[0], np.cumsum(counts)]) + ent_base
        reps = rows_of + 1
        first = np.cumsum(reps) - reps
        take = np.repeat(rowbase[:-1], reps) + (np.arange(int(reps.sum())) - np.repeat(first, reps))
        out_ptr.append(E[take])
        out_idx.append(c_idx[order])
        out_val.append(v32[order])
        off = np.full(plan.n_nodes + 1, -1, dtype=np.int64)
        off[leaves] = first + ptr_base
        offs.append(off)
        ent_base += order.shape[0]
        ptr_base += int(reps.sum())
    ent = np.zeros(ent_base, dtype=np.dtype([("val", np.float32), ("idx", np.int32)]))
    ent["val"] = np.concatenate(out_val) if ent_base else np.zeros(0, np.float32)
    ent["idx"] = np.concatenate(out_idx) if ent_base else np.zeros(0, np.int32)
    return np.concatenate(out_ptr).astype(np.int32), ent, offs[0], offs[1]


def factorize(plan, rowptr, col, val, device, sparse_leaves=True, tier_levels=0):
    """Numeric factorisation on `device`. rowptr/col: host int arrays (CSR pattern, original numbering), val: device
    fp32 tensor in CSR order. Returns a Factor: flat device tensors in the layout of include/largesteps_hip.h.
    tier_levels: the deepest `tier_levels` tree levels are stored in the tier kernels' layouts (csrc/nd_tier.h): dense
    nodes quad-interleaved along the reduction (u4 / d4), and -- with sparse_leaves -- leaves with 1 <= s <= 64 as one
    packed triangle of F_ss^-1 plus the sparse matrix block A_bs instead of the dense inverse and W."""
    V, A, top = plan.V, plan.arity, plan.levels - 1
    tier_levels = max(0, min(int(tier_levels), plan.levels))
    if tier_levels == 0:
        sparse_leaves = False
    sparse = np.zeros(plan.n_nodes + 1, dtype=bool)
    if sparse_leaves:
        ln = plan.level_nodes(top)
        if (plan.s[ln] <= 64).all():                 # all leaves alike: the tier kernel has no mixed leaf level
            sparse[ln] = plan.s[ln] >= 1              # (an empty leaf has no boundary either: it is skipped altogether)
    quad = np.zeros(plan.n_nodes + 1, dtype=bool)
    if tier_levels:
        quad[plan.level_off[plan.levels - tier_levels]:] = True
        quad &= ~sparse
        quad[0] = False
    s4, b4 = (plan.s + 3) & ~3, (plan.b + 3) & ~3
    ss = np.where(sparse | quad, 0, plan.s * plan.s)
    sb_ = np.where(sparse | quad, 0, plan.s * plan.b)
    u4_len = np.where(quad, s4 * plan.b, 0)
    d4_len = np.where(quad, (s4 + b4) * plan.s, 0)
    u4_off = np.concatenate([[0], np.cumsum(u4_len)])[:-1]
    d4_off = np.concatenate([[0], np.cumsum(d4_len)])[:-1]
    u4 = torch.zeros(max(int(u4_len.sum()), 4), dtype=torch.float32, device=device)
    d4 = torch.zeros(max(int(d4_len.sum()), 4), dtype=torch.float32, device=device)
    finv_off = np.concatenate([[0], np.cumsum(ss)])[:-1]
    w_off = np.concatenate([[0], np.cumsum(sb_)])[:-1]
    finv_size, w_size = int(ss.sum()), int(sb_.sum())
    tri_len = np.where(sparse, (plan.s * (plan.s + 1) // 2 + 3) & ~3, 0)
    tri_off = np.where(sparse, np.concatenate([[0], np.cumsum(tri_len)])[:-1], -1)
    tri = torch.zeros(max(int(tri_len.sum()), 4), dtype=torch.float32, device=device)
    rows = _row_index(np.asarray(rowptr).astype(np.int64))
    prow, pcol = plan.inv[rows], plan.inv[np.asarray(col).astype(np.int64)]
    node = plan.node_of_new[prow]
    own_end = plan.own_start + plan.s
    is_own = (pcol >= plan.own_start[node]) & (pcol < own_end[node])
    is_up = pcol >= own_end[node]
    k_node = np.repeat(np.arange(plan.n_nodes + 1), plan.b)
    keys = k_node * V + plan.bnd
    up_pos = np.zeros(prow.shape[0], dtype=np.int64)
    if is_up.any():
        at = np.searchsorted(keys, node[is_up] * V + pcol[is_up])
        up_pos[is_up] = at - plan.bnd_off[node[is_up]]
    r_loc = prow - plan.own_start[node]
    c_loc = pcol - plan.own_start[node]
    level = plan.level_of[node]
    # (+ 4 floats: the level kernels read rows with unaligned 16-byte loads and may touch 12 bytes behind the last entry, see the header)
    val64 = val.to(torch.float64)
    finv = torch.zeros(max(finv_size, 1) + 4, dtype=torch.float32, device=device)
    wf = torch.zeros(max(w_size, 1) + 4, dtype=torch.float32, device=device)
    wb = torch.zeros(max(w_size, 1) + 4, dtype=torch.float32, device=device)
    sp_ptr = np.zeros(1, dtype=np.int32)
    sp_ent = np.zeros(0, dtype=np.dtype([("val", np.float32), ("idx", np.int32)]))
    spb_off = np.full(plan.n_nodes + 1, -1, dtype=np.int64)
    sps_off = np.full(plan.n_nodes + 1, -1, dtype=np.int64)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)   # noqa: E731
    U_child = None
    child_B = 0
    for lv in range(top, -1, -1):
        nodes, s, b, S, B = _level_tables(plan, lv)
        n, m = nodes.shape[0], S + B
        first = int(nodes[0])
        if m == 0:
            U_child, child_B = None, 0
            continue
        stride = m + 1                                   # index m = dummy row / column (padding sink)
        F = torch.zeros((n, stride, stride), dtype=torch.float64, device=device)
        s_t, b_t = dev(s), dev(b)
        arS = torch.arange(S, device=device)
        if S:
            pad = (arS[None, :] >= s_t[:, None]).to(torch.float64)        # identity on the padded own rows
            F[:, :S, :S] += torch.diag_embed(pad)
        sel = level == lv
        q = node[sel] - first
        e_own = is_own[sel]
        e_up = is_up[sel]
        idx_e = np.flatnonzero(sel)
        flat = F.view(-1)
        if e_own.any():
            fi = q[e_own] * stride * stride + r_loc[sel][e_own] * stride + c_loc[sel][e_own]
            flat[dev(fi)] = val64[dev(idx_e[e_own])]
        if lv == top and sparse.any():
            val_host = val.detach().cpu().numpy()
            sp_ptr, sp_ent, spb_off, sps_off = _sparse_leaf_tables(plan, sparse, node[sel][e_up], r_loc[sel][e_up], up_pos[sel][e_up],
                                                                    val_host[idx_e[e_up]])
        if e_up.any():
            rr = r_loc[sel][e_up]
            bb = S + up_pos[sel][e_up]
            v = val64[dev(idx_e[e_up])]
            flat[dev(q[e_up] * stride * stride + bb * stride + rr)] = v
            flat[dev(q[e_up] * stride * stride + rr * stride + bb)] = v
        if lv < top and U_child is not None and child_B > 0:
            ch = plan.level_nodes(lv + 1)
            bc = plan.b[ch]
            P = np.full((ch.shape[0], child_B), m, dtype=np.int64)
            kk = np.arange(child_B)[None, :]
            valid = kk < bc[:, None]
            src = (plan.bnd_off[ch][:, None] + kk)[valid]
            pp = plan.ppos[src]
            par_s = np.repeat(plan.s[plan.parent[ch]], bc)
            P[valid] = np.where(pp < par_s, pp, pp - par_s + S)
            P_t = dev(P)
            ar_n = torch.arange(n, device=device)[:, None, None]
            for c in range(A):                       # child c of every parent: a strided slice of the child level
                Pq = P_t[c::A]
                F[ar_n, Pq[:, :, None], Pq[:, None, :]] += U_child[c::A]
        Fbs = F[:, S:m, :S]
        Fbb = F[:, S:m, S:m]
        if S:
            L = torch.linalg.cholesky(F[:, :S, :S])
            eye = torch.eye(S, dtype=torch.float64, device=device).expand(n, S, S)
            Fi = torch.cholesky_solve(eye, L)
            Fi = 0.5 * (Fi + Fi.transpose(1, 2))
            W = Fbs @ Fi                                                     # (n, B, S)
            U = Fbb - W @ Fbs.transpose(1, 2) if B else Fbb.clone()
            arB = torch.arange(B, device=device)
            ms = arS[None, :] < s_t[:, None]                                 # (n, S)
            mb = arB[None, :] < b_t[:, None]                                 # (n, B)
            dense_t = dev(~(sparse[nodes] | quad[nodes]))                    # (n,) nodes stored as plain dense blocks
            msd = ms & dense_t[:, None]
            f0 = int(finv_off[first])
            cnt = int(ss[nodes].sum())
            finv[f0:f0 + cnt] = Fi[msd[:, :, None] & msd[:, None, :]].to(torch.float32)
            if B:
                w0 = int(w_off[first])
                wc = int(sb_[nodes].sum())
                wb[w0:w0 + wc] = W[mb[:, :, None] & msd[:, None, :]].to(torch.float32)
                wf[w0:w0 + wc] = W.transpose(1, 2)[msd[:, :, None] & mb[:, None, :]].to(torch.float32)
            if quad[nodes].any():                                            # quad-interleaved streams of the dense tier nodes
                q_t = dev(quad[nodes])
                msq, mbq = ms & q_t[:, None], mb & q_t[:, None]
                s_n, b_n, s4_n = s_t.to(torch.int64), b_t.to(torch.int64), dev(s4[nodes])
                uo, do = dev(u4_off[nodes]), dev(d4_off[nodes])
                if B:
                    at = (mbq[:, :, None] & msq[:, None, :]).nonzero()        # (n, i, j): W[i][j]
                    n_, i_, j_ = at[:, 0], at[:, 1], at[:, 2]
                    w32 = W[n_, i_, j_].to(torch.float32)
                    u4[uo[n_] + ((j_ // 4) * b_n[n_] + i_) * 4 + j_ % 4] = w32
                    t_ = s4_n[n_] + i_                                        # W^T behind the padded Finv columns
                    d4[do[n_] + ((t_ // 4) * s_n[n_] + j_) * 4 + t_ % 4] = w32
                at = (msq[:, :, None] & msq[:, None, :]).nonzero()            # (n, j, t): Finv[j][t]
                n_, j_, t_ = at[:, 0], at[:, 1], at[:, 2]
                d4[do[n_] + ((t_ // 4) * s_n[n_] + j_) * 4 + t_ % 4] = Fi[n_, j_, t_].to(torch.float32)
            if sparse[nodes].any():                                          # packed lower triangles, one 16-byte aligned block per leaf
                mss = ms & dev(sparse[nodes])[:, None]
                low = torch.tril(torch.ones((S, S), dtype=torch.bool, device=device))
                pick = mss[:, :, None] & mss[:, None, :] & low[None]
                at = pick.nonzero()
                dest = dev(tri_off[nodes])[at[:, 0]] + at[:, 1] * (at[:, 1] + 1) // 2 + at[:, 2]
                tri[dest] = Fi[pick].to(torch.float32)
        else:
            U = Fbb.clone()
        U_child, child_B = U, B
        del F
    fac = Factor()
    fac.finv, fac.wf, fac.wb, fac.tri, fac.u4, fac.d4 = finv, wf, wb, tri, u4, d4
    fac.quad, fac.tier_levels = quad, tier_levels
    fac.finv_off_all = np.where(quad, d4_off, finv_off)
    fac.w_off_all = np.where(quad, u4_off, w_off)
    fac.sp_ptr = torch.from_numpy(np.ascontiguousarray(sp_ptr)).to(device)
    fac.sp_ent = torch.from_numpy(np.ascontiguousarray(sp_ent).view(np.int32).reshape(-1)).to(device) if sp_ent.shape[0] else \
        torch.zeros(2, dtype=torch.int32, device=device)
    fac.n_sp_ptr, fac.n_sp_ent = int(sp_ptr.shape[0]), int(sp_ent.shape[0])
    fac.finv_off, fac.w_off, fac.tri_off, fac.spb_off, fac.sps_off, fac.sparse = finv_off, w_off, tri_off, spb_off, sps_off, sparse
    return fac


class _Arrays(ctypes.Structure):
    """ls_direct_arrays of include/largesteps_hip.h"""
    _fields_ = [("V", ctypes.c_int64), ("levels", ctypes.c_int32), ("arity", ctypes.c_int32), ("h_nodes", ctypes.c_void_p),
                ("h_perm", ctypes.c_void_p), ("h_ppos", ctypes.c_void_p), ("n_bnd", ctypes.c_int64), ("h_push_ptr", ctypes.c_void_p),
                ("h_push_tgt", ctypes.c_void_p), ("n_front", ctypes.c_int64), ("d_finv", ctypes.c_void_p), ("d_wf", ctypes.c_void_p),
                ("d_wb", ctypes.c_void_p), ("d_u4", ctypes.c_void_p), ("d_d4", ctypes.c_void_p), ("d_tri", ctypes.c_void_p), ("d_sp_ptr", ctypes.c_void_p), ("d_sp_ent", ctypes.c_void_p),
                ("n_sp_ptr", ctypes.c_int64), ("n_sp_ent", ctypes.c_int64), ("shard_rank", ctypes.c_int32), ("shard_count", ctypes.c_int32)]


NODE_COLS = 12


class DirectHandle:
    """Owns the native ls_direct handle and the device arrays it points into."""

    def __init__(self, plan, fac, device):
        self.plan, self.fac, self.device = plan, fac, device
        nodes = np.zeros((plan.n_nodes + 1, NODE_COLS), dtype=np.int64)
        nodes[:, 0], nodes[:, 1], nodes[:, 2], nodes[:, 3] = plan.s, plan.b, plan.own_start, plan.bnd_off
        nodes[:, 4], nodes[:, 5], nodes[:, 6], nodes[:, 7] = plan.front_off, fac.finv_off_all, fac.w_off_all, plan.parent
        nodes[:, 8], nodes[:, 9], nodes[:, 10], nodes[:, 11] = fac.tri_off, fac.spb_off, fac.sps_off, fac.quad
        perm32 = plan.perm.astype(np.int32)
        ppos32 = plan.ppos.astype(np.int32)
        ptr32, tgt32 = plan.push_ptr.astype(np.int32), plan.push_tgt.astype(np.int32)
        as_p = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
        dp = lambda t: ctypes.c_void_p(t.data_ptr())         # noqa: E731
        arr = _Arrays(plan.V, plan.levels, plan.arity, as_p(nodes), as_p(perm32), as_p(ppos32), ppos32.shape[0], as_p(ptr32),
                      as_p(tgt32), ptr32.shape[0] - 1, dp(fac.finv), dp(fac.wf), dp(fac.wb), dp(fac.u4), dp(fac.d4), dp(fac.tri), dp(fac.sp_ptr),
                      dp(fac.sp_ent), fac.n_sp_ptr, fac.n_sp_ent, 0, 1)
        self._h = ctypes.c_void_p(None)
        with torch.cuda.device(device):
            _native.check(_native.lib().ls_direct_create(ctypes.byref(arr), device.index, _native.stream_of(device),
                                                         ctypes.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _native.lib().ls_direct_destroy(h)
            except Exception:
                pass
            self._h = None

    def solve(self, b, x):
        k = b.shape[1]
        with torch.cuda.device(self.device):
            _native.check(_native.lib().ls_direct_solve(self._h, _native.ptr(b), _native.ptr(x), k, _native.stream_of(self.device)))

    def set_option(self, name, value):
        _native.check(_native.lib().ls_direct_set(self._h, name.encode(), int(value)))

    def info(self):
        fe, nl = ctypes.c_int64(0), ctypes.c_int(0)
        ms = (ctypes.c_double * 3)()
        _native.check(_native.lib().ls_direct_info(self._h, ctypes.byref(fe), ctypes.byref(nl), ms))
        return dict(factor_entries=fe.value, launches=nl.value, up_ms=ms[0], down_ms=ms[1])


def build(csr, leaf_size=64, arity=4, max_front=8000, max_entries=3_000_000_000, max_level_bytes=48e9, sparse_leaves=None):
    """Plan + factorisation + native handle for the CSR side car of a matrix. Vertex positions come from compute_matrix;
    a matrix built elsewhere gets graph-distance pseudo-positions (nested.graph_embedding) if it is symmetric. Returns
    None when the mesh does not dissect well enough for this solver (front too large for LDS / factor too large)."""
    import time
    t0 = time.perf_counter()
    rowptr, col = csr.rowptr.cpu().numpy(), csr.col.cpu().numpy()
    if csr.positions is not None:
        positions = csr.positions.cpu().numpy()
    elif csr.symmetric:
        positions = graph_embedding(rowptr, col, csr.V)
    else:
        return None
    plan = NDPlan.build(rowptr, col, positions, leaf_size=leaf_size, arity=arity)
    t1 = time.perf_counter()
    if int((plan.s + plan.b).max()) > max_front or plan.factor_entries > max_entries:
        return None
    for lv in range(plan.levels):                      # the factorisation pads a level to its largest front (fp64)
        nodes, s, b, S, B = _level_tables(plan, lv)
        if nodes.shape[0] * float(S + B + 1) ** 2 * 8 * 3 > max_level_bytes:
            return None
    if sparse_leaves is None:
        sparse_leaves = not os.environ.get("LS_ND_DENSE_LEAVES")
    # the deepest LS_ND_TIER_H (default 3) levels go into the tier kernels' layouts: one launch per sweep for all of them.
    # A tier whose subtrees need more LDS than a workgroup has is refused by the native side: retry one level lower.
    tier = max(0, min(plan.levels, 6, int(os.environ.get("LS_ND_TIER_H", "3"))))
    while True:
        fac = factorize(plan, rowptr, col, csr.val, csr.device, sparse_leaves=sparse_leaves, tier_levels=tier)
        torch.cuda.synchronize(csr.device) if csr.device.type == "cuda" else None
        t2 = time.perf_counter()
        try:
            handle = DirectHandle(plan, fac, csr.device)
            break
        except RuntimeError as e:
            if tier == 0 or "does not fit" not in str(e):
                raise
            tier -= 1
    handle.timings = dict(plan_seconds=t1 - t0, factor_seconds=t2 - t1, handle_seconds=time.perf_counter() - t2)
    return handle
