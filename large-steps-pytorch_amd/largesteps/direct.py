"""
Factor-once / re-solve direct solver on the MI355X: numeric multifrontal factorisation of a nested-dissection plan
(largesteps/nested.py) and the handle of the native re-solve kernels (csrc/direct.hip, C ABI ls_direct_*).

Replaces: largesteps/solvers.py:26-39 of the reference (CholeskySolver: cholespy / CHOLMOD factorisation in the
constructor, two triangular solves per call). Here the constructor
  1. builds the elimination tree from the vertex positions (host, numpy, integer work only),
  2. factorises level by level ON THE DEVICE: every tree level is one batch of dense fronts (padded to the level's
     largest front), assembled by index arithmetic and factorised in fp64 with torch.linalg (rocSOLVER / rocBLAS):
     Finv = F_ss^-1, W = F_bs Finv, U = F_bb - W F_sb -> parent,
  3. packs Finv and W (twice: both sweep layouts) in fp32 and hands the device pointers to ls_direct_create.
Every solve afterwards is 2 * levels hand-written HIP launches that read the factor once (W twice).
"""
import ctypes

import numpy as np
import torch

from . import _native
from .nested import NDPlan, _row_index, graph_embedding


def _level_tables(plan, lv):
    nodes = plan.level_nodes(lv)
    s, b = plan.s[nodes], plan.b[nodes]
    return nodes, s, b, int(s.max(initial=0)), int(b.max(initial=0))


def factorize(plan, rowptr, col, val, device):
    """Numeric factorisation on `device`. rowptr/col: host int arrays (CSR pattern, original numbering), val: device
    fp32 tensor in CSR order. Returns (finv, wf, wb): flat fp32 device tensors in the layout of include/largesteps_hip.h."""
    V, A, top = plan.V, plan.arity, plan.levels - 1
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
    val64 = val.to(torch.float64)
    finv = torch.zeros(max(plan.finv_size, 1), dtype=torch.float32, device=device)
    wf = torch.zeros(max(plan.w_size, 1), dtype=torch.float32, device=device)
    wb = torch.zeros(max(plan.w_size, 1), dtype=torch.float32, device=device)
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
            f0 = int(plan.finv_off[first])
            cnt = int((s * s).sum())
            finv[f0:f0 + cnt] = Fi[ms[:, :, None] & ms[:, None, :]].to(torch.float32)
            if B:
                w0 = int(plan.w_off[first])
                wc = int((s * b).sum())
                wb[w0:w0 + wc] = W[mb[:, :, None] & ms[:, None, :]].to(torch.float32)
                wf[w0:w0 + wc] = W.transpose(1, 2)[ms[:, :, None] & mb[:, None, :]].to(torch.float32)
        else:
            U = Fbb.clone()
        U_child, child_B = U, B
        del F
    return finv, wf, wb


class DirectHandle:
    """Owns the native ls_direct handle and the device arrays it points into."""

    def __init__(self, plan, finv, wf, wb, device):
        self.plan, self.finv, self.wf, self.wb, self.device = plan, finv, wf, wb, device
        nodes = np.zeros((plan.n_nodes + 1, 8), dtype=np.int64)
        nodes[:, 0], nodes[:, 1], nodes[:, 2], nodes[:, 3] = plan.s, plan.b, plan.own_start, plan.bnd_off
        nodes[:, 4], nodes[:, 5], nodes[:, 6], nodes[:, 7] = plan.front_off, plan.finv_off, plan.w_off, plan.parent
        perm32 = plan.perm.astype(np.int32)
        ppos32 = plan.ppos.astype(np.int32)
        ptr32, tgt32 = plan.push_ptr.astype(np.int32), plan.push_tgt.astype(np.int32)
        as_p = lambda a: a.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
        self._h = ctypes.c_void_p(None)
        with torch.cuda.device(device):
            _native.check(_native.lib().ls_direct_create(plan.V, plan.levels, plan.arity, as_p(nodes), as_p(perm32), as_p(ppos32),
                                                         ppos32.shape[0], as_p(ptr32), as_p(tgt32), ptr32.shape[0] - 1,
                                                         _native.ptr(finv), _native.ptr(wf), _native.ptr(wb), device.index,
                                                         _native.stream_of(device), ctypes.byref(self._h)))

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
        return dict(factor_entries=fe.value, launches=nl.value, up_ms=ms[0], down_ms=ms[1], perm_ms=ms[2])


def build(csr, leaf_size=64, arity=4, max_front=8000, max_entries=3_000_000_000, max_level_bytes=48e9):
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
    finv, wf, wb = factorize(plan, rowptr, col, csr.val, csr.device)
    torch.cuda.synchronize(csr.device) if csr.device.type == "cuda" else None
    t2 = time.perf_counter()
    handle = DirectHandle(plan, finv, wf, wb, csr.device)
    handle.timings = dict(plan_seconds=t1 - t0, factor_seconds=t2 - t1, handle_seconds=time.perf_counter() - t2)
    return handle
