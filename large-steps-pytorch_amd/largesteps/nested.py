"""
Nested-dissection plan of the factor-once / re-solve direct solver (csrc/direct.hip) -- host-side symbolic analysis,
numpy only. This is the MI355X answer to the reference's default method (solvers.py:26-39: cholespy / CHOLMOD
factorisation, then two sparse triangular solves per call).

Why not a level-scheduled sparse triangular solve: its dependency chains are thousands of levels long on a mesh. A
nested-dissection elimination tree has log2(V / leaf) levels, and with the *multifrontal* formulation every level is
one batch of small dense matrix-vector products:

    tree      geometric bisection of the vertex positions (median split along the longest axis); the separator of a
              domain = its side-0 vertices that touch side 1. Node i owns its separator (leaves: their whole domain),
              children 2i, 2i+1 (heap numbering), D = depth of the leaves.
    ordering  deepest level first, root last: the vertices of a node are one contiguous range of the new numbering.
    front i   [own_i | bnd_i], bnd_i = the ancestors' vertices the subtree of i touches (filled graph), sorted.
    factor    F_i = A[front_i, front_i] restricted to entries with a row or column in own_i, plus the children's Schur
              complements U_c (extend-add). Stored per node, fp32:   Finv_i = F_ss^-1  (s x s),
              W_i = F_bs F_ss^-1 (b x s);  U_i = F_bb - W_i F_sb goes to the parent.
    solve     up   (leaves -> root): b'_s = b_s - (children's updates at own_i);  upd_i = W_i b'_s + (children's
                   updates at bnd_i)                       [pull: no atomics, one launch per level]
              down (root -> leaves): x_s = Finv_i b'_s - W_i^T x[bnd_i]

Per solve the GPU reads every W twice and every Finv once -- and launches 2 D + 1 kernels.
"""
import numpy as np


def _row_index(rowptr):
    return np.repeat(np.arange(rowptr.shape[0] - 1, dtype=np.int64), np.diff(rowptr))


class NDPlan:
    """Arrays indexed by heap node id carry one unused slot 0 (root = 1, children of i = 2i, 2i+1)."""

    @staticmethod
    def build(rowptr, col, positions, leaf_size=48):
        rowptr = np.asarray(rowptr).astype(np.int64)
        col = np.asarray(col).astype(np.int64)
        pos = np.asarray(positions, dtype=np.float64)
        V = rowptr.shape[0] - 1
        if V <= 0 or pos.shape[0] != V:
            raise ValueError("NDPlan.build: positions must have one row per matrix row")
        D = 0
        while (V >> D) > leaf_size:
            D += 1
        rows = _row_index(rowptr)
        node = np.ones(V, dtype=np.int64)
        fixed = np.zeros(V, dtype=bool)
        side_of = np.zeros(V, dtype=np.int8)
        for _ in range(D):
            idx = np.flatnonzero(~fixed)
            if idx.shape[0]:
                dom = node[idx]
                o1 = np.argsort(dom, kind="stable")
                idx, dom = idx[o1], dom[o1]
                starts = np.flatnonzero(np.concatenate([[True], dom[1:] != dom[:-1]]))
                counts = np.diff(np.concatenate([starts, [idx.shape[0]]]))
                seg = np.repeat(np.arange(starts.shape[0]), counts)
                p = pos[idx]
                ext = np.maximum.reduceat(p, starts, axis=0) - np.minimum.reduceat(p, starts, axis=0)
                key = p[np.arange(idx.shape[0]), np.argmax(ext, axis=1)[seg]]
                o2 = np.lexsort((key, seg))                       # by domain, then along the domain's longest axis
                rank = np.arange(idx.shape[0]) - starts[seg]      # seg is already sorted: o2 keeps the segments
                side_of[idx[o2]] = (rank >= (counts[seg] // 2)).astype(np.int8)
            live = ~fixed
            m = live[rows] & live[col] & (node[rows] == node[col]) & (side_of[rows] == 0) & (side_of[col] == 1)
            sep = np.zeros(V, dtype=bool)
            sep[rows[m]] = True
            fixed |= sep
            move = ~fixed
            node[move] = 2 * node[move] + side_of[move]
        return NDPlan._finish(V, D, rows, col, node)

    @staticmethod
    def _finish(V, D, rows, col, node):
        n_nodes = (1 << (D + 1)) - 1
        level_of = np.zeros(n_nodes + 1, dtype=np.int64)
        for lv in range(D + 1):
            level_of[1 << lv:1 << (lv + 1)] = lv
        perm = np.lexsort((np.arange(V), node, -level_of[node]))      # new -> old
        inv = np.empty(V, dtype=np.int64)
        inv[perm] = np.arange(V)
        s = np.bincount(node, minlength=n_nodes + 1).astype(np.int64)
        node_order = np.concatenate([np.arange(1 << lv, 1 << (lv + 1)) for lv in range(D, -1, -1)])
        own_start = np.zeros(n_nodes + 1, dtype=np.int64)
        own_start[node_order] = np.cumsum(s[node_order]) - s[node_order]
        own_end = own_start + s
        nn = node[perm]                                               # node of every new id
        prow, pcol = inv[rows], inv[col]
        up = pcol >= own_end[nn[prow]]                                # entries that reach an ancestor
        a_node, a_w = nn[prow[up]], pcol[up]
        a_level = level_of[a_node]
        # boundary sets, deepest level first (a node's set needs its children's)
        keys_by_level = [np.empty(0, np.int64)] * (D + 1)
        for lv in range(D, 0, -1):
            m = a_level == lv
            keys = a_node[m] * V + a_w[m]
            if lv < D:
                ck = keys_by_level[lv + 1]
                c_node, c_w = ck // V, ck % V
                par = c_node >> 1
                keep = c_w >= own_end[par]
                keys = np.concatenate([keys, par[keep] * V + c_w[keep]])
            keys_by_level[lv] = np.unique(keys)
        keys = np.concatenate(keys_by_level[1:]) if D > 0 else np.empty(0, np.int64)    # sorted by (node, w)
        k_node, bnd = keys // V, keys % V
        b = np.bincount(k_node, minlength=n_nodes + 1).astype(np.int64)
        bnd_off = np.concatenate([[0], np.cumsum(b)])[:-1]
        front_off = np.concatenate([[0], np.cumsum(s + b)])[:-1]
        # position of every boundary vertex of node c in its parent's front, and the inverse (pull) maps
        par = k_node >> 1
        in_own = bnd < own_end[par]
        assert (bnd[in_own] >= own_start[par[in_own]]).all(), "separator property violated"
        ppos = np.where(in_own, bnd - own_start[par], 0)
        if (~in_own).any():
            at = np.searchsorted(keys, par[~in_own] * V + bnd[~in_own])
            assert (keys[at] == par[~in_own] * V + bnd[~in_own]).all(), "child boundary not contained in parent front"
            ppos[~in_own] = s[par[~in_own]] + at - bnd_off[par[~in_own]]
        local_k = np.arange(keys.shape[0]) - bnd_off[k_node]
        maps = np.full((2, int((s + b).sum())), -1, dtype=np.int32)
        maps[k_node & 1, front_off[par] + ppos] = local_k
        plan = NDPlan()
        plan.V, plan.D, plan.n_nodes = int(V), int(D), int(n_nodes)
        plan.perm, plan.inv = perm, inv
        plan.s, plan.b, plan.own_start = s, b, own_start
        plan.bnd, plan.bnd_off, plan.front_off = bnd.astype(np.int64), bnd_off, front_off
        plan.ppos = ppos.astype(np.int64)
        plan.map0, plan.map1 = maps[0], maps[1]
        plan.finv_off = np.concatenate([[0], np.cumsum(s * s)])[:-1]
        plan.w_off = np.concatenate([[0], np.cumsum(s * b)])[:-1]
        plan.finv_size, plan.w_size = int((s * s).sum()), int((s * b).sum())
        plan.node_of_new = nn
        return plan

    def level_nodes(self, lv):
        return np.arange(1 << lv, 1 << (lv + 1))

    @property
    def factor_entries(self):
        """fp32 numbers one solve reads: every W twice (up and down sweep), every Finv once"""
        return 2 * self.w_size + self.finv_size

    # ---- numpy statement of the numeric factorisation and of the two sweeps (CPU tests; dense per node) ----------
    def factor_reference(self, rowptr, col, val):
        """Returns (finv, w): flat fp64 arrays; node i: Finv = finv[finv_off:+s*s].reshape(s,s), W = w[w_off:+b*s].reshape(b,s)."""
        rows = _row_index(np.asarray(rowptr).astype(np.int64))
        prow, pcol = self.inv[rows], self.inv[np.asarray(col).astype(np.int64)]
        val = np.asarray(val, dtype=np.float64)
        order = np.argsort(self.node_of_new[prow], kind="stable")
        prow, pcol, val = prow[order], pcol[order], val[order]
        ent_off = np.concatenate([[0], np.cumsum(np.bincount(self.node_of_new[prow], minlength=self.n_nodes + 1))])
        finv, w = np.zeros(self.finv_size), np.zeros(self.w_size)
        U = {}
        for lv in range(self.D, -1, -1):
            for i in self.level_nodes(lv):
                s, b, o = int(self.s[i]), int(self.b[i]), int(self.own_start[i])
                F = np.zeros((s + b, s + b))
                e0, e1 = ent_off[i], ent_off[i + 1]
                r, c, v = prow[e0:e1] - o, pcol[e0:e1], val[e0:e1]
                own = (c >= o) & (c < o + s)
                F[r[own], c[own] - o] = v[own]
                upm = c >= o + s
                bi = s + np.searchsorted(self.bnd[self.bnd_off[i]:self.bnd_off[i] + b], c[upm])
                F[bi, r[upm]] = v[upm]
                F[r[upm], bi] = v[upm]
                if lv < self.D:
                    for ch in (2 * i, 2 * i + 1):
                        bc = int(self.b[ch])
                        if bc:
                            pp = self.ppos[self.bnd_off[ch]:self.bnd_off[ch] + bc]
                            F[np.ix_(pp, pp)] += U.pop(ch)
                if s:
                    Fi = np.linalg.inv(F[:s, :s])
                    Fi = 0.5 * (Fi + Fi.T)
                    Wi = F[s:, :s] @ Fi
                    finv[self.finv_off[i]:self.finv_off[i] + s * s] = Fi.reshape(-1)
                    w[self.w_off[i]:self.w_off[i] + b * s] = Wi.reshape(-1)
                    U[i] = F[s:, s:] - Wi @ F[:s, s:]
                else:
                    U[i] = F[s:, s:]
        return finv, w

    def solve_reference(self, finv, w, rhs):
        """rhs in the ORIGINAL numbering, (V, k); returns x in the original numbering."""
        bp = np.asarray(rhs, dtype=np.float64)[self.perm]
        k = bp.shape[1]
        upd = np.zeros((self.bnd.shape[0], k))
        bprime = bp.copy()

        def pulled(i, lo, hi):          # children's updates at front positions [lo, hi) of node i
            out = np.zeros((hi - lo, k))
            if i * 2 <= self.n_nodes:
                for ch, mp in ((2 * i, self.map0), (2 * i + 1, self.map1)):
                    m = mp[self.front_off[i] + lo:self.front_off[i] + hi]
                    out[m >= 0] += upd[self.bnd_off[ch] + m[m >= 0]]
            return out

        for lv in range(self.D, -1, -1):
            for i in self.level_nodes(lv):
                s, b, o = int(self.s[i]), int(self.b[i]), int(self.own_start[i])
                bprime[o:o + s] = bp[o:o + s] - pulled(i, 0, s)
                W = w[self.w_off[i]:self.w_off[i] + b * s].reshape(b, s)
                upd[self.bnd_off[i]:self.bnd_off[i] + b] = W @ bprime[o:o + s] + pulled(i, s, s + b)
        x = np.zeros_like(bp)
        for lv in range(0, self.D + 1):
            for i in self.level_nodes(lv):
                s, b, o = int(self.s[i]), int(self.b[i]), int(self.own_start[i])
                Fi = finv[self.finv_off[i]:self.finv_off[i] + s * s].reshape(s, s)
                W = w[self.w_off[i]:self.w_off[i] + b * s].reshape(b, s)
                x[o:o + s] = Fi @ bprime[o:o + s] - W.T @ x[self.bnd[self.bnd_off[i]:self.bnd_off[i] + b]]
        out = np.empty_like(x)
        out[self.perm] = x
        return out
