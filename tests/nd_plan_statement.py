"""
TEST CODE -- numpy statement of the symbolic analysis the product does in C++ (csrc/nd_plan.cpp): nested-dissection plan
of the factor-once / re-solve direct solver (csrc/direct.hip). This is the MI355X answer to the reference's default method (solvers.py:26-39: cholespy / CHOLMOD
factorisation, then two sparse triangular solves per call).

Why not a level-scheduled sparse triangular solve: its dependency chains are thousands of levels long on a mesh. A
nested-dissection elimination tree has log2(V / leaf) levels, and with the *multifrontal* formulation every level is
one batch of small dense matrix-vector products:

    tree      geometric bisection of the vertex positions (median split along the longest axis); the separator of a
              domain = the end points of its cut edges on one side (the side with fewer of them). A tree node merges log2(arity) bisection rounds: it owns
              the separators of those rounds (leaves: their whole domain) and has `arity` children.
    ordering  deepest level first, root last: the vertices of a node are one contiguous range of the new numbering.
    front i   [own_i | bnd_i], bnd_i = the ancestors' vertices the subtree of i touches (filled graph), sorted.
    factor    F_i = A[front_i, front_i] restricted to entries with a row or column in own_i, plus the children's Schur
              complements U_c (extend-add). Stored per node, fp32:   Finv_i = F_ss^-1  (s x s),
              W_i = F_bs F_ss^-1 (b x s);  U_i = F_bb - W_i F_sb goes to the parent.
    solve     up   (leaves -> root): b'_s = b_s - (children's updates at own_i);  upd_i = W_i b'_s + (children's
                   updates at bnd_i)                       [no atomics, one launch per level]
              down (root -> leaves): x_s = Finv_i b'_s - W_i^T x[bnd_i]

Per solve the GPU reads every W twice and every Finv once -- one launch per upper tree level and sweep, one per sweep for the deepest levels.
"""
import numpy as np


def _row_index(rowptr):
    return np.repeat(np.arange(rowptr.shape[0] - 1, dtype=np.int64), np.diff(rowptr))


def _bfs(rowptr, col, V, start, dist):
    """level-synchronous BFS from `start` over the CSR pattern; fills dist (float64, -1 = unreached); returns the last vertex reached"""
    dist[start] = 0.0
    frontier = np.array([start], dtype=np.int64)
    last, d = start, 0.0
    while frontier.shape[0]:
        last = int(frontier[0])
        lens = rowptr[frontier + 1] - rowptr[frontier]
        total = int(lens.sum())
        if total == 0:
            break
        first = np.cumsum(lens) - lens
        nb = col[np.arange(total, dtype=np.int64) - np.repeat(first, lens) + np.repeat(rowptr[frontier], lens)]
        nb = np.unique(nb[dist[nb] < 0])
        d += 1.0
        dist[nb] = d
        frontier = nb
    return last


def graph_embedding(rowptr, col, V):
    """Pseudo-positions for a matrix that comes without vertex positions (not built by compute_matrix): graph distances
    from three mutually far landmarks (double-sweep BFS), per connected component; components are laid out side by side.
    Only the spatial ORDER these coordinates induce matters to the bisection -- level sets of a graph distance are
    separators as thin as the mesh allows."""
    rowptr = np.asarray(rowptr).astype(np.int64)
    col = np.asarray(col).astype(np.int64)
    pos = np.zeros((V, 3), dtype=np.float64)
    done = np.zeros(V, dtype=bool)
    offset = 0.0
    while not done.all():
        seed = int(np.flatnonzero(~done)[0])
        d0 = np.full(V, -1.0)
        p1 = _bfs(rowptr, col, V, seed, d0)
        comp = d0 >= 0
        d1 = np.full(V, -1.0)
        p2 = _bfs(rowptr, col, V, p1, d1)
        d2 = np.full(V, -1.0)
        _bfs(rowptr, col, V, p2, d2)
        p3 = int(np.argmax(np.where(comp, np.minimum(d1, d2), -1.0)))      # far from both landmarks
        d3 = np.full(V, -1.0)
        _bfs(rowptr, col, V, p3, d3)
        pos[comp, 0] = d1[comp] + offset
        pos[comp, 1] = d2[comp]
        pos[comp, 2] = d3[comp]
        offset += float(d1[comp].max()) + 2.0
        done |= comp
        if comp.sum() <= 2 and (~done).sum() > 4096:
            # a swarm of isolated vertices (unreferenced rows): no structure to find, lay the rest out in index order
            rest = np.flatnonzero(~done)
            pos[rest, 0] = offset + np.arange(rest.shape[0])
            break
    return pos


class NDPlan:
    """Node ids are 1-based and level-major: level l holds arity^l nodes starting at level_off[l]; node (l, q) has the
    children (l + 1, arity * q + c). Arrays indexed by node id carry one unused slot 0."""

    @staticmethod
    def build(rowptr, col, positions, leaf_size=48, arity=4, smooth=4):
        """smooth: the bisection runs on positions averaged `smooth` times over the mesh neighbours (only their spatial
        order matters here). On a rough surface -- noise amplitude above the edge length, as on scans -- a cutting
        plane through the raw positions crosses the surface in a fractal band and the separators stay thousands of
        vertices wide however small the domains get (250k-vertex noisy sphere: 1481 factor numbers per vertex; after
        2-5 averaging passes 215, the smooth plane's figure).
        arity (2, 4 or 8): every tree node merges log2(arity) rounds of bisection -- its own block is the union of
        the 1 + 2 + .. separators of those rounds, its children are the arity sub-domains. Fewer, fatter levels:
        the re-solve is latency bound (one dependent launch per level and sweep), so trading a few percent more
        factor entries for half (a third) of the launches pays. The leaf domains always form their own last level."""
        rowptr = np.asarray(rowptr).astype(np.int64)
        col = np.asarray(col).astype(np.int64)
        pos = np.asarray(positions, dtype=np.float64)
        V = rowptr.shape[0] - 1
        if V <= 0 or pos.shape[0] != V:
            raise ValueError("NDPlan.build: positions must have one row per matrix row")
        if arity not in (2, 4, 8):
            raise ValueError("NDPlan.build: arity must be 2, 4 or 8")
        m = int(np.log2(arity))
        D = 0
        while (V >> D) > leaf_size:
            D += 1
        D = -(-D // m) * m                                   # bisection rounds: a multiple of log2(arity)
        rows = _row_index(rowptr)
        if smooth > 0 and (np.diff(rowptr) > 0).all():
            cnt = np.diff(rowptr).astype(np.float64)[:, None]
            for _ in range(int(smooth)):                     # p_i <- mean of p over row i's columns (diagonal included)
                pos = np.add.reduceat(pos[col], rowptr[:-1], axis=0) / cnt
        one_way = rows < col
        er, ec = rows[one_way], col[one_way]                 # every undirected edge once
        node = np.ones(V, dtype=np.int64)                    # binary heap id of the domain a vertex lives in
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
                lo = np.minimum.reduceat(p, starts, axis=0)
                ext = np.maximum.reduceat(p, starts, axis=0) - lo
                ax = np.argmax(ext, axis=1)
                sel = np.arange(starts.shape[0])
                key = p[np.arange(idx.shape[0]), ax[seg]]
                # by domain, then along the domain's longest axis: one sort of (domain index + position fraction in [0, 1))
                frac = (key - lo[sel, ax][seg]) / np.maximum(ext[sel, ax], 1e-300)[seg]
                o2 = np.argsort(seg + np.minimum(frac, 1.0 - 1e-9), kind="stable")
                rank = np.arange(idx.shape[0]) - starts[seg]      # seg is already sorted: o2 keeps the segments
                side_of[idx[o2]] = (rank >= (counts[seg] // 2)).astype(np.int8)
            # edges that can still be cut: both ends live and in the same domain (domains only ever split, separator
            # vertices never come back: an edge that fails this once is dropped for good -- the list shrinks every round)
            keep = (node[er] == node[ec]) & ~fixed[er] & ~fixed[ec]
            er, ec = er[keep], ec[keep]
            s_r, s_c = side_of[er], side_of[ec]
            cut = s_r != s_c
            r_cut, c_cut, r0 = er[cut], ec[cut], s_r[cut] == 0
            # the end points of the cut edges on either side separate the domain: take the smaller set, per domain
            end0 = np.zeros(V, dtype=bool)
            end1 = np.zeros(V, dtype=bool)
            end0[np.where(r0, r_cut, c_cut)] = True
            end1[np.where(r0, c_cut, r_cut)] = True
            n_dom = int(node.max()) + 1
            use1 = np.bincount(node[end1], minlength=n_dom) < np.bincount(node[end0], minlength=n_dom)
            sep = np.where(use1[node], end1, end0)
            fixed |= sep
            move = ~fixed
            node[move] = 2 * node[move] + side_of[move]
        # binary (level lb, heap id) -> merged (level, index): separators of rounds m*l .. m*l + m - 1 form level l,
        # the leaf domains (binary level D) form the last level D / m
        lb = np.floor(np.log2(node)).astype(np.int64)
        lvl = np.where(lb == D, D // m, lb // m)
        anc = node >> np.where(lb == D, 0, lb - m * (lb // m))
        q = anc - (np.int64(1) << (m * lvl))
        return NDPlan._finish(V, D // m + 1, arity, rows, col, lvl, q)

    @staticmethod
    def _finish(V, levels, arity, rows, col, lvl, q):
        level_off = np.array([1 + (arity ** l - 1) // (arity - 1) for l in range(levels + 1)], dtype=np.int64)
        n_nodes = int(level_off[levels] - 1)
        node = level_off[lvl] + q                                        # node id of every vertex
        level_of = np.zeros(n_nodes + 1, dtype=np.int64)
        parent = np.zeros(n_nodes + 1, dtype=np.int64)
        child_ix = np.zeros(n_nodes + 1, dtype=np.int64)
        for l in range(levels):
            ids = np.arange(level_off[l], level_off[l + 1])
            level_of[ids] = l
            if l:
                parent[ids] = level_off[l - 1] + (ids - level_off[l]) // arity
                child_ix[ids] = (ids - level_off[l]) % arity
        perm = np.lexsort((np.arange(V), node, -level_of[node]))         # new -> old: deepest level first
        inv = np.empty(V, dtype=np.int64)
        inv[perm] = np.arange(V)
        s = np.bincount(node, minlength=n_nodes + 1).astype(np.int64)
        node_order = np.concatenate([np.arange(level_off[l], level_off[l + 1]) for l in range(levels - 1, -1, -1)])
        own_start = np.zeros(n_nodes + 1, dtype=np.int64)
        own_start[node_order] = np.cumsum(s[node_order]) - s[node_order]
        own_end = own_start + s
        nn = node[perm]                                                  # node of every new id
        prow, pcol = inv[rows], inv[col]
        up = pcol >= own_end[nn[prow]]                                   # entries that reach an ancestor
        a_node, a_w = nn[prow[up]], pcol[up]
        a_level = level_of[a_node]
        # boundary sets, deepest level first (a node's set needs its children's)
        keys_by_level = [np.empty(0, np.int64)] * levels
        for l in range(levels - 1, 0, -1):
            msk = a_level == l
            keys = a_node[msk] * V + a_w[msk]
            if l < levels - 1:
                ck = keys_by_level[l + 1]
                c_node, c_w = ck // V, ck % V
                par = parent[c_node]
                keep = c_w >= own_end[par]
                keys = np.concatenate([keys, par[keep] * V + c_w[keep]])
            keys_by_level[l] = np.unique(keys)
        keys = np.concatenate(keys_by_level[1:]) if levels > 1 else np.empty(0, np.int64)    # sorted by (node, w)
        k_node, bnd = keys // V, keys % V
        b = np.bincount(k_node, minlength=n_nodes + 1).astype(np.int64)
        bnd_off = np.concatenate([[0], np.cumsum(b)])[:-1]
        front_off = np.concatenate([[0], np.cumsum(s + b)])[:-1]
        # position of every boundary vertex of a node in its parent's front [own | boundary]
        par = parent[k_node]
        in_own = bnd < own_end[par]
        if not (bnd[in_own] >= own_start[par[in_own]]).all():
            raise ValueError("NDPlan: separator property violated")
        ppos = np.where(in_own, bnd - own_start[par], 0)
        if (~in_own).any():
            at = np.searchsorted(keys, par[~in_own] * V + bnd[~in_own])
            at = np.minimum(at, keys.shape[0] - 1)
            if not (keys[at] == par[~in_own] * V + bnd[~in_own]).all():
                raise ValueError("NDPlan: child boundary not contained in parent front")
            ppos[~in_own] = s[par[~in_own]] + at - bnd_off[par[~in_own]]
        # push lists of the down sweep: front position -> the children's boundary entries that are this vertex
        gpos = front_off[par] + ppos                                     # global front position of every boundary entry
        order = np.argsort(gpos, kind="stable")
        n_front = int((s + b).sum())
        push_ptr = np.concatenate([[0], np.cumsum(np.bincount(gpos, minlength=n_front))]).astype(np.int64)
        push_tgt = order.astype(np.int64)                                # index into the concatenated boundary vectors
        plan = NDPlan()
        plan.V, plan.levels, plan.arity, plan.n_nodes = int(V), int(levels), int(arity), int(n_nodes)
        plan.D = plan.levels - 1
        plan.level_off, plan.level_of, plan.parent, plan.child_ix = level_off, level_of, parent, child_ix
        plan.perm, plan.inv = perm, inv
        plan.s, plan.b, plan.own_start = s, b, own_start
        plan.bnd, plan.bnd_off, plan.front_off = bnd.astype(np.int64), bnd_off, front_off
        plan.ppos = ppos.astype(np.int64)
        plan.push_ptr, plan.push_tgt = push_ptr, push_tgt
        plan.finv_off = np.concatenate([[0], np.cumsum(s * s)])[:-1]
        plan.w_off = np.concatenate([[0], np.cumsum(s * b)])[:-1]
        plan.finv_size, plan.w_size = int((s * s).sum()), int((s * b).sum())
        plan.node_of_new = nn
        return plan

    def level_nodes(self, lv):
        return np.arange(self.level_off[lv], self.level_off[lv + 1])

    def children(self, i):
        lv = int(self.level_of[i])
        if lv + 1 >= self.levels:
            return []
        first = int(self.level_off[lv + 1] + (i - self.level_off[lv]) * self.arity)
        return list(range(first, first + self.arity))

    @property
    def factor_entries(self):
        """fp32 numbers one solve reads: every W twice (up and down sweep), every Finv once"""
        return 2 * self.w_size + self.finv_size
