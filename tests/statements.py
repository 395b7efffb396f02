"""
Numpy statements of device kernels (TEST code): what csrc/direct.hip + largesteps/direct.py (nested-dissection
factorisation and the two sweeps) and csrc/pcg.hip's k_patch_cheb (one launch of the LDS-resident s-step Chebyshev
kernel) compute, written against the same host-side plans (largesteps.nested.NDPlan, largesteps.patches.PatchPlan;
`self` below is such a plan). They let the CPU tests prove the plans and the algorithms without a GPU; the product
never imports this module.
"""
import numpy as np

from nd_plan_statement import _row_index


# ---- nested-dissection direct solver: numeric factorisation and the two sweeps (dense per node) ----------------
def nd_factor(self, rowptr, col, val):
    """Returns (finv, w): flat fp64 arrays; node i: Finv = finv[finv_off:+s*s].reshape(s,s), W = w[w_off:+b*s].reshape(b,s)."""
    rows = _row_index(np.asarray(rowptr).astype(np.int64))
    prow, pcol = self.inv[rows], self.inv[np.asarray(col).astype(np.int64)]
    val = np.asarray(val, dtype=np.float64)
    order = np.argsort(self.node_of_new[prow], kind="stable")
    prow, pcol, val = prow[order], pcol[order], val[order]
    ent_off = np.concatenate([[0], np.cumsum(np.bincount(self.node_of_new[prow], minlength=self.n_nodes + 1))])
    finv, w = np.zeros(self.finv_size), np.zeros(self.w_size)
    U = {}
    for lv in range(self.levels - 1, -1, -1):
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
            for ch in self.children(i):
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

def nd_solve(self, finv, w, rhs):
    """rhs in the ORIGINAL numbering, (V, k); returns x in the original numbering. Same data flow as csrc/direct.hip:
    `slots` (one per front position and child, zero unless pushed) upwards, push lists downwards."""
    bp = np.asarray(rhs, dtype=np.float64)[self.perm]
    k = bp.shape[1]
    n_front = int((self.s + self.b).sum())
    slots = np.zeros((n_front, self.arity, k))
    bprime = bp.copy()
    for lv in range(self.levels - 1, -1, -1):
        for i in self.level_nodes(lv):
            s, b, o, f = int(self.s[i]), int(self.b[i]), int(self.own_start[i]), int(self.front_off[i])
            bprime[o:o + s] = bp[o:o + s] - slots[f:f + s].sum(axis=1)
            W = w[self.w_off[i]:self.w_off[i] + b * s].reshape(b, s)
            upd = W @ bprime[o:o + s] + slots[f + s:f + s + b].sum(axis=1)
            if b:
                pf = int(self.front_off[self.parent[i]])
                slots[pf + self.ppos[self.bnd_off[i]:self.bnd_off[i] + b], int(self.child_ix[i])] = upd
    x = np.zeros_like(bp)
    xb = np.zeros((self.bnd.shape[0], k))
    for lv in range(self.levels):
        for i in self.level_nodes(lv):
            s, b, o, f = int(self.s[i]), int(self.b[i]), int(self.own_start[i]), int(self.front_off[i])
            Fi = finv[self.finv_off[i]:self.finv_off[i] + s * s].reshape(s, s)
            W = w[self.w_off[i]:self.w_off[i] + b * s].reshape(b, s)
            xbi = xb[self.bnd_off[i]:self.bnd_off[i] + b]
            x[o:o + s] = Fi @ bprime[o:o + s] - W.T @ xbi
            front_x = np.concatenate([x[o:o + s], xbi])
            for p_ in range(s + b):
                t0, t1 = self.push_ptr[f + p_], self.push_ptr[f + p_ + 1]
                xb[self.push_tgt[t0:t1]] = front_x[p_]
    assert np.array_equal(x[self.bnd], xb), "every boundary entry received its vertex's x"
    out = np.empty_like(x)
    out[self.perm] = x
    return out


# ---- k_patch_cheb ------------------------------------------------------------------------------------------------
def patch_steps(self, offdiag, b_new, cur, prev, c1, c2):
    """One launch: len(c1) <= depth Chebyshev steps on every patch from the global iterates (cur, prev) in the
    NEW numbering; returns the two newest iterates (newest, second newest) on the owned rows."""
    out_cur, out_prev = cur.copy(), prev.copy()
    S = len(c1)
    for row in self.table:
        own_start, n_own, n_rows, n_local, W, og, oc, od = (int(t) for t in row[:8])
        lim = row[8:]
        gid = np.concatenate([np.arange(own_start, own_start + n_own), self.ghost_gid[og:og + n_local - n_own]])
        A = np.vstack([cur[gid], np.zeros((1, cur.shape[1]), cur.dtype)])
        B = np.vstack([prev[gid], np.zeros((1, cur.shape[1]), cur.dtype)])
        ell = self.cols16[oc:oc + W * n_rows].reshape(W, n_rows).astype(np.int64)
        d = self.diag[od:od + n_rows][:, None]
        bl = b_new[gid[:n_rows]]
        for j, (a1, a2) in enumerate(zip(c1, c2)):
            m = int(lim[S - 1 - j])                        # rows still needed by the own vertices
            s = A[ell[:, :m]].sum(axis=0)                  # (m, k) sum of the neighbours
            ax = d[:m] * A[:m] + offdiag * s
            B[:m] = A[:m] + a1 * (A[:m] - B[:m]) + a2 * (bl[:m] - ax) / d[:m]
            A, B = B, A
        out_cur[own_start:own_start + n_own] = A[:n_own]
        out_prev[own_start:own_start + n_own] = B[:n_own]
    return out_cur, out_prev
