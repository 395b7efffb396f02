"""
numpy statement of the vertex-block shard plan (what csrc/shard_plan.cpp computes natively behind ls_shard_plan_*): the CPU tests
check the native plan against this, array for array. TEST INFRASTRUCTURE (moved out of the package in round 3).
"""
import numpy as np


def block_bounds(V, P):
    """Contiguous vertex blocks: rank r owns [bounds[r], bounds[r+1])."""
    return np.array([(r * V) // P for r in range(P + 1)], dtype=np.int64)


class ShardPlan:
    """Everything rank `rank` of `P` needs to know about its block of a V x V CSR matrix (host side, numpy).

    depth = 1 (PCG): the shard computes its owned rows; columns are [owned | halo layer 1].
    depth = s > 1 (Chebyshev, one halo exchange per s iterations): the shard ALSO computes the ghost layers
    1..s-1 redundantly (layer j = vertices at graph distance j from the block) and reads layer s, so that s
    iterations can run between two exchanges -- after j iterations the layers > s-j are stale, the owned rows never are.

    Local ids: [owned (n_own) | ghost layers 1..s-1 (computed) | ghost layer s (read only)]; the first `n_rows`
    local ids are the rows of the local matrix. Ghosts are ordered by (owner rank, global id) inside each of the
    two ghost groups, so every (owner, group) pair is one contiguous range:
    recv : [(src_rank, offset_in_ghost_region, count)]   in local order
    send : [(dst_rank, local_row_ids int32)]             the matching owned rows, message for message
    """

    def __init__(self, rank, P, lo, hi, depth, rowptr, col, val, ghost_global, n_ghost_rows, recv, send):
        self.rank, self.P, self.lo, self.hi, self.depth = rank, P, int(lo), int(hi), int(depth)
        self.n_own = int(hi - lo)
        self.n_halo = int(ghost_global.shape[0])
        self.n_rows = self.n_own + int(n_ghost_rows)          # rows of the local matrix (owned + computed ghosts)
        self.n_cols = self.n_own + self.n_halo
        self.rowptr, self.col, self.val = rowptr, col, val
        self.halo_global, self.recv, self.send = ghost_global, recv, send

    @staticmethod
    def _entries(rowptr, rows):
        """Flat positions (into col / val) of all entries of the CSR rows `rows`, row after row, + the row lengths."""
        starts = rowptr[rows]
        lens = rowptr[rows + 1] - starts
        total = int(lens.sum())
        if total == 0:
            return np.empty(0, np.int64), lens
        first = np.cumsum(lens) - lens                       # position of each row's first entry in the output
        pos = np.arange(total, dtype=np.int64) - np.repeat(first, lens) + np.repeat(starts, lens)
        return pos, lens

    @staticmethod
    def _layers(rowptr, col, V, lo, hi, depth):
        """Ghost layers 1..depth of the block [lo, hi): breadth-first search on the matrix pattern."""
        seen = np.zeros(V, dtype=bool)
        seen[lo:hi] = True
        frontier = np.arange(lo, hi, dtype=np.int64)
        layers = []
        for _ in range(depth):
            pos, _ = ShardPlan._entries(rowptr, frontier)
            nb = np.unique(col[pos])
            nb = nb[~seen[nb]]
            seen[nb] = True
            layers.append(nb.astype(np.int64))
            frontier = nb
        return layers

    @staticmethod
    def _ghost_groups(rowptr, col, V, bounds, q, depth):
        """(computed ghosts, read-only ghosts) of rank q, each sorted by (owner, global id)."""
        layers = ShardPlan._layers(rowptr, col, V, bounds[q], bounds[q + 1], depth)
        inner = np.sort(np.concatenate(layers[:-1])) if depth > 1 else np.empty(0, np.int64)
        outer = np.sort(layers[-1])
        return inner, outer      # contiguous blocks => sorting by id sorts by owner first

    @staticmethod
    def build(rowptr, col, val, V, P, rank, depth=1):
        rowptr = np.asarray(rowptr).astype(np.int64)
        col = np.asarray(col).astype(np.int64)
        val = np.asarray(val, dtype=np.float32)
        if P < 1 or not (0 <= rank < P):
            raise ValueError(f"invalid rank {rank} of {P}")
        if P > max(V, 1):
            raise ValueError(f"cannot cut {V} vertices into {P} non-empty blocks")
        if depth < 1:
            raise ValueError("halo depth must be >= 1")
        bounds = block_bounds(V, P)
        lo, hi = bounds[rank], bounds[rank + 1]
        inner, outer = ShardPlan._ghost_groups(rowptr, col, V, bounds, rank, depth)
        ghosts = np.concatenate([inner, outer])
        glob = np.concatenate([np.arange(lo, hi), ghosts])
        lut = np.full(V, -1, dtype=np.int64)
        lut[glob] = np.arange(glob.shape[0])
        rows_global = glob[: (hi - lo) + inner.shape[0]]
        pos, lens = ShardPlan._entries(rowptr, rows_global)    # rows in local order, columns still global
        local_rowptr = np.concatenate([[0], np.cumsum(lens)])
        local_col = lut[col[pos]]
        assert (local_col >= 0).all(), "a computed row references a column outside the halo"
        # per-owner contiguous ranges of the two ghost groups
        recv = []
        for group, base in ((inner, 0), (outer, inner.shape[0])):
            owner = np.searchsorted(bounds, group, side="right") - 1
            for q in np.unique(owner):
                idx = np.nonzero(owner == q)[0]
                assert idx[-1] - idx[0] + 1 == idx.shape[0]
                recv.append((int(q), int(base + idx[0]), int(idx.shape[0])))
        # what the others need from me, in THEIR order (group by group, ids ascending)
        send = []
        groups_of = {q: ShardPlan._ghost_groups(rowptr, col, V, bounds, q, depth) for q in range(P) if q != rank}
        for gi in (0, 1):
            for q in range(P):
                if q == rank:
                    continue
                g = groups_of[q][gi]
                mine = g[(g >= lo) & (g < hi)]
                if mine.shape[0]:
                    send.append((q, (mine - lo).astype(np.int32)))
        # the receiver walks its recv list group by group and, inside a group, owner by owner: same order here
        return ShardPlan(rank, P, lo, hi, depth, local_rowptr.astype(np.int32), local_col.astype(np.int32),
                         val[pos].astype(np.float32), ghosts, inner.shape[0], recv, send)


