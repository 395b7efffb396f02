"""
Patch plan of the LDS-resident, s-step Chebyshev kernel (csrc/pcg.hip, k_patch_cheb) -- host-side analysis, numpy only.

Idea (temporal blocking): a Chebyshev step x_{k+1} = x_k + c1 (x_k - x_{k-1}) + c2 D^-1 (b - M x_k) only couples
mesh neighbours. Cut the mesh into compact patches of a few thousand vertices (recursive coordinate bisection), give every
patch its ghost layers 1..s (vertices at graph distance <= s) and let ONE workgroup keep the patch's two iterates in
LDS for s consecutive steps: the ghost layers go stale one layer per step, the patch's own vertices never do. HBM
then sees the matrix and the vectors once per s iterations instead of once per iteration; the gathers that dominate a
sparse matrix-vector product become LDS reads.

The plan renumbers the vertices patch-major (`perm`: new id -> old id), so a patch's own vertices are one contiguous
range of every solver-internal vector; only ghosts are addressed through an id list.

Per patch p (all arrays concatenated, offsets in `table`):
    local ids      [0, n_own) own | [n_own, n_rows) ghost layers 1..s-1 (recomputed) | [n_rows, n_local) layer s (read only)
    ghost_gid      new global id of every local vertex >= n_own
    cols16         (W, n_rows) uint16, ELL by columns: local id of the t-th off-diagonal neighbour of row r
                   (padding -> n_local, a zero slot of the LDS buffers)
    diag           (n_rows,) fp32 diagonal entries

Shrinking steps: after step j of a launch the values of layers > s-j are stale and never read again by anything that
reaches an own vertex, so step j only recomputes rows of layers <= s-j (`lim` in the table): ~half of the ghost rows.
"""
import numpy as np

MAX_DEPTH = 12
TABLE_COLS = 8 + MAX_DEPTH  # own_start, n_own, n_rows, n_local, W, off_gid, off_cols, off_diag | lim[0..MAX_DEPTH-1]
#                  lim[m] = rows of layers <= m (lim[0] = n_own, lim[depth-1] = n_rows): with S steps left in a launch only
#                  the layers <= S-1 still influence the own vertices, so step j (0-based) of S computes rows < lim[S-1-j]


def _spread3(x):
    """insert two zero bits after each of the 21 low bits of a uint64 array"""
    x = x & np.uint64(0x1FFFFF)
    x = (x | (x << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
    x = (x | (x << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
    x = (x | (x << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
    x = (x | (x << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
    x = (x | (x << np.uint64(2))) & np.uint64(0x1249249249249249)
    return x


def _morton(q):
    return _spread3(q[:, 0]) | (_spread3(q[:, 1]) << np.uint64(1)) | (_spread3(q[:, 2]) << np.uint64(2))


def cell_patches(positions, patch_size, min_patches=256, min_patch_size=192):
    """Cut the vertices into 2^m spatially compact, equally sized patches by recursive coordinate bisection (median
    split along the longest axis of every box), m the smallest depth with ceil(V / 2^m) <= patch_size. Equal sizes keep
    the workgroups of the patch kernel balanced; 2^m >= 256 patches fill the 256 CUs evenly.
    Returns (perm new->old: patch-major, scan-line order inside a patch; starts: first new id of every patch, + [V])."""
    p = np.asarray(positions, dtype=np.float64)
    V = p.shape[0]
    levels = 0
    while -(-V // (1 << levels)) > patch_size:
        levels += 1
    # at least one patch per CU (256) as long as the patches do not become tiny
    while (1 << levels) < min_patches and V // (1 << (levels + 1)) >= min_patch_size:
        levels += 1
    boxes = [np.arange(V, dtype=np.int64)]
    for _ in range(levels):
        nxt = []
        for idx in boxes:
            if idx.shape[0] <= 1:
                nxt += [idx, idx[:0]]
                continue
            q = p[idx]
            axis = int(np.argmax(q.max(axis=0) - q.min(axis=0)))
            half = idx.shape[0] // 2
            part = np.argpartition(q[:, axis], half)
            nxt += [idx[part[:half]], idx[part[half:]]]
        boxes = nxt
    lo = p.min(axis=0)
    ext = float(np.maximum(p.max(axis=0) - lo, 1e-30).max())
    fine = _morton(np.clip(((p - lo) / ext * ((1 << 21) - 1)).astype(np.uint64), 0, (1 << 21) - 1))
    boxes = [b for b in boxes if b.shape[0]]
    # patches along a Morton curve of their centres (neighbouring patches run on neighbouring workgroups / XCD L2s)
    centre = np.array([p[b].mean(axis=0) for b in boxes])
    ckey = _morton(np.clip(((centre - lo) / ext * ((1 << 21) - 1)).astype(np.uint64), 0, (1 << 21) - 1))
    order, starts = [], [0]
    for bi in np.argsort(ckey, kind="stable"):
        b = boxes[bi]
        # scan-line order inside the patch (rows along its longest axis, stacked along the second longest): the lanes
        # of a wavefront then process consecutive vertices of one mesh row and their neighbours are consecutive local
        # ids too -- stride-1 LDS gathers instead of the bank conflicts a space-filling curve produces
        q = p[b]
        e = q.max(axis=0) - q.min(axis=0)
        a1, a2 = np.argsort(-e)[:2]
        delta = max(float(np.sqrt(max(e[a1] * e[a2], 1e-300) / max(b.shape[0], 1))), 1e-30)
        row = np.floor((q[:, a2] - q[:, a2].min()) / delta + 0.5).astype(np.int64)
        order.append(b[np.lexsort((fine[b], q[:, a1], row))])
        starts.append(starts[-1] + b.shape[0])
    return np.concatenate(order).astype(np.int64), np.asarray(starts, dtype=np.int64)


def _entries(rowptr, rows):
    starts = rowptr[rows]
    lens = rowptr[rows + 1] - starts
    total = int(lens.sum())
    if total == 0:
        return np.empty(0, np.int64), lens
    first = np.cumsum(lens) - lens
    return np.arange(total, dtype=np.int64) - np.repeat(first, lens) + np.repeat(starts, lens), lens


class PatchPlan:
    def __init__(self, V, perm, depth, patch_size, table, ghost_gid, cols16, diag, max_local, max_rows, max_width):
        self.V, self.perm, self.depth, self.patch_size = int(V), perm, int(depth), int(patch_size)
        self.table, self.ghost_gid, self.cols16, self.diag = table, ghost_gid, cols16, diag
        self.n_patches = int(table.shape[0])
        self.max_local, self.max_rows, self.max_width = int(max_local), int(max_rows), int(max_width)

    @property
    def redundancy(self):
        """computed rows / owned rows"""
        return float(self.table[:, 2].sum()) / max(1, int(self.table[:, 1].sum()))

    @staticmethod
    def build(rowptr, col, diag, positions, patch_size=4096, depth=8, cap_local=6500, min_depth=2, cap_rows=8192):
        """rowptr/col: CSR pattern of M (diagonal included, old numbering); diag: (V,) diagonal of M; patch_size: upper
        bound of a patch's own vertices. Tries depth, depth-1, ... until every patch's local vertex count fits
        `cap_local` (LDS) and its computed rows fit `cap_rows` (8 rows per thread of a 1024-thread workgroup); returns None
        if even `min_depth` does not fit (the caller keeps the one-step kernel)."""
        rowptr = np.asarray(rowptr).astype(np.int64)
        col = np.asarray(col).astype(np.int64)
        V = rowptr.shape[0] - 1
        if V == 0:
            return None
        if not 1 <= depth <= MAX_DEPTH:
            raise ValueError(f"patch depth must be in [1, {MAX_DEPTH}]")
        perm, starts = cell_patches(positions, patch_size)
        inv = np.empty(V, dtype=np.int64)
        inv[perm] = np.arange(V)
        pos, lens = _entries(rowptr, perm)                   # matrix in the new numbering
        nrp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        ncol = inv[col[pos]]
        ndiag = np.asarray(diag, dtype=np.float32)[perm]
        for d in range(depth, min_depth - 1, -1):
            plan = PatchPlan._try(V, perm, nrp, ncol, ndiag, starts, patch_size, d, cap_local, cap_rows)
            if plan is not None:
                return plan
        return None

    @staticmethod
    def _try(V, perm, rowptr, col, diag, starts, patch_size, depth, cap_local, cap_rows=8192):
        seen = np.zeros(V, dtype=bool)
        lut = np.full(V + 1, -1, dtype=np.int64)
        tables, gids, colss, diags = [], [], [], []
        off_gid = off_cols = off_diag = 0
        max_local = max_rows = max_width = 0
        for s0, s1 in zip(starts[:-1], starts[1:]):
            own = np.arange(s0, s1, dtype=np.int64)
            seen[own] = True
            layers, frontier, n_local = [], own, own.shape[0]
            for _ in range(depth):
                p, _l = _entries(rowptr, frontier)
                nb = np.unique(col[p])
                nb = nb[~seen[nb]]
                seen[nb] = True
                layers.append(nb)
                frontier = nb
                n_local += nb.shape[0]
                if n_local > cap_local:
                    break
            touched = np.concatenate([own] + layers)
            seen[touched] = False
            if n_local > cap_local:
                return None
            local = touched                                    # [own | L1 | ... | Ls]
            n_own = own.shape[0]
            n_rows = n_own + sum(l.shape[0] for l in layers[:-1])
            if n_rows > cap_rows:
                return None
            lut[local] = np.arange(local.shape[0])
            rows = local[:n_rows]
            p, lens = _entries(rowptr, rows)
            c = col[p]
            r_of = np.repeat(np.arange(n_rows), lens)
            keep = c != rows[r_of]                              # drop the diagonal
            c, r_of = c[keep], r_of[keep]
            lc = lut[c]
            assert (lc >= 0).all()
            deg = np.bincount(r_of, minlength=n_rows)
            W = int(deg.max()) if n_rows else 0
            ell = np.full((W, n_rows), local.shape[0], dtype=np.uint16)      # padding -> zero slot
            slot = np.arange(c.shape[0]) - np.repeat(np.cumsum(deg) - deg, deg)
            ell[slot, r_of] = lc
            lut[local] = -1
            lims = np.cumsum([n_own] + [l.shape[0] for l in layers[:-1]]).tolist()
            lims += [n_rows] * (MAX_DEPTH - len(lims))
            tables.append([s0, n_own, n_rows, local.shape[0], W, off_gid, off_cols, off_diag] + lims)
            gids.append(local[n_own:].astype(np.int32))
            colss.append(ell.reshape(-1))
            diags.append(diag[rows])
            off_gid += local.shape[0] - n_own
            off_cols += W * n_rows
            off_diag += n_rows
            max_local, max_rows, max_width = max(max_local, local.shape[0]), max(max_rows, n_rows), max(max_width, W)
        if off_cols >= 2 ** 31 or off_gid >= 2 ** 31:
            return None
        cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.empty(0, dt)  # noqa: E731
        return PatchPlan(V, perm, depth, patch_size, np.asarray(tables, dtype=np.int32).reshape(-1, TABLE_COLS),
                         cat(gids, np.int32), cat(colss, np.uint16), cat(diags, np.float32), max_local, max_rows, max_width)
