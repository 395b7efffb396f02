"""
oracle.laplacian -- numpy restatement of largesteps/geometry.py (TEST INFRASTRUCTURE, see oracle/__init__.py).

All functions return coalesced COO triplets in torch's coalesce order (row-major sorted, unique):
    rows int64 (nnz,), cols int64 (nnz,), vals (nnz,) float32 (or float64 with dtype=np.float64)
"""
import numpy as np

f32 = np.float32


def _coalesce(rows, cols, vals, V, dtype):
    """Sum duplicates, sort row-major: what torch's .coalesce() returns (geometry.py:94,133).
    Duplicates are accumulated in input order in `dtype` (np.add.at is sequential)."""
    key = rows.astype(np.int64) * V + cols.astype(np.int64)
    uk, inv = np.unique(key, return_inverse=True)
    out = np.zeros(uk.shape[0], dtype=dtype)
    np.add.at(out, inv.reshape(-1), vals.astype(dtype))
    return (uk // V).astype(np.int64), (uk % V).astype(np.int64), out


def unique_directed_edges(faces):
    """geometry.py:80-82: ii = f[:,[1,2,0]], jj = f[:,[2,0,1]], adj = unique columns of
    [[ii;jj],[jj;ii]] -> every undirected edge once per direction, lexicographically sorted."""
    f = np.asarray(faces).astype(np.int64)
    ii = f[:, [1, 2, 0]].reshape(-1)
    jj = f[:, [2, 0, 1]].reshape(-1)
    a = np.concatenate([ii, jj])
    b = np.concatenate([jj, ii])
    adj = np.unique(np.stack([a, b], axis=0), axis=1)
    return adj[0], adj[1]


def uniform_laplacian(V, faces, dtype=f32):
    """geometry.py:65-94. L = D - A on unique undirected edges; a degenerate face index pair
    (i,i) is kept exactly as the reference keeps it (it contributes -1 + 1 + ... on the diagonal)."""
    r, c = unique_directed_edges(faces)
    rows = np.concatenate([r, r])
    cols = np.concatenate([c, r])
    vals = np.concatenate([-np.ones(r.shape[0]), np.ones(r.shape[0])])
    return _coalesce(rows, cols, vals, V, dtype)


def face_cotangents(verts, faces, dtype=f32):
    """geometry.py:20-41 in the reference's own operation order. Returns (F,3) [cota, cotb, cotc] / 4.
    With dtype=float32 every intermediate is rounded to fp32 like the torch ops do."""
    v = np.asarray(verts).astype(dtype)
    f = np.asarray(faces).astype(np.int64)
    v0, v1, v2 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]

    def fma(a, b, c):  # fp32 fused multiply-add emulated through fp64 (product exact, one extra rounding <2^-29 odds)
        if dtype is not f32:
            return a * b + c
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)

    def norm(d):
        # torch .norm(dim=1) of 3 components on CPU evaluates sqrt(fma(z,z,fma(y,y,x*x))) (verified
        # bit-exact against torch 2.10 CPU, tests/golden); the HIP assembler uses the same chain.
        d = d.astype(dtype)
        x, y, z = d[:, 0], d[:, 1], d[:, 2]
        return np.sqrt(fma(z, z, fma(y, y, (x * x).astype(dtype)))).astype(dtype)

    A = norm(v1 - v2)
    B = norm(v0 - v2)
    C = norm(v0 - v1)
    half = dtype(0.5)
    s = (half * ((A + B).astype(dtype) + C).astype(dtype)).astype(dtype)
    prod = (((s * (s - A)).astype(dtype) * (s - B)).astype(dtype) * (s - C)).astype(dtype)
    area = np.sqrt(np.maximum(prod, dtype(1e-12))).astype(dtype)
    A2, B2, C2 = (A * A).astype(dtype), (B * B).astype(dtype), (C * C).astype(dtype)
    cota = (((B2 + C2).astype(dtype) - A2).astype(dtype) / area).astype(dtype)
    cotb = (((A2 + C2).astype(dtype) - B2).astype(dtype) / area).astype(dtype)
    cotc = (((A2 + B2).astype(dtype) - C2).astype(dtype) / area).astype(dtype)
    cot = np.stack([cota, cotb, cotc], axis=1)
    return (cot / dtype(4.0)).astype(dtype)


def _cot_triplets(verts, faces, dtype):
    """Uncoalesced W + W^T of geometry.py:43-56: cota -> (f1,f2), cotb -> (f2,f0), cotc -> (f0,f1)."""
    f = np.asarray(faces).astype(np.int64)
    cot = face_cotangents(verts, f, dtype).reshape(-1)          # (F*3,) in [cota,cotb,cotc] order per face
    ii = f[:, [1, 2, 0]].reshape(-1)
    jj = f[:, [2, 0, 1]].reshape(-1)
    rows = np.concatenate([ii, jj])
    cols = np.concatenate([jj, ii])
    w = np.concatenate([cot, cot])
    return rows, cols, w


def cot_laplacian(verts, faces, dtype=f32):
    """geometry.py:3-63, returned coalesced. Diagonal = column sums of W (geometry.py:59),
    accumulated in `dtype` in triplet order (the reference's order is implementation defined)."""
    V = np.asarray(verts).shape[0]
    rows, cols, w = _cot_triplets(verts, faces, dtype)
    diag = np.zeros(V, dtype=dtype)
    np.add.at(diag, cols, w)
    idx = np.arange(V, dtype=np.int64)
    return _coalesce(np.concatenate([idx, rows]), np.concatenate([idx, cols]),
                     np.concatenate([diag, -w]), V, dtype)


def matrix_coefficients(lambda_, alpha=None):
    """(a, b) of M = a I + b L, geometry.py:127-132, including the reference's ValueError text.
    The scalars are rounded to fp32 where torch rounds them (python double -> fp32 scalar)."""
    if alpha is None:
        return 1.0, float(lambda_)
    if alpha < 0.0 or alpha >= 1.0:
        raise ValueError(f"Invalid value for alpha: {alpha} : it should take values between 0 (included) and 1 (excluded)")
    return 1.0 - alpha, float(alpha)


def compute_matrix(verts, faces, lambda_, alpha=None, cotan=False, dtype=f32):
    """geometry.py:96-133. fp32 operation order:
       off-diagonal = sum over contributing triplets of fl(b * (-w));  diagonal = fl(a*1 + fl(b * L_ii))
       where for the uniform Laplacian L_ii is an exact small integer."""
    verts = np.asarray(verts)
    V = verts.shape[0]
    a, b = matrix_coefficients(lambda_, alpha)
    a, b = dtype(a), dtype(b)
    idx = np.arange(V, dtype=np.int64)
    if cotan:
        rows, cols, w = _cot_triplets(verts, faces, dtype)
        diag = np.zeros(V, dtype=dtype)
        np.add.at(diag, cols, w)                                 # geometry.py:59
        # uncoalesced L = [diag entries | -W entries]; b*L scales each entry, eye added, coalesce sums
        lr = np.concatenate([idx, rows, idx])
        lc = np.concatenate([idx, cols, idx])
        lv = np.concatenate([(b * diag).astype(dtype), (b * (-w)).astype(dtype), np.full(V, a, dtype=dtype)])
        # order of the sum on the diagonal: a + b*diag (two terms, commutative)
        return _coalesce(lr, lc, lv, V, dtype)
    r, c, lv = uniform_laplacian(V, faces, dtype)
    mv = (b * lv).astype(dtype)
    rows = np.concatenate([idx, r])
    cols = np.concatenate([idx, c])
    vals = np.concatenate([np.full(V, a, dtype=dtype), mv])
    return _coalesce(rows, cols, vals, V, dtype)


def to_csr(rows, cols, vals, V):
    """Row-major COO -> (rowptr int64 (V+1), cols, vals)."""
    rowptr = np.zeros(V + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr), cols, vals
