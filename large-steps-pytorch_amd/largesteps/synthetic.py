"""
Deterministic synthetic meshes for the parity tests and the benchmark.

The reference ships no mesh assets, so every BASELINE.json
config runs on a generated stand-in:

* ``plane(n)``           n x n grid, 2 triangles per cell          (cfg 4: n=1000, cfg 5: n=2000)
* ``icosphere(n)``       geodesic sphere of frequency n, V=10n^2+2 (cfg 1: 16, cfg 2: 84, cfg 3: 158)
* ``perturb(...)``       seeded radial / tangential noise ("bunny"/"dragon" stand-ins)

numpy only: this module has no device code and is shared by bench.py, the tests
and the oracle.
"""
import numpy as np

__all__ = ["plane", "icosphere", "perturb", "config_mesh", "CONFIGS", "folded_sheet", "scroll", "shells"]


def plane(n, dtype=np.float32, index_dtype=np.int64):
    """n x n vertex grid. Vertex id = y*n + x, v = (x/(n-1), y/(n-1), 0.1 sin(2 pi x/(n-1))).
    Cell i = y*n + x (x,y < n-1) gives faces (i, i+1, i+n+1) and (i, i+n+1, i+n)."""
    if n < 2:
        raise ValueError("plane needs n >= 2")
    xs = np.arange(n, dtype=np.float64) / (n - 1)
    X, Y = np.meshgrid(xs, xs, indexing="xy")          # X varies along axis 1 (x), Y along axis 0 (y)
    v = np.stack([X, Y, 0.1 * np.sin(2.0 * np.pi * X)], axis=-1).reshape(-1, 3).astype(dtype)
    x = np.arange(n - 1)
    y = np.arange(n - 1)
    i = (y[:, None] * n + x[None, :]).reshape(-1)
    f = np.empty((i.size, 2, 3), dtype=index_dtype)
    f[:, 0, 0] = i
    f[:, 0, 1] = i + 1
    f[:, 0, 2] = i + n + 1
    f[:, 1, 0] = i
    f[:, 1, 1] = i + n + 1
    f[:, 1, 2] = i + n
    return v, f.reshape(-1, 3)


_T = (1.0 + 5.0 ** 0.5) / 2.0
_ICO_V = np.array(
    [[-1, _T, 0], [1, _T, 0], [-1, -_T, 0], [1, -_T, 0],
     [0, -1, _T], [0, 1, _T], [0, -1, -_T], [0, 1, -_T],
     [_T, 0, -1], [_T, 0, 1], [-_T, 0, -1], [-_T, 0, 1]], dtype=np.float64)
_ICO_F = np.array(
    [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
     [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
     [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
     [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)


def icosphere(n, dtype=np.float32, index_dtype=np.int64):
    """Geodesic icosphere of frequency n: each icosahedron face is split n x n, points are
    projected on the unit sphere and shared vertices merged. V = 10 n^2 + 2, F = 20 n^2.
    Vertices are ordered lexicographically by (x, y, z) (spatially banded)."""
    if n < 1:
        raise ValueError("icosphere needs n >= 1")
    base = _ICO_V / np.linalg.norm(_ICO_V[0])
    # barycentric lattice of one face: (i, j) with i + j <= n
    ii, jj = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
    keep = (ii + jj) <= n
    ii, jj = ii[keep], jj[keep]
    lid = -np.ones((n + 1, n + 1), dtype=np.int64)
    lid[ii, jj] = np.arange(ii.size)
    # local triangles
    a, b = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    up = (a + b) <= n - 1
    au, bu = a[up], b[up]
    tri_up = np.stack([lid[au, bu], lid[au + 1, bu], lid[au, bu + 1]], axis=1)
    dn = (a + b) <= n - 2
    ad, bd = a[dn], b[dn]
    tri_dn = np.stack([lid[ad + 1, bd], lid[ad + 1, bd + 1], lid[ad, bd + 1]], axis=1)
    tri = np.concatenate([tri_up, tri_dn], axis=0)
    w1 = ii[:, None] / n
    w2 = jj[:, None] / n
    w0 = 1.0 - w1 - w2
    pts, tris = [], []
    for k, (p, q, r) in enumerate(_ICO_F):
        P = w0 * base[p] + w1 * base[q] + w2 * base[r]
        pts.append(P)
        tris.append(tri + k * ii.size)
    P = np.concatenate(pts, axis=0)
    P /= np.linalg.norm(P, axis=1, keepdims=True)
    T = np.concatenate(tris, axis=0)
    # merge shared vertices (edges / corners of the 20 patches)
    key = np.round(P * 1e7).astype(np.int64)
    _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
    inv = inv.reshape(-1)
    v = P[first].astype(dtype)
    f = inv[T].astype(index_dtype)
    assert v.shape[0] == 10 * n * n + 2 and f.shape[0] == 20 * n * n
    return v, f


def perturb(v, radial=0.0, tangential=0.0, edge=None, seed=0):
    """Seeded noise: v *= 1 + radial*N(0,1) (per vertex), then a tangential jitter of
    `tangential` * edge (uniform in [-1,1]^3, projected off the radial direction)."""
    rng = np.random.default_rng(seed)
    v64 = v.astype(np.float64)
    if radial:
        v64 = v64 * (1.0 + radial * rng.standard_normal((v.shape[0], 1)))
    if tangential:
        if edge is None:
            raise ValueError("tangential jitter needs the reference edge length")
        d = rng.uniform(-1.0, 1.0, size=v.shape)
        nrm = v64 / np.maximum(np.linalg.norm(v64, axis=1, keepdims=True), 1e-30)
        d -= (d * nrm).sum(1, keepdims=True) * nrm
        v64 = v64 + tangential * edge * d
    return v64.astype(v.dtype)


# Surfaces that are folded in space: layers that are neighbours in space and far apart on the surface (cloth, ears, coils). The
# connectivity is the plane's / the sphere's; only the embedding differs -- what a fill-reducing ordering must not depend on.
def folded_sheet(n, gap=1e-3, dtype=np.float32):
    """plane(n) folded once along x = 1/2, the two layers `gap` apart."""
    v, f = plane(n, dtype=np.float64)
    x = v[:, 0]
    return np.stack([np.where(x < 0.5, x, 1.0 - x), v[:, 1], np.where(x < 0.5, 0.0, gap)], 1).astype(dtype), f


def scroll(n, turns, dtype=np.float32):
    """plane(n) rolled up around the y axis: x -> (r cos 2 pi T x, y, r sin 2 pi T x), r = 0.2 + 0.8 x."""
    v, f = plane(n, dtype=np.float64)
    x = v[:, 0]
    r = 0.2 + 0.8 * x
    return np.stack([r * np.cos(2 * np.pi * turns * x), v[:, 1], r * np.sin(2 * np.pi * turns * x)], 1).astype(dtype), f


def shells(n, gap=1e-3, dtype=np.float32):
    """two concentric icosphere(n) shells, radii 1 and 1 + gap (two components)."""
    v, f = icosphere(n, dtype=np.float64)
    return np.concatenate([v, v * (1.0 + gap)]).astype(dtype), np.concatenate([f, f + v.shape[0]])


# BASELINE.json configs -> stand-in meshes
CONFIGS = {
    "cfg1_icosphere2k": dict(kind="icosphere", n=16, lambda_=10.0, alpha=None, cotan=False),
    "cfg2_bunny70k": dict(kind="icosphere", n=84, radial=0.05, lambda_=19.0, alpha=None, cotan=False),
    "cfg3_dragon250k": dict(kind="icosphere", n=158, radial=0.05, tangential=0.25, lambda_=None, alpha=0.95, cotan=True),
    "cfg4_plane1m": dict(kind="plane", n=1000, lambda_=50.0, alpha=None, cotan=False),
    "cfg5_plane4m": dict(kind="plane", n=2000, lambda_=50.0, alpha=None, cotan=False),
    # not BASELINE.json configs: surfaces folded in space (bench.py --workload ..., tests): the plane's connectivity, another embedding
    "scroll250k": dict(kind="scroll", n=500, turns=3, lambda_=19.0, alpha=None, cotan=False),
    "scroll10_250k": dict(kind="scroll", n=500, turns=10, lambda_=19.0, alpha=None, cotan=False),
    "folded250k": dict(kind="folded", n=500, lambda_=19.0, alpha=None, cotan=False),
    "shells250k": dict(kind="shells", n=112, lambda_=19.0, alpha=None, cotan=False),
    # the headline size on surfaces that are not the plane (round 6): cfg3's recipe at 1M vertices (a closed, noisy scan: what the
    # reference's figures/*/generate_data.py feed it), its uniform-Laplacian twin, a rolled and a folded 1000 x 1000 sheet
    "cfg4b_sphere1m": dict(kind="icosphere", n=316, radial=0.05, tangential=0.25, lambda_=None, alpha=0.95, cotan=True),
    "cfg4b_sphere1m_uniform": dict(kind="icosphere", n=316, radial=0.05, tangential=0.25, lambda_=50.0, alpha=None, cotan=False),
    "scroll1m": dict(kind="scroll", n=1000, turns=3, lambda_=50.0, alpha=None, cotan=False),
    "folded1m": dict(kind="folded", n=1000, lambda_=50.0, alpha=None, cotan=False),
}


def config_mesh(name):
    """Return (verts fp32, faces int64, params dict) of a named BASELINE.json config."""
    c = dict(CONFIGS[name])
    if c["kind"] == "plane":
        v, f = plane(c["n"])
    elif c["kind"] == "scroll":
        v, f = scroll(c["n"], c["turns"])
    elif c["kind"] == "folded":
        v, f = folded_sheet(c["n"])
    elif c["kind"] == "shells":
        v, f = shells(c["n"])
    else:
        v, f = icosphere(c["n"])
        if c.get("radial") or c.get("tangential"):
            edge = 1.2 / c["n"]       # ~ mean edge length of a frequency-n unit geodesic sphere
            v = perturb(v, radial=c.get("radial", 0.0), tangential=c.get("tangential", 0.0), edge=edge, seed=0)
    return v, f, c
