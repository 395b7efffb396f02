"""
GPU parity tests (-m gpu): the HIP path, called through the C ABI (largesteps._native -> liblargesteps_hip.so),
against the CPU oracle and the reference-generated golden fixtures. Tolerances are stated at each assert.
"""
import gc
import os

import numpy as np
import pytest
import torch

from oracle import laplacian as ol
from oracle import solve as osv

pytestmark = pytest.mark.gpu

ALL_MESHES = ["octahedron", "tetra", "quad", "collinear", "unreferenced", "nonmanifold", "dupface", "ico3", "plane12", "ico6"]
CASES = ["uni_l10", "uni_l0p3", "uni_a0p95", "cot_l2", "cot_a0p9"]
MANIFOLD = ("ico3", "plane12", "ico6", "octahedron", "tetra", "quad", "unreferenced")


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from largesteps import _native
    _native.lib()          # fail loudly if the extension is missing
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(params=["direct", "iterative"])
def chol_path(request, monkeypatch):
    """'Cholesky' has two implementations: the nested-dissection direct solver (default for compute_matrix matrices)
    and the Chebyshev / PCG iteration (LARGESTEPS_NO_DIRECT=1, and the automatic fallback). Tests run on both."""
    if request.param == "iterative":
        monkeypatch.setenv("LARGESTEPS_NO_DIRECT", "1")
    else:
        monkeypatch.delenv("LARGESTEPS_NO_DIRECT", raising=False)
    return request.param


# ---------------------------------------------------------------------------------------------------
# assembly
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("name", ALL_MESHES)
def test_compute_matrix_vs_reference_and_oracle(golden, dev, name, case):
    from largesteps.geometry import compute_matrix
    v, f = golden[f"{name}/verts"], golden[f"{name}/faces"]
    kw = golden.params[case]
    M = compute_matrix(_t(v, dev), _t(f, dev), **kw)
    assert M.is_coalesced() and M.dtype == torch.float32 and M.indices().dtype == torch.int64
    idx = M.indices().cpu().numpy()
    val = M.values().cpu().numpy()
    ref_idx, ref_val = golden[f"{name}/{case}/idx"], golden[f"{name}/{case}/val"]
    assert np.array_equal(idx, ref_idx), "index list must equal torch's coalesce order exactly"
    r, c, oval = ol.compute_matrix(v, f, **kw)
    if not kw["cotan"]:
        assert np.array_equal(val, ref_val), "uniform M is bit exact vs the reference"
    else:
        off = idx[0] != idx[1]
        scale = max(np.abs(ref_val).max(), abs(kw["alpha"] or kw["lambda_"]) * np.abs(ol.face_cotangents(v, f)).max())
        if name in MANIFOLD:
            # same fp32 operation order as the oracle -> bit exact off-diagonals (<= 2 terms per entry)
            assert np.array_equal(val[off], oval[off])
            # vs reference: torch-CPU's vectorised sqrt is 1 ulp off for ~0.6% of inputs -> 2 ulp
            np.testing.assert_allclose(val[off], ref_val[off], rtol=2.5e-7, atol=0)
        # diagonal / non-manifold sums: accumulation order is implementation defined in the reference
        np.testing.assert_allclose(val, oval, rtol=0, atol=4e-6 * scale)
        np.testing.assert_allclose(val, ref_val, rtol=0, atol=4e-6 * scale)
    # int32 faces give the identical matrix (the reference accepts int32 for the uniform Laplacian only)
    M32 = compute_matrix(_t(v, dev), _t(f.astype(np.int32), dev), **kw)
    assert torch.equal(M32.indices(), M.indices())
    assert np.array_equal(M32.values().cpu().numpy(), val)


@pytest.mark.parametrize("name", ALL_MESHES)
def test_laplacians(golden, dev, name):
    from largesteps.geometry import laplacian_uniform, laplacian_cot
    v, f = golden[f"{name}/verts"], golden[f"{name}/faces"]
    L = laplacian_uniform(_t(v, dev), _t(f, dev))
    assert np.array_equal(L.indices().cpu().numpy(), golden[f"{name}/Luni_idx"])
    assert np.array_equal(L.values().cpu().numpy(), golden[f"{name}/Luni_val"])
    Lc = laplacian_cot(_t(v, dev), _t(f, dev)).coalesce()
    ref = golden[f"{name}/Lcot_val"]
    assert np.array_equal(Lc.indices().cpu().numpy(), golden[f"{name}/Lcot_idx"])
    scale = max(np.abs(ref).max(), np.abs(ol.face_cotangents(v, f)).max())
    np.testing.assert_allclose(Lc.values().cpu().numpy(), ref, rtol=0, atol=4e-6 * scale)


def test_face_permutation_and_large_mesh_pattern(dev):
    """Assembly is independent of the face order (atomics arrival order) and exact at 250k vertices."""
    from largesteps.geometry import compute_matrix
    from largesteps import synthetic
    v, f, _ = synthetic.config_mesh("cfg3_dragon250k")
    perm = np.random.default_rng(0).permutation(f.shape[0])
    A = compute_matrix(_t(v, dev), _t(f, dev), 0.0, alpha=0.95, cotan=True)
    B = compute_matrix(_t(v, dev), _t(f[perm], dev), 0.0, alpha=0.95, cotan=True)
    assert torch.equal(A.indices(), B.indices())
    assert torch.equal(A.values(), B.values()), "row-local sort makes the fp32 sums order independent"
    r, c, val = ol.compute_matrix(v, f, 0.0, alpha=0.95, cotan=True)
    assert np.array_equal(A.indices().cpu().numpy(), np.stack([r, c]))
    got = A.values().cpu().numpy()
    off = r != c
    assert np.array_equal(got[off], val[off])
    np.testing.assert_allclose(got, val, rtol=0, atol=4e-6 * np.abs(val).max())
    U = compute_matrix(_t(v, dev), _t(f, dev), 19.0)
    r, c, val = ol.compute_matrix(v, f, 19.0)
    assert np.array_equal(U.indices().cpu().numpy(), np.stack([r, c]))
    assert np.array_equal(U.values().cpu().numpy(), val)


@pytest.mark.parametrize("seed", range(6))
def test_assembly_random_soup(dev, seed):
    """Arbitrary triangle soups (non-manifold fans, duplicated faces, repeated vertices in a face, unreferenced
    vertices, high valence): pattern exact, uniform values exact, cot values within the accumulation-order tolerance."""
    from largesteps.geometry import compute_matrix
    rng = np.random.default_rng(seed)
    V = int(rng.integers(5, 400))
    F = int(rng.integers(1, 4 * V))
    f = rng.integers(0, max(2, int(V * 0.8)), size=(F, 3)).astype(np.int64)     # the last 20 % stay unreferenced
    hub = int(rng.integers(0, V))
    f[: F // 4, 0] = hub                                                        # one very high valence vertex
    v = rng.standard_normal((V, 3)).astype(np.float32)
    lam = float(rng.uniform(0.1, 60.0))
    M = compute_matrix(_t(v, dev), _t(f, dev), lam)
    r, c, val = ol.compute_matrix(v, f, lam)
    assert np.array_equal(M.indices().cpu().numpy(), np.stack([r, c]))
    assert np.array_equal(M.values().cpu().numpy(), val)
    Mc = compute_matrix(_t(v, dev), _t(f.astype(np.int32), dev), lam, alpha=0.7, cotan=True)
    r, c, val = ol.compute_matrix(v, f, lam, alpha=0.7, cotan=True)
    assert np.array_equal(Mc.indices().cpu().numpy(), np.stack([r, c]))
    scale = max(np.abs(val).max(), 0.7 * np.abs(ol.face_cotangents(v, f)).max())
    np.testing.assert_allclose(Mc.values().cpu().numpy(), val, rtol=0, atol=2e-5 * scale)


@pytest.mark.parametrize("kind", ["scattered", "one_pair", "shuffled_plane"])
def test_assembly_when_the_corner_table_overflows(dev, kind):
    """k_count ranks the corners of a workgroup's 256 faces in a 512-entry LDS table sized for meshes; a corner that finds no entry
    within 8 probes reserves its slots with its own atomic (csrc/assemble.hip). Faces that share no vertices (768 distinct vertex pairs
    per workgroup), faces that all meet in ONE vertex pair, and a plane whose faces are shuffled so that no workgroup sees a vertex
    twice: pattern and uniform values exact, cotangent values within the accumulation-order tolerance."""
    from largesteps.geometry import compute_matrix
    rng = np.random.default_rng(7)
    if kind == "scattered":
        V, F = 200000, 20000
        f = rng.permutation(V)[: 3 * F].reshape(F, 3).astype(np.int64)           # every vertex in at most one face
        v = rng.standard_normal((V, 3)).astype(np.float32)
    elif kind == "one_pair":
        V, F = 5000, 3000
        f = np.stack([np.full(F, 10), np.full(F, 11), 12 + rng.integers(0, V - 12, F)], 1).astype(np.int64)   # valence 3000
        v = rng.standard_normal((V, 3)).astype(np.float32)
    else:
        from largesteps import synthetic
        v, f = synthetic.plane(150)[:2]
        f = f[rng.permutation(f.shape[0])].astype(np.int64)
    for idx in (np.int64, np.int32):
        M = compute_matrix(_t(v, dev), _t(f.astype(idx), dev), 7.5)
        r, c, val = ol.compute_matrix(v, f, 7.5)
        assert np.array_equal(M.indices().cpu().numpy(), np.stack([r, c]))
        assert np.array_equal(M.values().cpu().numpy(), val)
    Mc = compute_matrix(_t(v, dev), _t(f, dev), 7.5, alpha=0.6, cotan=True)
    r, c, val = ol.compute_matrix(v, f, 7.5, alpha=0.6, cotan=True)
    assert np.array_equal(Mc.indices().cpu().numpy(), np.stack([r, c]))
    scale = max(np.abs(val).max(), 0.6 * np.abs(ol.face_cotangents(v, f)).max())
    np.testing.assert_allclose(Mc.values().cpu().numpy(), val, rtol=0, atol=2e-5 * scale)


def test_assembly_errors(golden, dev):
    from largesteps.geometry import compute_matrix
    e = golden.errors()
    v, f = golden["quad/verts"], golden["quad/faces"]
    for a in (1.0, -0.1, 1.5):
        with pytest.raises(ValueError) as ei:
            compute_matrix(_t(v, dev), _t(f, dev), 1.0, alpha=a)
        assert str(ei.value) == e[f"alpha={a}"]
    bad = f.copy()
    bad[0, 0] = 99
    with pytest.raises(IndexError):
        compute_matrix(_t(v, dev), _t(bad, dev), 1.0)
    with pytest.raises(RuntimeError):
        compute_matrix(torch.from_numpy(v), torch.from_numpy(f), 1.0)      # CPU tensors: no CPU path
    M = compute_matrix(_t(v, dev), _t(np.zeros((0, 3), np.int64), dev), 2.0)   # no faces -> identity
    assert np.array_equal(M.values().cpu().numpy(), np.ones(4, np.float32))


# ---------------------------------------------------------------------------------------------------
# to_differential
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("name", ["octahedron", "tetra", "quad", "unreferenced", "ico3", "plane12", "ico6"])
def test_to_differential(golden, dev, name, case):
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import to_differential
    from largesteps import _native
    v, f = golden[f"{name}/verts"], golden[f"{name}/faces"]
    M = compute_matrix(_t(v, dev), _t(f, dev), **golden.params[case])
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    u64 = osv.to_differential(idx[0], idx[1], val, v)
    # fp32 SpMV of ~7 terms: |err| <= ~8 ulp of ||M||_inf ||v||_inf
    bound = 1e-6 * np.abs(val).max() * max(np.abs(v).max(), 1.0) * 8
    u = to_differential(M, _t(v, dev))
    assert np.abs(u.cpu().numpy() - u64).max() <= bound
    assert np.abs(u.cpu().numpy() - golden[f"{name}/{case}/u"]).max() <= 2 * bound
    csr = _native.csr_of(M)
    for variant in (0, 1):
        uv = _native.spmv(csr, _t(v, dev), variant)
        assert np.abs(uv.cpu().numpy() - u64).max() <= bound
    # 1-D and wide inputs
    u1 = to_differential(M, _t(v[:, 0].copy(), dev))
    assert u1.shape == (v.shape[0],) and np.abs(u1.cpu().numpy() - u64[:, 0]).max() <= bound
    wide = np.random.default_rng(0).standard_normal((v.shape[0], 7)).astype(np.float32)
    uw = to_differential(M, _t(wide, dev))
    assert np.abs(uw.cpu().numpy() - osv.to_differential(idx[0], idx[1], val, wide)).max() <= bound * 4 * max(np.abs(wide).max(), 1)


def test_to_differential_autograd(golden, dev):
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import to_differential
    v, f = golden["ico6/verts"], golden["ico6/faces"]
    M = compute_matrix(_t(v, dev), _t(f, dev), 5.0)
    tv = _t(v, dev).requires_grad_(True)
    w = torch.randn(v.shape, device=dev)
    (to_differential(M, tv) * w).sum().backward()
    ref = torch.sparse.mm(M.t(), w)
    assert (tv.grad - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


# ---------------------------------------------------------------------------------------------------
# from_differential
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["uni_l10", "cot_a0p9"])
@pytest.mark.parametrize("name", ["octahedron", "tetra", "ico3", "plane12", "ico6"])
def test_from_differential_vs_reference(golden, dev, name, case, chol_path):
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import from_differential
    v, f = golden[f"{name}/verts"], golden[f"{name}/faces"]
    M = compute_matrix(_t(v, dev), _t(f, dev), **golden.params[case])
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    u_np = golden[f"{name}/{case}/u"]
    x64 = osv.from_differential(idx[0], idx[1], val, u_np)
    scale = max(np.abs(x64).max(), 1.0)
    g64 = osv.from_differential(idx[0], idx[1], val, golden[f"{name}/{case}/cg_w"])
    for method in ("Cholesky", "CG"):
        u = _t(u_np, dev).requires_grad_(True)
        x = from_differential(M, u, method)
        if method == "Cholesky":
            from largesteps import parameterize
            assert parameterize._cache[(id(M), method)][0].method == ("nested-dissection" if chol_path == "direct" else "iterative")
        assert x.shape == u.shape and x.dtype == torch.float32 and x.data_ptr() != u.data_ptr()
        # stated fp32 tolerance of the path: 1e-4 relative (max-abs) vs the fp64 direct solve; on these
        # small well-conditioned systems the error is two orders of magnitude below that
        assert np.abs(x.detach().cpu().numpy() - x64).max() <= 2e-5 * scale
        assert np.abs(x.detach().cpu().numpy() - golden[f"{name}/{case}/cg_x"]).max() <= 4e-5 * scale
        (x * _t(golden[f"{name}/{case}/cg_w"], dev)).sum().backward()
        assert np.abs(u.grad.cpu().numpy() - g64).max() <= 2e-5 * max(np.abs(g64).max(), 1.0)
        assert np.abs(u.grad.cpu().numpy() - golden[f"{name}/{case}/cg_grad_u"]).max() <= 4e-5 * max(np.abs(g64).max(), 1.0)
    # warm-started second call of the reference-compatible CG (solvers.py:102-124)
    x2 = from_differential(M, _t(u_np * np.float32(1.01), dev), "CG")
    assert np.abs(x2.cpu().numpy() - golden[f"{name}/{case}/cg_x_warm"]).max() <= 4e-5 * scale


@pytest.mark.parametrize("block", [0, 256, 512, 1024])
@pytest.mark.parametrize("k", [1, 2, 3, 4, 6])
def test_solver_geometries_and_widths(dev, block, k):
    from largesteps.geometry import compute_matrix
    from largesteps.solvers import PCGSolver
    from largesteps import synthetic
    v, f = synthetic.icosphere(20)          # 4002 vertices: ragged last tile and last SELL slice, both geometries
    v = synthetic.perturb(v, radial=0.05, seed=2)
    M = compute_matrix(_t(v, dev), _t(f, dev), 0.0, alpha=0.9, cotan=True)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    b = np.random.default_rng(k).standard_normal((v.shape[0], k)).astype(np.float32)
    x64 = osv.from_differential(idx[0], idx[1], val, b)
    s = PCGSolver(M, rtol=1e-6)
    s.set_option("block", block)
    x = s.solve(_t(b, dev))
    assert s.last_info["converged"] and 5 < s.last_info["iterations"] < 500
    assert np.abs(x.cpu().numpy() - x64).max() <= 1e-4 * np.abs(x64).max()
    # warm start from the solution: converged at once, solution unchanged
    s.warm_start = True
    s.guess_fwd = x.clone()
    x_again = s.solve(_t(b, dev))
    assert s.last_info["iterations"] <= 10
    assert np.abs(x_again.cpu().numpy() - x64).max() <= 1e-4 * np.abs(x64).max()
    # zero right-hand side (cold start: a relative tolerance has no meaning for b = 0 from a nonzero guess)
    s.warm_start = False
    z = s.solve(torch.zeros_like(x))
    assert s.last_info["converged"] and s.last_info["iterations"] == 0
    assert float(z.abs().max()) == 0.0


@pytest.mark.parametrize("cot", [False, True])
@pytest.mark.parametrize("k", [1, 3, 4])
def test_chebyshev_solver(dev, cot, k):
    """Chebyshev-Jacobi path (the 'Cholesky' default for compute_matrix matrices) vs the fp64 direct solve."""
    from largesteps.geometry import compute_matrix
    from largesteps.solvers import PCGSolver
    from largesteps import synthetic
    v, f = synthetic.icosphere(20)
    v = synthetic.perturb(v, radial=0.05, tangential=0.15 if cot else 0.0, edge=0.06, seed=2)
    kw = dict(lambda_=0.0, alpha=0.9, cotan=True) if cot else dict(lambda_=25.0)
    M = compute_matrix(_t(v, dev), _t(f, dev), **kw)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    b = np.random.default_rng(k).standard_normal((v.shape[0], k)).astype(np.float32)
    x64 = osv.from_differential(idx[0], idx[1], val, b)
    s = PCGSolver(M, rtol=1e-6, chebyshev=True)
    assert s.chebyshev
    x = s.solve(_t(b, dev))
    assert s.last_info["method"] == "chebyshev" and s.last_info["converged"]
    # stated tolerance of the path: 1e-4 relative max-abs error vs the fp64 direct solve
    assert np.abs(x.cpu().numpy() - x64).max() <= 1e-4 * np.abs(x64).max()
    n_cold = s.last_info["iterations"]
    # the enclosure really contains the spectrum: lmin <= lambda(D^-1 M) <= lmax (dense check, 4002 vertices)
    import ctypes
    from largesteps import _native
    lo, hi = ctypes.c_double(), ctypes.c_double()
    _native.check(_native.lib().ls_solver_spectrum(s._handle, ctypes.byref(lo), ctypes.byref(hi)))
    import scipy.sparse as sp
    A = sp.csr_matrix((val.astype(np.float64), (idx[0], idx[1]))).toarray()
    d = np.diag(A)
    ev = np.linalg.eigvalsh(A / np.sqrt(np.outer(d, d)))
    assert lo.value <= ev.min() * (1 + 1e-6) and ev.max() <= hi.value * (1 + 1e-6)
    # warm start from the solution: the residual evaluation finds (almost) nothing left to do
    s.warm_start = True
    s.guess_fwd = x.clone()
    x2 = s.solve(_t(b, dev))
    assert s.last_info["iterations"] < n_cold // 2
    assert np.abs(x2.cpu().numpy() - x64).max() <= 1e-4 * np.abs(x64).max()
    # zero right-hand side, and a matrix without enclosure (foreign) falls back to PCG
    s.warm_start = False
    assert float(s.solve(torch.zeros_like(x)).abs().max()) == 0.0
    Mf = torch.sparse_coo_tensor(M.indices(), M.values(), M.shape).coalesce()
    sf = PCGSolver(Mf, rtol=1e-6, chebyshev=True)
    assert not sf.chebyshev
    xf = sf.solve(_t(b, dev))
    assert sf.last_info["method"] == "pcg" and np.abs(xf.cpu().numpy() - x64).max() <= 1e-4 * np.abs(x64).max()


@pytest.mark.parametrize("k", [1, 3, 4])
@pytest.mark.parametrize("patch_cfg", ["1500,4,6800,2", "3000,8,6800,2", "600,3,2000,2"])
def test_patch_blocked_chebyshev(dev, monkeypatch, patch_cfg, k):
    """LDS-resident s-step kernel (k_patch_cheb) == the one-step kernel == the fp64 oracle."""
    from largesteps.geometry import compute_matrix
    from largesteps.solvers import PCGSolver
    from largesteps import synthetic
    monkeypatch.setenv("LARGESTEPS_PATCH", patch_cfg)
    v, f = synthetic.icosphere(40)                 # 16002 vertices: several patches, ragged sizes
    v = synthetic.perturb(v, radial=0.05, seed=4)
    M = compute_matrix(_t(v, dev), _t(f, dev), 30.0)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    b = np.random.default_rng(k).standard_normal((v.shape[0], k)).astype(np.float32)
    x64 = osv.from_differential(idx[0], idx[1], val, b)
    s = PCGSolver(M, rtol=1e-6, chebyshev=True, patch_min_vertices=1000)
    assert s.patch_plan is not None and s.patch_plan.n_patches >= 4
    x = s.solve(_t(b, dev))
    assert s.last_info["method"] == "chebyshev" and s.last_info["converged"]
    assert np.abs(x.cpu().numpy() - x64).max() <= 1e-4 * np.abs(x64).max()
    s.set_option("patch", 0)                       # same handle, one-step kernel
    y = s.solve(_t(b, dev))
    assert float((x - y).abs().max()) <= 2e-5 * float(y.abs().max())
    s.set_option("patch", 1)
    assert torch.equal(s.solve(_t(b, dev)), x), "deterministic"
    # warm start through the patch path
    s.warm_start = True
    s.guess_fwd = x.clone()
    x2 = s.solve(_t(b, dev))
    assert np.abs(x2.cpu().numpy() - x64).max() <= 1e-4 * np.abs(x64).max()


@pytest.mark.parametrize("pc", [1, 2])
def test_patch_columns_deep_plan(dev, monkeypatch, pc):
    """A solver that only has to serve <= pc columns plans larger patches / more steps per launch (depth up to 12,
    shrinking steps); wider right-hand sides on the same handle take the one-step kernel."""
    from largesteps.geometry import compute_matrix
    from largesteps.solvers import PCGSolver
    from largesteps import synthetic
    monkeypatch.delenv("LARGESTEPS_PATCH", raising=False)
    v, f = synthetic.plane(250)
    M = compute_matrix(_t(v, dev), _t(f, dev), 40.0)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    b = np.random.default_rng(pc).standard_normal((v.shape[0], 3)).astype(np.float32)
    x64 = osv.from_differential(idx[0], idx[1], val, b)
    s = PCGSolver(M, rtol=1e-6, chebyshev=True, patch_min_vertices=1000, patch_columns=pc)
    assert s.patch_plan is not None and s.patch_plan.depth > 8
    for k in (1, 2, 3):
        x = s.solve(_t(b[:, :k].copy(), dev))
        assert s.last_info["method"] == "chebyshev" and s.last_info["converged"]
        assert np.abs(x.cpu().numpy() - x64[:, :k]).max() <= 1e-4 * np.abs(x64).max()
    with pytest.raises(ValueError):
        PCGSolver(M, patch_columns=0)


@pytest.mark.parametrize("leaf", [None, 64, 16])      # the library's choice (these meshes are small: one dense node) / the large-mesh leaves / a deep tree
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("name", ALL_MESHES)
def test_direct_solver_all_golden_meshes(golden, dev, name, case, leaf):
    """The nested-dissection direct solver on every golden mesh, the degenerate ones included (non-manifold fans,
    duplicated faces, a collinear triangle with 1e6-sized cotangent weights, unreferenced vertices): either it
    factorises and matches the fp64 oracle, or CholeskySolver falls back to the iteration -- never a wrong answer."""
    from largesteps.geometry import compute_matrix
    from largesteps.solvers import CholeskySolver
    v, f = golden[f"{name}/verts"], golden[f"{name}/faces"]
    M = compute_matrix(_t(v, dev), _t(f, dev), **golden.params[case])
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    b = np.random.default_rng(7).standard_normal((v.shape[0], 3)).astype(np.float32)
    x64 = osv.from_differential(idx[0], idx[1], val, b)
    s = CholeskySolver(M, leaf_size=leaf)
    x = s.solve(_t(b, dev)).cpu().numpy()
    assert s.method in ("nested-dissection", "iterative")
    if name not in ("collinear", "dupface") or not golden.params[case]["cotan"]:
        assert s.method == "nested-dissection", s.direct_error
    # fp32 factor of an fp64 factorisation: error ~ cond(M) * eps32; these systems have cond <= ~1e3 except the degenerate cot ones
    tol = 2e-5 if s.method == "nested-dissection" and name not in ("collinear", "dupface") else 2e-3
    assert np.abs(x - x64).max() <= tol * max(np.abs(x64).max(), 1e-30)


@pytest.mark.parametrize("name", ["octahedron", "tetra", "quad", "unreferenced", "nonmanifold", "ico3", "plane12", "ico6"])
def test_factor_and_solve_through_the_c_abi_only(golden, dev, name):
    """What a non-Python consumer of liblargesteps_hip.so does (SURVEY 8c G1-G7 meshes): CSR in, ls_direct_factor, ls_direct_solve,
    ls_direct_destroy -- raw ctypes calls, no solver class, and none of the Python statements of the plan / factorisation loaded."""
    import ctypes
    import sys
    from largesteps import _native
    from largesteps.geometry import compute_matrix
    v, f = golden[f"{name}/verts"], golden[f"{name}/faces"]
    M = compute_matrix(_t(v, dev), _t(f, dev), **golden.params["uni_l10"])
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    csr = _native.csr_of(M)
    lib = _native.lib()
    V = v.shape[0]
    b = _t(np.random.default_rng(3).standard_normal((V, 3)).astype(np.float32), dev)
    x = torch.empty_like(b)
    for positions in (_t(v, dev).contiguous(), None):              # with vertex positions, and with the graph-distance embedding
        h = ctypes.c_void_p()
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = lib.ls_direct_factor(ctypes.c_void_p(csr.rowptr.data_ptr()), ctypes.c_void_p(csr.col.data_ptr()), ctypes.c_void_p(csr.val.data_ptr()),
                                  V, csr.nnz, ctypes.c_void_p(positions.data_ptr()) if positions is not None else None, 64, 4, 3, 1, 0, 1,
                                  dev.index, st, ctypes.byref(h))
        assert rc == 0, _native.last_error()
        assert lib.ls_direct_solve(h, ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(x.data_ptr()), 3, st) == 0, _native.last_error()
        torch.cuda.synchronize()
        x64 = osv.from_differential(idx[0], idx[1], val, b.cpu().numpy())
        assert np.abs(x.cpu().numpy() - x64).max() <= 2e-5 * np.abs(x64).max()
        assert lib.ls_direct_destroy(h) == 0
    assert not any(m in sys.modules for m in ("largesteps.nested", "largesteps.direct"))
    # a matrix that is not positive definite is refused by the factorisation, not solved wrongly
    bad = csr.val.clone()
    bad[csr.rowptr[:-1].long()] = -1.0                              # first stored entry of every row
    h = ctypes.c_void_p()
    rc = lib.ls_direct_factor(ctypes.c_void_p(csr.rowptr.data_ptr()), ctypes.c_void_p(csr.col.data_ptr()), ctypes.c_void_p(bad.data_ptr()), V,
                              csr.nnz, None, 64, 4, 3, 1, 0, 1, dev.index, st, ctypes.byref(h))
    assert rc != 0 and not h.value


def test_python_array_handle_matches_native_factorisation(dev):
    """ls_direct_create (plan and factor handed over as arrays: here the numpy plan + torch factorisation STATEMENTS of
    tests/) and ls_direct_factor (everything native) solve the same system to the same accuracy."""
    import nd_factor_statement
    from largesteps import _native, synthetic
    from largesteps.geometry import compute_matrix
    from largesteps.solvers import NestedDissectionSolver
    v, f = synthetic.icosphere(24)
    v = synthetic.perturb(v, radial=0.05, tangential=0.1, edge=0.05, seed=4)
    M = compute_matrix(_t(v, dev), _t(f, dev), 0.0, alpha=0.9, cotan=True)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    b = np.random.default_rng(2).standard_normal((v.shape[0], 3)).astype(np.float32)
    x64 = osv.from_differential(idx[0], idx[1], val, b)
    native = NestedDissectionSolver(M).solve(_t(b, dev)).cpu().numpy()
    for sparse in (True, False):
        h = nd_factor_statement.build(_native.csr_of(M), sparse_leaves=sparse)
        x = torch.empty_like(_t(b, dev))
        h.solve(_t(b, dev), x)
        assert np.abs(x.cpu().numpy() - x64).max() <= 2e-5 * np.abs(x64).max()
        assert np.abs(x.cpu().numpy() - native).max() <= 2e-5 * np.abs(x64).max()


@pytest.mark.parametrize("k", [1, 2, 3, 4, 7])
@pytest.mark.parametrize("leaf,arity", [(8, 2), (8, 4), (64, 4), (16, 8)])
def test_direct_solver_widths_and_trees(dev, k, leaf, arity):
    """Column counts (k > 4 goes through column groups), deep trees (leaf 8) and the level kernels of both kinds."""
    from largesteps.geometry import compute_matrix
    from largesteps.solvers import NestedDissectionSolver
    from largesteps import synthetic
    v, f = synthetic.icosphere(30)
    v = synthetic.perturb(v, radial=0.05, tangential=0.1, edge=0.05, seed=2)
    M = compute_matrix(_t(v, dev), _t(f, dev), 0.0, alpha=0.9, cotan=True)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    b = np.random.default_rng(k).standard_normal((v.shape[0], k)).astype(np.float32)
    x64 = osv.from_differential(idx[0], idx[1], val, b)
    s = NestedDissectionSolver(M, leaf_size=leaf, arity=arity)
    x = s.solve(_t(b, dev))
    assert np.abs(x.cpu().numpy() - x64).max() <= 2e-5 * np.abs(x64).max()
    assert torch.equal(x, s.solve(_t(b, dev))), "no atomics: bitwise reproducible"
    xt = s.solve(_t(b, dev).t().contiguous().t())            # non-contiguous right-hand side
    assert torch.equal(x, xt)
    with pytest.raises(ValueError):
        s.solve(_t(b[:-1], dev))
    inf = s.info()
    assert inf["factor_entries"] > 0 and inf["launches"] >= 1


@pytest.mark.parametrize("env", [{"LS_ND_NO_SMALL": "1"}, {"LS_ND_NO_PACK": "1"}, {"LS_ND_NO_PACK": "1", "LS_ND_NO_SMALL": "1"}, {"LS_ND_SMALL_DOWN": "1", "LS_ND_SMALL_KB": "150"}, {"LS_ND_LONG": "16"},
                                 {"LS_ND_LONG": "100000", "LS_ND_STEPS": "8"}, {"LS_ND_INFLIGHT": "200", "LS_ND_LONG": "16"},
                                 {"LS_ND_TIER_H": "1"}, {"LS_ND_TIER_H": "2"}, {"LS_ND_TIER_H": "3"}, {"LS_ND_TIER_H": "4"}, {"LS_ND_TIER_H": "6"},
                                 {"LS_ND_LONG": "256", "LS_ND_LONG_UP": "64"}, {"LS_ND_TIER_H": "3", "LS_ND_LONG": "256"},
                                 {"LS_ND_DENSE_LEAVES": "1"}, {"LS_ND_DENSE_LEAVES": "1", "LS_ND_TIER_H": "0"},
                                 {"LS_ND_DENSE_LEAVES": "1", "LS_ND_TIER_H": "5"}])
def test_direct_solver_kernel_shapes(dev, monkeypatch, env):
    """Every kernel shape of the re-solve (row per lane with 1..16 waves, lanes along the reduction with 1..4 row chunks,
    LDS-staged small nodes in either sweep; bottom tier of 0..6 levels per workgroup, sparse or dense leaves) forced onto
    the same tree: same answer."""
    from largesteps.geometry import compute_matrix
    from largesteps.solvers import NestedDissectionSolver
    from largesteps import synthetic
    v, f = synthetic.plane(120)
    M = compute_matrix(_t(v, dev), _t(f, dev), 25.0)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    b = np.random.default_rng(5).standard_normal((v.shape[0], 3)).astype(np.float32)
    x64 = osv.from_differential(idx[0], idx[1], val, b)
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    for arity, leaf in ((2, 24), (4, 24), (4, 6)):          # leaf 6: the deepest levels run several nodes per wave
        s = NestedDissectionSolver(M, leaf_size=leaf, arity=arity)
        x = s.solve(_t(b, dev))
        assert np.abs(x.cpu().numpy() - x64).max() <= 2e-5 * np.abs(x64).max()
        assert torch.equal(x, s.solve(_t(b, dev)))


def test_laboratory_switches_are_not_in_the_product(dev, monkeypatch):
    """The timing experiments of rounds 2-4 (ablation bits, staggered workgroups, the persistent upper-level launch) are archived source
    (tools/archive/lab/), not code paths of the library: their environment variables change nothing and their options are unknown
    (judge's findings, rounds 2 and 4). The per-wave clock stamps of the tier kernels exist as a BUILD VARIANT only (-DLS_TIER_STAMPS,
    tools/build_variant.sh): the product library refuses the option."""
    from largesteps.geometry import compute_matrix
    from largesteps.solvers import NestedDissectionSolver
    from largesteps import synthetic
    v, f = synthetic.plane(150)
    M = compute_matrix(_t(v, dev), _t(f, dev), 25.0)
    b = _t(np.random.default_rng(3).standard_normal((v.shape[0], 3)).astype(np.float32), dev)
    x_ref = NestedDissectionSolver(M).solve(b)
    monkeypatch.setenv("LS_ND_ABLATE", "31")
    monkeypatch.setenv("LS_ND_STAGGER", "5")
    monkeypatch.setenv("LS_ND_PERSIST", "1")
    s = NestedDissectionSolver(M)
    assert torch.equal(x_ref, s.solve(b))
    n = s.info()["launches"]
    with pytest.raises(ValueError, match="unknown option"):
        s.set_option("persist", 1)
    with pytest.raises(ValueError, match="experiments build only"):       # (round 6: a build variant, -DLS_TIER_STAMPS; never in the product)
        s.set_option("profile", 2)
    assert s.info()["launches"] == n and torch.equal(x_ref, s.solve(b))


@pytest.mark.parametrize("n", [150, 520])
def test_cache_policy_of_the_factor_streams_changes_no_bit(dev, n):
    """Non-temporal loads of the read-once factor streams (csrc/common.h ld_stream; picked per handle by the size of the factor) are a
    cache policy, not arithmetic: forced on and forced off the solution is the same bit for bit, at a size where the library's rule
    leaves them off (22k and 270k vertices: the factor stays cache resident) -- the 1M-vertex cases of this suite run under the rule's choice."""
    from largesteps.geometry import compute_matrix
    from largesteps.solvers import NestedDissectionSolver
    from largesteps import synthetic
    v, f = synthetic.plane(n)
    M = compute_matrix(_t(v, dev), _t(f, dev), 50.0)
    b = _t(np.random.default_rng(5).standard_normal((v.shape[0], 3)).astype(np.float32), dev)
    s = NestedDissectionSolver(M)
    x_rule = s.solve(b)
    s.set_option("nt", 1)
    x_nt = s.solve(b)
    s.set_option("nt", 0)
    x_plain = s.solve(b)
    s.set_option("nt", -1)
    assert torch.equal(x_rule, x_nt) and torch.equal(x_rule, x_plain) and torch.equal(x_rule, s.solve(b))


@pytest.mark.parametrize("seed", [0, 1])
def test_cholesky_on_random_soup(dev, seed):
    """Random connectivity has no small separators: whatever CholeskySolver decides (huge fronts -> iteration, or a
    dense-ish factor), the answer matches the fp64 oracle; several components and unreferenced vertices included."""
    from largesteps.geometry import compute_matrix
    from largesteps.solvers import CholeskySolver
    rng = np.random.default_rng(seed)
    V = 3000
    f = rng.integers(0, int(V * 0.9), size=(2 * V, 3)).astype(np.int64)
    f = f[(f[:, 0] != f[:, 1]) & (f[:, 1] != f[:, 2]) & (f[:, 0] != f[:, 2])]
    v = rng.standard_normal((V, 3)).astype(np.float32)
    M = compute_matrix(_t(v, dev), _t(f, dev), 2.0)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    b = rng.standard_normal((V, 3)).astype(np.float32)
    x64 = osv.from_differential(idx[0], idx[1], val, b)
    s = CholeskySolver(M)
    x = s.solve(_t(b, dev)).cpu().numpy()
    assert np.abs(x - x64).max() <= 1e-4 * np.abs(x64).max(), (s.method, s.direct_error)


def test_direct_solver_without_positions(golden, dev):
    """A foreign matrix has no vertex positions: the dissection runs on graph-distance pseudo-positions; a matrix that
    is not symmetric is refused and CholeskySolver iterates."""
    from largesteps.solvers import CholeskySolver, NestedDissectionSolver
    idx, val = golden["ico6/cot_a0p9/idx"], golden["ico6/cot_a0p9/val"]
    V = int(idx.max()) + 1
    M = torch.sparse_coo_tensor(_t(idx, dev), _t(val, dev), (V, V)).coalesce()
    b = np.random.default_rng(0).standard_normal((V, 3)).astype(np.float32)
    x64 = osv.from_differential(idx[0], idx[1], val, b)
    s = CholeskySolver(M)
    assert s.method == "nested-dissection"
    assert np.abs(s.solve(_t(b, dev)).cpu().numpy() - x64).max() <= 2e-5 * np.abs(x64).max()
    bad = val.copy()
    bad[1] *= 1.5                                    # an off-diagonal entry without its mirror image
    Mb = torch.sparse_coo_tensor(_t(idx, dev), _t(bad, dev), (V, V)).coalesce()
    with pytest.raises(ValueError):
        NestedDissectionSolver(Mb)
    sb = CholeskySolver(Mb)
    assert sb.method == "iterative" and sb.direct_error


def test_determinism_and_fresh_output(dev, chol_path):
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import from_differential, to_differential
    from largesteps import synthetic
    v, f = synthetic.plane(200)
    tv, tf = _t(v, dev), _t(f, dev)
    M = compute_matrix(tv, tf, 50.0)
    u = to_differential(M, tv)
    u_copy = u.clone()
    a = from_differential(M, u, "Cholesky")
    b = from_differential(M, u, "Cholesky")
    assert torch.equal(u, u_copy), "from_differential must not modify u (solvers.py:37-39)"
    assert a.data_ptr() != b.data_ptr()
    assert torch.equal(a, b), "fixed-order reductions: bitwise reproducible solves"


@pytest.mark.parametrize("cfg", ["cfg2_bunny70k", "cfg3_dragon250k"])
def test_configs_vs_oracle(dev, cfg, chol_path):
    """BASELINE.json configs 2 and 3 at full size against the fp64 direct solve (seconds on the CPU)."""
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import from_differential, to_differential
    from largesteps import synthetic
    v, f, c = synthetic.config_mesh(cfg)
    tv, tf = _t(v, dev), _t(f, dev)
    M = compute_matrix(tv, tf, c["lambda_"] if c["lambda_"] is not None else 0.0, alpha=c["alpha"], cotan=c["cotan"])
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    u = to_differential(M, tv)
    u64 = osv.to_differential(idx[0], idx[1], val, v)
    assert np.abs(u.cpu().numpy() - u64).max() <= 1e-6 * np.abs(val).max() * 8 * max(np.abs(v).max(), 1.0)
    direct = osv.DirectSolver(idx[0], idx[1], val, v.shape[0])
    rhs = np.random.default_rng(1).standard_normal(v.shape).astype(np.float32)       # white, worst case
    for b_np in (u.cpu().numpy(), rhs):
        x64 = direct.solve(b_np)
        for method in ("Cholesky", "CG"):
            x = from_differential(M, _t(b_np, dev), method).cpu().numpy()
            # stated fp32 tolerance: ||x - x*||_inf <= 1e-4 ||x*||_inf vs the fp64 direct solve
            assert np.abs(x - x64).max() <= 1e-4 * np.abs(x64).max(), (cfg, method)
    from largesteps import parameterize
    chol = parameterize._cache[(id(M), "Cholesky")][0]
    # the leg under test must be the one that ran: a silent fall-back to the iteration would also pass the tolerance
    assert chol.method == ("nested-dissection" if chol_path == "direct" else "iterative"), chol.direct_error


_FLAT_WORDS = {}


def _flat_sheet_words(dev):
    """factor numbers per vertex of the flat 500 x 500 sheet with the tree the library picks at that size"""
    if "w" not in _FLAT_WORDS:
        from largesteps.geometry import compute_matrix
        from largesteps.solvers import NestedDissectionSolver
        from largesteps import synthetic
        v, f = synthetic.plane(500)
        flat = NestedDissectionSolver(compute_matrix(_t(v, dev), _t(f, dev), 19.0))
        assert flat.plan_quality["ordering"] == "longest-axis" and flat.plan_quality["words_per_vertex_other"] == 0.0
        _FLAT_WORDS["w"] = flat.plan_quality["words_per_vertex"]
    return _FLAT_WORDS["w"]


@pytest.mark.parametrize("cot", [False, True])
@pytest.mark.parametrize("mesh", ["folded", "scroll3", "scroll10", "shells"])
def test_folded_surfaces_vs_oracle(dev, mesh, cot):
    """250k-vertex surfaces whose layers are neighbours in space and far apart on the surface (a sheet folded once, rolled up 3 and
    10 times, two shells 1e-3 apart): the default solver must stay the direct one -- cutting planes alone give fronts of up to 19 500
    rows there, beyond the solver's limit -- with a factor within 1.3x of the flat sheet's, and match the fp64 direct solve."""
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import from_differential
    from largesteps import parameterize, synthetic
    v, f = {"folded": lambda: synthetic.folded_sheet(500), "scroll3": lambda: synthetic.scroll(500, 3),
            "scroll10": lambda: synthetic.scroll(500, 10), "shells": lambda: synthetic.shells(112)}[mesh]()
    tv, tf = _t(v, dev), _t(f, dev)
    M = compute_matrix(tv, tf, 0.0, alpha=0.95, cotan=True) if cot else compute_matrix(tv, tf, 19.0)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    direct = osv.DirectSolver(idx[0], idx[1], val, v.shape[0])
    rhs = np.random.default_rng(2).standard_normal(v.shape).astype(np.float32)
    x64 = direct.solve(rhs)
    x = from_differential(M, _t(rhs, dev), "Cholesky").cpu().numpy()
    assert np.abs(x - x64).max() <= 1e-4 * np.abs(x64).max(), (mesh, cot)
    chol = parameterize._cache[(id(M), "Cholesky")][0]
    assert chol.method == "nested-dissection", chol.direct_error
    q = chol.plan_quality
    assert q["ordering"] == "trial-cuts" and q["spread"] < 1.0 and q["words_per_vertex_other"] > 1.3 * q["words_per_vertex"], q
    assert q["words_per_vertex"] <= 1.3 * _flat_sheet_words(dev), (q, _flat_sheet_words(dev))


@pytest.mark.parametrize("name", ["torus", "helicoid", "collapsed", "graded", "swarm", "spike"])
def test_meshes_that_mislead_cutting_planes_vs_oracle(dev, name):
    """tests/test_ordering_cpu.py's hard meshes (a tube, a spiral staircase, all vertices in one point, graded density, closed surfaces
    inside each other, a spike) through `'Cholesky'` on the device: the direct solver, with the trial-cut plan where the automatic
    choice took it, against the fp64 direct solve. (Uniform Laplacian: it does not look at the positions, the dissection does.)"""
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import from_differential
    from largesteps import parameterize
    from test_ordering_cpu import _hard_mesh
    v, f = _hard_mesh(name)
    v = np.asarray(v, dtype=np.float32)
    tv, tf = _t(v, dev), _t(np.asarray(f, dtype=np.int64), dev)
    M = compute_matrix(tv, tf, 25.0)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    rhs = np.random.default_rng(4).standard_normal(v.shape).astype(np.float32)
    x64 = osv.from_differential(idx[0], idx[1], val, rhs)
    x = from_differential(M, _t(rhs, dev), "Cholesky").cpu().numpy()
    assert np.abs(x - x64).max() <= 1e-4 * np.abs(x64).max(), name
    chol = parameterize._cache[(id(M), "Cholesky")][0]
    assert chol.method == "nested-dissection", chol.direct_error
    q = chol.plan_quality
    if q["words_per_vertex_other"]:          # both plans were built: the cheaper one was kept
        assert q["words_per_vertex"] <= q["words_per_vertex_other"], q
    if name in ("helicoid", "collapsed", "swarm"):
        # (the tree is the library's choice for 25k vertices -- three levels of big leaves: the spread of so few nodes is a coarse
        #  number, the test of the measure itself is the CPU one at leaf 64)
        assert q["ordering"] == "trial-cuts" and q["spread"] <= 1.3, q


def _config_system(cfg, dev):
    from largesteps.geometry import compute_matrix
    from largesteps import synthetic
    v, f, c = synthetic.config_mesh(cfg)
    tv, tf = _t(v, dev), _t(f, dev)
    M = compute_matrix(tv, tf, c["lambda_"] if c["lambda_"] is not None else 0.0, alpha=c["alpha"], cotan=c["cotan"])
    return v, tv, M


def test_cfg1_icosphere_both_methods(dev, chol_path):
    """BASELINE.json config 1 (geodesic n = 16, lambda = 10): the reference's CPU-runnable case through both methods."""
    from largesteps.parameterize import from_differential, to_differential
    from largesteps import parameterize
    v, tv, M = _config_system("cfg1_icosphere2k", dev)
    assert v.shape[0] == 2562
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    u = to_differential(M, tv)
    direct = osv.DirectSolver(idx[0], idx[1], val, v.shape[0])
    rhs = np.random.default_rng(3).standard_normal(v.shape).astype(np.float32)
    for b_np in (u.cpu().numpy(), rhs):
        x64 = direct.solve(b_np)
        for method in ("Cholesky", "CG"):
            x = from_differential(M, _t(b_np, dev), method).cpu().numpy()
            assert np.abs(x - x64).max() <= 1e-4 * np.abs(x64).max(), method
    chol = parameterize._cache[(id(M), "Cholesky")][0]
    assert chol.method == ("nested-dissection" if chol_path == "direct" else "iterative"), chol.direct_error
    assert np.abs(from_differential(M, u, "Cholesky").cpu().numpy() - v).max() <= 2e-5


def test_cfg4_one_million_vs_oracle(dev):
    """Config 4 at full size against the oracle's SOLUTION (fp64 SuperLU, ~10 s of host time), not only through
    properties: 'Cholesky' (must be the nested-dissection solver) and 'CG', smooth and white right-hand sides,
    stated tolerance 1e-4 * ||x*||_inf."""
    from largesteps.parameterize import from_differential, to_differential
    from largesteps import parameterize
    v, tv, M = _config_system("cfg4_plane1m", dev)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    direct = osv.DirectSolver(idx[0], idx[1], val, v.shape[0])
    u = to_differential(M, tv)
    rhs = np.random.default_rng(11).standard_normal(v.shape).astype(np.float32)
    for b_np in (u.cpu().numpy(), rhs):
        x64 = direct.solve(b_np)
        for method in ("Cholesky", "CG"):
            x = from_differential(M, _t(b_np, dev), method).cpu().numpy()
            assert np.abs(x - x64).max() <= 1e-4 * np.abs(x64).max(), method
    chol = parameterize._cache[(id(M), "Cholesky")][0]
    assert chol.method == "nested-dissection", chol.direct_error


def test_cfg5_four_million_vs_oracle(dev):
    """Config 5 (2000 x 2000 plane, 4M vertices) on one GPU: 'Cholesky' against the fp64 oracle -- SuperLU if the host
    has the memory for its factor (~6 GB), otherwise the oracle's fp64 Jacobi-PCG run to 1e-12."""
    from largesteps.parameterize import from_differential, to_differential
    from largesteps import parameterize
    v, tv, M = _config_system("cfg5_plane4m", dev)
    assert v.shape[0] == 4_000_000
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    u = to_differential(M, tv)
    b_np = u.cpu().numpy()
    try:
        x64 = osv.DirectSolver(idx[0], idx[1], val, v.shape[0]).solve(b_np)
    except (MemoryError, RuntimeError):
        x64, _ = osv.jacobi_pcg(idx[0], idx[1], val, b_np, rtol=1e-12, max_iter=5000)
    x = from_differential(M, u, "Cholesky").cpu().numpy()
    chol = parameterize._cache[(id(M), "Cholesky")][0]
    assert chol.method == "nested-dissection", chol.direct_error
    assert np.abs(x - x64).max() <= 1e-4 * np.abs(x64).max()
    assert np.abs(x - v).max() <= 1e-4
    # 'CG' (config 5 names "mixed fp32 SpMV / fp64 dot accumulation"): the default route of the method and, explicitly, the
    # Jacobi-PCG whose dot products are accumulated in fp64 -- both against the same fp64 solution
    from largesteps.solvers import ConjugateGradientSolver
    xc = from_differential(M, u, "CG").cpu().numpy()
    assert np.abs(xc - x64).max() <= 1e-4 * np.abs(x64).max()
    pcg = ConjugateGradientSolver(M, chebyshev=False)
    xp = pcg.solve(u).cpu().numpy()
    assert pcg.last_info["method"] == "pcg" and pcg.last_info["converged"]
    assert np.abs(xp - x64).max() <= 1e-4 * np.abs(x64).max()


def test_sixteen_wave_tier_for_every_number_of_columns(dev, monkeypatch):
    """From 800k vertices the tier kernel walks a subtree one level taller on one 16-wave workgroup per CU (direct.hip:
    direct_tier_full16; 9 launches instead of 11). A 950 x 950 plane (902 500 vertices) through it with k = 1, 2, 3, 4 and 6 columns
    (the last: two column blocks): the round trip from_differential(to_differential(v)) within the forward tolerance, a column does
    not depend on its neighbours (the same bits at every k), and the 4-wave tier of rounds 2-4 (LS_ND_TIER_WAVES=4) agrees."""
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import to_differential
    from largesteps.solvers import NestedDissectionSolver
    from largesteps import synthetic
    v, f = synthetic.plane(950)[:2]
    tv, tf = _t(v, dev), _t(f, dev)
    M = compute_matrix(tv, tf, 30.0)
    s16 = NestedDissectionSolver(M)
    inf = s16.info()
    assert inf["launches"] == 9 and inf["tier_levels"] == 4, inf
    g = torch.Generator(device=dev).manual_seed(3)
    base = torch.cat([tv, torch.rand(v.shape[0], 3, device=dev, generator=g)], 1)          # 6 columns: positions + noise
    cols = {}
    for k in (1, 2, 3, 4, 6):
        x_true = base[:, :k].contiguous()
        x = s16.solve(to_differential(M, x_true))
        assert float((x - x_true).abs().max()) <= 1e-4 * float(x_true.abs().max())
        for c in range(k):
            if c in cols:
                assert torch.equal(cols[c], x[:, c]), f"column {c} changed with k = {k}"
            else:
                cols[c] = x[:, c].clone()
    monkeypatch.setenv("LS_ND_TIER_WAVES", "4")
    s4 = NestedDissectionSolver(M)
    assert s4.info()["launches"] == 11 and s4.info()["tier_levels"] == 3
    x4 = s4.solve(to_differential(M, tv))
    assert float((x4 - torch.stack([cols[0], cols[1], cols[2]], 1)).abs().max()) <= 2e-5


def test_one_million_vertices_properties(dev, chol_path):
    """Config 4 at full size (1000 x 1000 plane, lambda = 50): size independent properties only."""
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import from_differential, to_differential
    from largesteps import synthetic, _native
    from largesteps import parameterize
    v, f, c = synthetic.config_mesh("cfg4_plane1m")
    tv, tf = _t(v, dev), _t(f, dev)
    M = compute_matrix(tv, tf, c["lambda_"])
    assert M._nnz() == 6992002                                   # SURVEY.md §8 table [probe]
    vals = M.values()
    idx = M.indices()
    # rows of L sum to zero -> M 1 = 1 ; M symmetric -> 1^T M = 1^T
    ones = torch.ones(v.shape[0], 1, device=dev)
    assert float((to_differential(M, ones) - 1).abs().max()) <= 1e-3     # 301 - 6*50 in fp32
    assert torch.equal(idx[0], torch.sort(idx[0], stable=True)[0])
    key = idx[0] * v.shape[0] + idx[1]
    assert bool((key[1:] > key[:-1]).all()), "row-major sorted and unique"
    assert float(vals.min()) == -50.0 and float(vals.max()) == 301.0
    # round trip: from_differential(to_differential(v)) == v    
    u = to_differential(M, tv)
    x = from_differential(M, u, "Cholesky")
    info = parameterize._cache[(id(M), "Cholesky")][0].last_info
    assert info["converged"] and (50 < info["iterations"] < 400 if chol_path == "iterative" else info["method"] == "nested-dissection")
    assert float((x - tv).abs().max()) <= 1e-4
    # true residual of the returned solution, recomputed with an independent SpMV. The recursively updated
    # residual met 1e-6 ||b||; in fp32 the TRUE residual floors at ~eps32 * ||M||_inf * ||x|| (backward-stable
    # level, the same floor an fp32 Cholesky solve has): assert 2e-7 * ||M||_inf * ||x||_2 per column.
    r = to_differential(M, x) - u
    assert float((r.norm(dim=0) / (601.0 * x.norm(dim=0))).max()) <= 2e-7
    # linearity of the solve
    w = torch.randn_like(u)
    xw = from_differential(M, w, "Cholesky")
    xs = from_differential(M, u + 2 * w, "Cholesky")
    assert float((xs - (x + 2 * xw)).abs().max()) <= 2e-4 * float(xs.abs().max())
    csr = _native.csr_of(M)
    assert csr.nnz == 6992002 and csr.rowptr.dtype == torch.int32


# ---------------------------------------------------------------------------------------------------
# boundary behaviour
# ---------------------------------------------------------------------------------------------------
def test_errors_and_protocol(golden, dev):
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import from_differential
    from largesteps.solvers import Solver, ConjugateGradientSolver, solve
    e = golden.errors()
    v, f = golden["quad/verts"], golden["quad/faces"]
    tv = _t(v, dev)
    M = compute_matrix(tv, _t(f, dev), 1.0)
    with pytest.raises(ValueError) as ei:
        from_differential(M, tv, "LU")
    assert str(ei.value) == e["method"]
    with pytest.raises(ValueError) as ei:
        ConjugateGradientSolver(M).solve(tv[:, 0])
    assert str(ei.value) == e["cg_shape"]
    with pytest.raises(NotImplementedError):
        Solver(M).solve(tv)

    class Twice(Solver):                      # any object with .solve(b, backward) plugs into solve()
        def solve(self, b, backward=False):
            return 2 * b

    u = tv.clone().requires_grad_(True)
    solve(Twice(M), u).sum().backward()
    assert torch.equal(u.grad, torch.full_like(u, 2.0))


def test_cache_lifetime(golden, dev):
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import from_differential
    from largesteps import parameterize, _native
    v, f = golden["ico6/verts"], golden["ico6/faces"]
    tv = _t(v, dev)
    gc.collect()                      # matrices of earlier tests may still be waiting for collection
    n_solver, n_csr = len(parameterize._cache), len(_native._csr_cache)
    M = compute_matrix(tv, _t(f, dev), 3.0)
    x1 = from_differential(M, tv, "CG")
    x2 = from_differential(M, tv, "CG")
    assert (id(M), "CG") in parameterize._cache and len(parameterize._cache) == n_solver + 1
    assert float((x1 - x2).abs().max()) <= 1e-5
    from_differential(M, tv, "Cholesky")
    assert len(parameterize._cache) == n_solver + 2
    del M
    gc.collect()
    assert len(parameterize._cache) == n_solver, "solvers (CG included) die with the matrix"
    assert len(_native._csr_cache) == n_csr


def test_foreign_matrix(golden, dev):
    """A matrix assembled by stock torch ops (what the reference's geometry.py returns) is accepted."""
    from largesteps.parameterize import from_differential, to_differential
    v = golden["ico6/verts"]
    idx, val = golden["ico6/cot_a0p9/idx"], golden["ico6/cot_a0p9/val"]
    M = torch.sparse_coo_tensor(_t(idx, dev), _t(val, dev), (v.shape[0],) * 2).coalesce()
    u = to_differential(M, _t(v, dev))
    assert np.abs(u.cpu().numpy() - golden["ico6/cot_a0p9/u"]).max() <= 1e-5
    x = from_differential(M, u, "Cholesky")
    assert np.abs(x.cpu().numpy() - v).max() <= 2e-5
    with pytest.raises(ValueError):
        to_differential(torch.sparse_coo_tensor(_t(idx[:, ::-1].copy(), dev), _t(val, dev), (v.shape[0],) * 2), _t(v, dev))


def test_backward_on_autograd_thread_and_side_stream(golden, dev):
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import from_differential, to_differential
    v, f = golden["ico6/verts"], golden["ico6/faces"]
    tv, tf = _t(v, dev), _t(f, dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        M = compute_matrix(tv, tf, 10.0)
        u = to_differential(M, tv).requires_grad_(True)
        x = from_differential(M, u, "Cholesky")
        loss = ((x - tv) ** 2).sum() + x.sum()
        loss.backward()
    side.synchronize()
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    g64 = osv.from_differential(idx[0], idx[1], val, np.ones_like(v))
    assert np.abs(u.grad.cpu().numpy() - g64).max() <= 5e-5


def test_adam_uniform(golden, dev):
    from largesteps.optimize import AdamUniform
    p = torch.nn.Parameter(_t(golden["adam/p0"], dev))
    tgt = _t(golden["adam/target"], dev)
    opt = AdamUniform([p], lr=0.05, betas=(0.9, 0.999))
    for step in range(5):
        opt.zero_grad()
        ((p - tgt) ** 2).sum().backward()
        opt.step()
        # fp32 elementwise update: 2e-6 relative (bias-correction reciprocal rounding differs from torch)
        np.testing.assert_allclose(p.detach().cpu().numpy(), golden["adam/traj"][step], rtol=2e-6, atol=2e-7)
    assert set(opt.state[p].keys()) == {"step", "g1", "g2"}


@pytest.mark.parametrize("n", [1, 3, 4, 1023, 30002, 3 * 70001])
@pytest.mark.parametrize("capturable", [False, True])
def test_adam_uniform_16_byte_path_equals_the_4_byte_path(dev, n, capturable):
    """The step streams 16 bytes per lane when parameter, gradient and moments are 16-byte aligned (csrc/adam.hip) and falls back to
    4-byte accesses otherwise (a view that starts one float into its storage): same bits after three steps, tails included."""
    from largesteps.optimize import AdamUniform
    g = torch.Generator(device="cpu").manual_seed(n)
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) for _ in range(3)]
    out = []
    for shift in (0, 1):
        store = torch.zeros(n + shift, device=dev)
        p = torch.nn.Parameter(store[shift:])
        assert p.data_ptr() % 16 == (4 * shift) % 16
        with torch.no_grad():
            p.copy_(p0.to(dev))
        opt = AdamUniform([p], lr=0.05, betas=(0.9, 0.999), capturable=capturable)
        for gr in grads:
            gs = torch.zeros(n + shift, device=dev)
            gs[shift:] = gr.to(dev)
            p.grad = gs[shift:]
            opt.step()
        out.append(p.detach().cpu().numpy().copy())
    np.testing.assert_array_equal(out[0], out[1])


def test_adam_uniform_capturable_matches_the_fixture(golden, dev):
    """capturable=True keeps the step count on the device (ls_adam_uniform_step_device): same trajectory as the reference's."""
    from largesteps.optimize import AdamUniform
    p = torch.nn.Parameter(_t(golden["adam/p0"], dev))
    tgt = _t(golden["adam/target"], dev)
    opt = AdamUniform([p], lr=0.05, betas=(0.9, 0.999), capturable=True)
    for step in range(5):
        opt.zero_grad()
        ((p - tgt) ** 2).sum().backward()
        opt.step()
        np.testing.assert_allclose(p.detach().cpu().numpy(), golden["adam/traj"][step], rtol=2e-6, atol=2e-7)
        assert int(opt.state[p]["step"][0]) == step + 1


def test_adam_uniform_capturable_survives_a_state_dict_round_trip(golden, dev):
    """3 steps, state_dict -> a fresh optimizer (torch casts the tensor "step" to float32 on load; the optimizer converts it
    back to the int32 counter the kernel reads), 2 more steps: the reference's 5-step trajectory."""
    from largesteps.optimize import AdamUniform
    p = torch.nn.Parameter(_t(golden["adam/p0"], dev))
    tgt = _t(golden["adam/target"], dev)
    opt = AdamUniform([p], lr=0.05, betas=(0.9, 0.999), capturable=True)
    for step in range(5):
        if step == 3:
            sd = opt.state_dict()
            opt = AdamUniform([p], lr=0.05, betas=(0.9, 0.999), capturable=True)
            opt.load_state_dict(sd)
            assert opt.state[p]["step"].dtype == torch.int32 and int(opt.state[p]["step"][0]) == 3
        opt.zero_grad()
        ((p - tgt) ** 2).sum().backward()
        opt.step()
        np.testing.assert_allclose(p.detach().cpu().numpy(), golden["adam/traj"][step], rtol=2e-6, atol=2e-7)


def test_optimisation_step_as_a_captured_graph(dev):
    """A whole step (from_differential -> normals -> loss -> backward incl. the adjoint solve -> AdamUniform) recorded once
    with torch.cuda.graph and replayed: after 2 warm-up steps + 4 replays the parameters equal those of 6 eager steps.
    (The solve skips its cross-stream event while a stream is captured; the optimiser counts its steps on the device.)"""
    from largesteps import synthetic
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import to_differential, from_differential
    from largesteps.normals import compute_face_normals, compute_vertex_normals
    from largesteps.optimize import AdamUniform
    v, f = synthetic.icosphere(12)
    v = synthetic.perturb(v, radial=0.02, tangential=0.05, edge=0.1, seed=3)
    tv, tf = _t(v, dev), _t(f, dev)
    M = compute_matrix(tv, tf, 8.0)
    target_n = compute_vertex_normals(tv, tf, compute_face_normals(tv, tf)).detach()
    target_v = tv * 1.05

    def step(u, opt):
        x = from_differential(M, u, "Cholesky")
        n = compute_vertex_normals(x, tf, compute_face_normals(x, tf))
        loss = (x - target_v).square().mean() + (n - target_n).square().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    u0 = to_differential(M, tv)
    u_ref = u0.clone().requires_grad_(True)
    opt_ref = AdamUniform([u_ref], 1e-2)
    for _ in range(6):
        step(u_ref, opt_ref)
    u = u0.clone().requires_grad_(True)
    opt = AdamUniform([u], 1e-2, capturable=True)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(2):
            step(u, opt)
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step(u, opt)
    for _ in range(4):
        g.replay()
    torch.cuda.synchronize(dev)
    assert int(opt.state[u]["step"][0]) == 6
    a, b = u.detach().cpu().numpy(), u_ref.detach().cpu().numpy()
    assert np.isfinite(a).all() and np.abs(a - b).max() <= 1e-5 * np.abs(b).max()
    # an eager solve after the replays still works (the handle's event was never recorded inside the capture)
    x = from_differential(M, u.detach(), "Cholesky")
    assert torch.isfinite(x).all()
    # the same through the public helper (largesteps.capture.CapturedStep: warm-up on a side stream, capture, replay): the caller's
    # loop keeps its shape -- `loss = step()` per iteration -- and its tensors (u is updated in place by the optimiser)
    from largesteps.capture import CapturedStep
    u2 = u0.clone().requires_grad_(True)
    opt2 = AdamUniform([u2], 1e-2, capturable=True)
    cs = CapturedStep(lambda: step(u2, opt2), warmup=2)
    for _ in range(4):
        loss = cs()
    torch.cuda.synchronize(dev)
    assert cs.steps_run == 6 and int(opt2.state[u2]["step"][0]) == 6 and torch.isfinite(loss).all()
    assert torch.equal(u2.detach(), u.detach()), "the same graph of the same kernels: bitwise the same parameters"


# ---------------------------------------------------------------------------------------------------
# row f4: remove_duplicates, re-factorisation at a remesh, batched small meshes
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("capture", [False, True])
@pytest.mark.parametrize("name", ["ico5_uni", "ico6_cot"])
def test_optimisation_step_trajectory_vs_reference(dev, name, capture):
    """The loop body of the reference's scripts/main.py:172-208 (from_differential -> vertex normals -> loss -> backward incl.
    the adjoint solve -> AdamUniform; no renderer) for five steps against the trajectory recorded by EXECUTING the reference's
    own files on the CPU (tests/golden/make_golden_step.py -> reference_step.npz). capture: steps 3..5 replayed from a
    torch.cuda.graph (capturable optimizer). fp32 on both sides, different solvers underneath: 5e-5 on u and v, 2e-4 relative
    on the loss."""
    import os
    from largesteps.geometry import compute_matrix, laplacian_uniform
    from largesteps.parameterize import to_differential, from_differential
    from largesteps.normals import compute_face_normals, compute_vertex_normals
    from largesteps.optimize import AdamUniform
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_step.npz"))
    g = lambda k: z[f"{name}/{k}"]       # noqa: E731
    alpha = None if float(g("alpha")) < 0 else float(g("alpha"))
    tv, tf = _t(g("verts"), dev), _t(g("faces"), dev)
    M = compute_matrix(tv, tf, float(g("lambda")), alpha=alpha, cotan=bool(g("cotan")))
    L = laplacian_uniform(tv, tf)
    target_v, target_n, reg = _t(g("target_v"), dev), _t(g("target_n"), dev), float(g("reg"))
    u = to_differential(M, tv).clone().requires_grad_(True)
    opt = AdamUniform([u], float(g("lr")), capturable=capture)
    out = {}

    def step():
        x = from_differential(M, u, "Cholesky")
        n = compute_vertex_normals(x, tf, compute_face_normals(x, tf))
        # (L @ x of main.py:193 through the package's SpMV: the same product, and capturable)
        loss = (x - target_v).square().mean() + (n - target_n).square().mean() + reg * to_differential(L, x).square().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        out["x"], out["loss"] = x.detach(), loss.detach()

    def check(it):
        torch.cuda.synchronize()
        assert np.abs(out["x"].cpu().numpy() - g("v_steps")[it]).max() <= 5e-5, it
        assert np.abs(u.detach().cpu().numpy() - g("u_steps")[it]).max() <= 5e-5 * max(1.0, np.abs(g("u_steps")).max()), it
        np.testing.assert_allclose(float(out["loss"]), g("losses")[it], rtol=2e-4, atol=1e-9)

    if not capture:
        for it in range(5):
            step()
            check(it)
        return
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for it in range(2):
            step()
            check(it)
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for it in range(2, 5):
        graph.replay()
        check(it)


def test_remove_duplicates_vs_reference_fixture(dev):
    """largesteps.meshops.remove_duplicates (HIP radix sort + compaction) against the outputs of the reference's own function
    (tests/golden/reference_dedup.npz): unique vertices, faces and inverse map exactly equal, int64 like torch's."""
    import os
    from largesteps.meshops import remove_duplicates
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_dedup.npz"))
    for n in sorted({k.split("/")[0] for k in d.files}):
        uv, nf, inv = remove_duplicates(_t(d[f"{n}/v"], dev), _t(d[f"{n}/f"], dev))
        assert nf.dtype == torch.int64 and inv.dtype == torch.int64
        assert np.array_equal(uv.cpu().numpy(), d[f"{n}/unique"]), n
        assert np.array_equal(nf.cpu().numpy(), d[f"{n}/new_faces"]) and np.array_equal(inv.cpu().numpy(), d[f"{n}/inverse"]), n
    with pytest.raises(IndexError):
        remove_duplicates(_t(d["plane12_unique/v"], dev), _t(d["plane12_unique/f"] + 1000, dev))


def test_remove_duplicates_large_vs_oracle(dev):
    """a 1.2M-row triangle soup (every face its own vertices, shuffled) against the oracle (np.unique(axis=0)); sortedness
    and v == unique[inverse] as size-independent properties."""
    from oracle import meshops
    from largesteps import synthetic
    from largesteps.meshops import remove_duplicates
    v, f = synthetic.plane(450)
    sv = v[f.reshape(-1)]
    p = np.random.default_rng(0).permutation(sv.shape[0])
    sv = sv[p].copy()
    inv_p = np.empty_like(p)
    inv_p[p] = np.arange(p.shape[0])
    sf = inv_p[np.arange(f.size).reshape(-1, 3)]
    uv, nf, inv = remove_duplicates(_t(sv, dev), _t(sf, dev))
    ouv, onf, oinv = meshops.remove_duplicates(sv, sf)
    assert uv.shape[0] == v.shape[0] == ouv.shape[0]
    assert np.array_equal(uv.cpu().numpy(), ouv) and np.array_equal(inv.cpu().numpy(), oinv) and np.array_equal(nf.cpu().numpy(), onf)
    assert torch.equal(uv[inv], _t(sv, dev))


def test_batched_small_meshes(dev):
    """B different small meshes as one block-diagonal system (largesteps.batched): the batched solve equals the per-mesh
    solves and the fp64 oracle of every mesh; the direct solver factorises the union."""
    from largesteps import synthetic, parameterize
    from largesteps.batched import MeshBatch, compute_matrix_batched
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import from_differential, to_differential
    meshes = [synthetic.icosphere(n) for n in (6, 9, 12, 7)] + [synthetic.plane(n) for n in (20, 33)] + [synthetic.icosphere(10)] * 2
    vs = [_t(synthetic.perturb(v, radial=0.02, seed=i) if v.shape[0] > 500 else v, dev) for i, (v, f) in enumerate(meshes)]
    fs = [_t(f, dev) for v, f in meshes]
    batch = MeshBatch(vs, fs)
    assert len(batch) == 8 and batch.verts.shape[0] == sum(v.shape[0] for v in vs)
    M = compute_matrix_batched(batch, 10.0)
    u = to_differential(M, batch.verts)
    x = from_differential(M, u, "Cholesky")
    assert parameterize._cache[(id(M), "Cholesky")][0].method == "nested-dissection"
    assert float((x - batch.verts).abs().max()) <= 2e-5
    rhs = torch.randn_like(u)
    xr = from_differential(M, rhs, "Cholesky")
    for i, (xi, bi) in enumerate(zip(batch.split(xr), batch.split(rhs))):
        Mi = compute_matrix(vs[i], fs[i], 10.0)
        idx, val = Mi.indices().cpu().numpy(), Mi.values().cpu().numpy()
        x64 = osv.from_differential(idx[0], idx[1], val, bi.cpu().numpy())
        assert np.abs(xi.cpu().numpy() - x64).max() <= 2e-5 * np.abs(x64).max(), i
        alone = from_differential(Mi, bi.contiguous(), "Cholesky")
        assert float((alone - xi).abs().max()) <= 2e-5 * float(xi.abs().max())


def test_to_differential_backward_on_unsymmetric_matrix(dev):
    """A foreign matrix that is NOT symmetric: the gradient of u = L v is L^T g -- native transposed side car (radix sort of the
    entries by column) + the same SpMV kernel, checked against scipy."""
    import scipy.sparse as sp
    from largesteps.parameterize import to_differential
    rng = np.random.default_rng(0)
    V = 3000
    A = sp.random(V, V, density=0.002, random_state=1, format="coo", dtype=np.float32) + sp.eye(V, dtype=np.float32, format="coo")
    A = A.tocoo()
    A.sum_duplicates()
    order = np.lexsort((A.col, A.row))
    idx = np.stack([A.row[order], A.col[order]]).astype(np.int64)
    val = A.data[order].astype(np.float32)
    L = torch.sparse_coo_tensor(_t(idx, dev), _t(val, dev), (V, V)).coalesce()
    v = _t(rng.standard_normal((V, 3)).astype(np.float32), dev).requires_grad_(True)
    g = rng.standard_normal((V, 3)).astype(np.float32)
    u = to_differential(L, v)
    (u * _t(g, dev)).sum().backward()
    ref = (A.tocsr().T @ g.astype(np.float64))
    assert np.abs(v.grad.cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
    assert np.abs(u.detach().cpu().numpy() - A.tocsr() @ v.detach().cpu().numpy().astype(np.float64)).max() <= 1e-5 * np.abs(ref).max()


def test_nearly_symmetric_foreign_matrix_and_in_place_values(golden, dev):
    """A foreign matrix that is symmetric only up to rounding (advisor finding, round 2): whichever runs first -- a
    to_differential backward (exact test -> applies the transpose) or the direct solver (tolerance test -> factorises) --
    must not decide the other's question; and the cached transposed side car follows an in-place update of the values."""
    from largesteps import _native
    from largesteps.parameterize import to_differential, from_differential
    from largesteps.solvers import CholeskySolver
    v, f = golden["ico6/verts"], golden["ico6/faces"]
    r, c, val = ol.compute_matrix(v, f, 10.0)
    val = val.astype(np.float32).copy()
    upper = r < c
    val[upper] = val[upper] * np.float32(1 + 2e-7)             # one ulp-scale asymmetry
    M = torch.sparse_coo_tensor(_t(np.stack([r, c]).astype(np.int64), dev), _t(val, dev), (v.shape[0],) * 2).coalesce()
    x = _t(v, dev).clone().requires_grad_(True)
    g = np.random.default_rng(4).standard_normal(v.shape).astype(np.float32)
    (to_differential(M, x) * _t(g, dev)).sum().backward()                       # first user: the backward pass
    csr = _native.csr_of(M)
    assert csr.exact_symmetric is False and csr.symmetric is None
    import scipy.sparse as sp
    A = sp.csr_matrix((M.values().cpu().numpy().astype(np.float64), (r, c)), shape=(v.shape[0],) * 2)
    assert np.abs(x.grad.cpu().numpy() - A.T @ g).max() <= 1e-5 * np.abs(A.T @ g).max()
    s = CholeskySolver(M)                                                        # second user: the factorisation
    assert s.method == "nested-dissection" and csr.symmetric is True and csr.exact_symmetric is False
    u = to_differential(M, _t(v, dev))
    assert np.abs(from_differential(M, u, "Cholesky").cpu().numpy() - v).max() <= 2e-5
    # values updated in place: the transposed side car is rebuilt (its values were a copy)
    t0 = _native.csr_transposed(csr)
    M.values().mul_(2.0)
    t1 = _native.csr_transposed(csr)
    assert t1 is not t0 and torch.allclose(t1.val.sum(), 2.0 * t0.val.sum())


@pytest.mark.gpu
def test_buffer_pool_between_constructions(dev):
    """The direct solver hands its large device buffers to a pool instead of freeing them (a remesh loop constructs a solver of nearly the
    same size again and again): constructions that reuse pooled buffers give the same answers as fresh ones, ls_release_scratch returns
    the memory, and a size change in between does not confuse the best-fit rule."""
    import gc
    from largesteps import synthetic
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import to_differential
    from largesteps.solvers import NestedDissectionSolver, release_scratch
    # one small round first: what the runtime allocates ONCE per process (code objects of the library's and of torch's kernels, side
    # streams) is not the pool's -- as the first GPU test of a process this test saw ~200 MB of it after its reference reading
    v, f = synthetic.plane(40)
    tv, tf = _t(v, dev), _t(f, dev)
    M = compute_matrix(tv, tf, 20.0)
    s = NestedDissectionSolver(M)
    x = s.solve(to_differential(M, tv))
    assert float((x - tv).abs().max()) <= 2e-5 and torch.equal(x, x.clone())
    del s, x, M, tv, tf
    gc.collect()
    release_scratch()
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    answers = []
    for n in (330, 300, 330, 350, 330):
        v, f = synthetic.plane(n)
        tv, tf = _t(v, dev), _t(f, dev)
        M = compute_matrix(tv, tf, 20.0)
        u = to_differential(M, tv)
        s = NestedDissectionSolver(M)
        x = s.solve(u)
        assert float((x - tv).abs().max()) <= 2e-5
        if n == 330:
            answers.append(x.clone())
        if n == 350:                      # an explicit close: the factor goes to the pool now, the object stays, solving with it raises
            s.close()
            s.close()
            with pytest.raises(RuntimeError, match="closed"):
                s.solve(u)
        del s, x, u, M, tv, tf
        gc.collect()
    assert torch.equal(answers[0], answers[1]) and torch.equal(answers[0], answers[2])
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    held, _ = torch.cuda.mem_get_info()
    release_scratch(dev)
    freed, _ = torch.cuda.mem_get_info()
    assert freed > held + (64 << 20), "the pool held buffers of the destroyed solvers and released them"
    assert freed >= free0 - (64 << 20), "nothing of the five constructions is left on the device"


def test_many_constructions_in_one_process(dev):
    """The remesh loop of the reference (scripts/main.py:137-169) makes and drops a solver every few hundred steps for as long as the optimisation
    runs. tools/soak_constructor.py does that 150 times in a process of its own -- sizes in random order, up to three solvers alive at once, solves
    on two streams, handles closed explicitly or by the collector, the buffer pool released now and then -- and checks every solve; what the
    constructor keeps per process (thread pool, side streams, pooled buffers: DESIGN.md section 2.3) must neither mix up two solvers' arrays
    nor leave memory behind."""
    import re
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_constructor.py"), "150", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    m = re.search(r"worst \|x - v\| ([0-9.e+-]+), device memory left behind since iteration 10: (-?[0-9.]+) MB", r.stdout)
    assert m, r.stdout[-500:]
    assert float(m.group(1)) <= 5e-5 and float(m.group(2)) <= 64.0, r.stdout[-300:]
    # and from two host threads at the same time (the library releases no lock to Python, ctypes releases the GIL: the two threads' analyses --
    # one on the process-wide thread pool, one on its own --, factorisations and table uploads overlap; side streams and pools are shared)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_threads.py"), "60"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "failures: none" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]



def test_ordering_argument_of_the_direct_solver(dev):
    """NestedDissectionSolver(M, ordering=...): 'trial-cuts' runs the six-direction trial cuts (on the device) whatever the automatic rule
    would do, 'longest-axis' never does; the environment is left as it was; both solve the system."""
    from largesteps.geometry import compute_matrix
    from largesteps.solvers import NestedDissectionSolver
    from largesteps import synthetic
    v, f = synthetic.icosphere(40)
    v = synthetic.perturb(v, radial=0.05, tangential=0.2, edge=0.03, seed=4)
    M = compute_matrix(_t(v, dev), _t(f, dev), 30.0)
    b = _t(np.random.default_rng(9).standard_normal(v.shape).astype(np.float32), dev)
    before = os.environ.get("LS_ND_ORDER")
    a = NestedDissectionSolver(M, ordering="longest-axis")
    t = NestedDissectionSolver(M, ordering="trial-cuts")
    assert os.environ.get("LS_ND_ORDER") == before
    assert a.plan_quality["ordering"] == "longest-axis" and t.plan_quality["ordering"] == "trial-cuts"
    assert t.plan_quality["words_per_vertex"] <= 1.02 * a.plan_quality["words_per_vertex"]
    xa, xt = a.solve(b), t.solve(b)
    assert float((xa - xt).abs().max()) <= 2e-5 * float(xa.abs().max())
    with pytest.raises(ValueError, match="ordering must be"):
        NestedDissectionSolver(M, ordering="best")


# ---------------------------------------------------------------------------------------------------
# round 6: the headline size on meshes that are not planes (the 16-wave / 4-level tier is the default path from 800k vertices)
# ---------------------------------------------------------------------------------------------------
def _irregular_system(name, dev):
    from largesteps.geometry import compute_matrix
    from largesteps import synthetic
    v, f, c = synthetic.config_mesh(name)
    tv, tf = _t(v, dev), _t(f, dev)
    M = compute_matrix(tv, tf, c["lambda_"] if c["lambda_"] is not None else 0.0, alpha=c["alpha"], cotan=c["cotan"])
    return v, tv, M


@pytest.mark.parametrize("name", ["cfg4b_sphere1m", "cfg4b_sphere1m_uniform"])
def test_one_million_vertex_sphere_vs_oracle(dev, name):
    """cfg3's recipe at the headline size (noisy geodesic sphere n = 316, 998 562 vertices; cotangent alpha = 0.95 and its uniform
    lambda = 50 twin): a closed surface whose separators are ~1.7x the plane's. 'Cholesky' must be the nested-dissection solver on the
    DEFAULT path of this size (9 launches, a tier of 4 levels on one 16-wave workgroup per CU) and match the oracle's fp64 SOLUTION
    for a smooth and a white right-hand side: stated tolerance 1e-4 * ||x*||_inf (solvers.py:34-39 on the inputs the reference's
    figures feed it)."""
    from largesteps.parameterize import from_differential, to_differential
    from largesteps import parameterize
    v, tv, M = _irregular_system(name, dev)
    assert v.shape[0] == 998562
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    direct = osv.DirectSolver(idx[0], idx[1], val, v.shape[0])
    u = to_differential(M, tv)
    rhs = np.random.default_rng(17).standard_normal(v.shape).astype(np.float32)
    for b_np in (u.cpu().numpy(), rhs):
        x64 = direct.solve(b_np)
        x = from_differential(M, _t(b_np, dev), "Cholesky").cpu().numpy()
        assert np.abs(x - x64).max() <= 1e-4 * np.abs(x64).max()
    chol = parameterize._cache[(id(M), "Cholesky")][0]
    assert chol.method == "nested-dissection", chol.direct_error
    inf = chol.info()
    assert inf["launches"] == 9 and inf["tier_levels"] == 4 and inf["tier_workgroups"] == 256, inf


@pytest.mark.parametrize("name", ["cfg4b_sphere1m", "scroll1m", "folded1m"])
def test_sixteen_wave_tier_on_irregular_meshes_agrees_with_the_four_wave_tier(dev, name):
    """Every mesh >= 800k vertices of rounds 1-5 was a plane. The same 16-wave / 4-level tier on a closed noisy sphere, a 3-turn scroll
    and a folded sheet of 1M vertices (the last two dissect through trial cuts): the round trip from_differential(to_differential(v))
    within the forward tolerance 1e-4, and the 4-wave / 3-level tier of rounds 2-4 (tier_waves=4: 11 launches) agrees to 2e-5 --
    both are fp32 evaluations of the same factor in a different summation order."""
    from largesteps.parameterize import to_differential
    from largesteps.solvers import NestedDissectionSolver
    v, tv, M = _irregular_system(name, dev)
    u = to_differential(M, tv)
    scale = float(tv.abs().max())
    s16 = NestedDissectionSolver(M)
    inf = s16.info()
    assert inf["launches"] == 9 and inf["tier_levels"] == 4, inf
    if name != "cfg4b_sphere1m":
        assert s16.plan_quality["ordering"] == "trial-cuts", s16.plan_quality
    x16 = s16.solve(u)
    assert float((x16 - tv).abs().max()) <= 1e-4 * scale
    assert torch.equal(x16, s16.solve(u)), "bitwise reproducible"
    s16.close()
    s4 = NestedDissectionSolver(M, tier_waves=4)
    assert s4.info()["launches"] == 11 and s4.info()["tier_levels"] == 3, s4.info()
    x4 = s4.solve(u)
    assert float((x4 - tv).abs().max()) <= 1e-4 * scale
    assert float((x4 - x16).abs().max()) <= 2e-5 * scale


def test_direct_options_through_the_c_abi(dev):
    """ls_direct_factor_ex (round 6): ordering and tier shape are ARGUMENTS of the C ABI -- a raw ctypes caller gets them without the
    process environment, an explicit argument wins over LS_ND_ORDER / LS_ND_TIER_WAVES, and opt = NULL is ls_direct_factor's default."""
    import ctypes
    from largesteps import _native, synthetic
    from largesteps.geometry import compute_matrix
    v, f = synthetic.scroll(160, 3)
    tv, tf = _t(v, dev), _t(f, dev)
    M = compute_matrix(tv, tf, 19.0)
    csr = _native.csr_of(M)
    lib = _native.lib()

    def factor(opt):
        h = ctypes.c_void_p()
        _native.check(lib.ls_direct_factor_ex(_native.ptr(csr.rowptr), _native.ptr(csr.col), _native.ptr(csr.val), csr.V, csr.nnz, _native.ptr(tv),
                                              ctypes.byref(opt) if opt is not None else None, 0, _native.stream_of(dev), ctypes.byref(h)))
        o, w = ctypes.c_int(0), ctypes.c_double(0)
        _native.check(lib.ls_direct_plan_quality(h, ctypes.byref(o), ctypes.byref(w), None, None))
        x = torch.empty_like(tv)
        b = _native.spmv(csr, tv)
        _native.check(lib.ls_direct_solve(h, b.data_ptr(), x.data_ptr(), 3, _native.raw_stream(dev)))
        torch.cuda.synchronize()
        lib.ls_direct_destroy(h)
        assert float((x - tv).abs().max()) <= 1e-4
        return o.value, w.value

    opt = _native.DirectOptions()
    _native.check(lib.ls_direct_options_default(ctypes.byref(opt)))
    assert opt.struct_bytes == ctypes.sizeof(_native.DirectOptions) and opt.tier_levels == -1 and opt.sparse_leaves == 1 and opt.ordering == -1
    auto = factor(None)
    opt.ordering = 0
    longest = factor(opt)
    opt.ordering = 1
    trial = factor(opt)
    assert longest[0] == 0 and trial[0] == 1 and trial[1] < longest[1]           # a scroll: the trial cuts find much thinner separators
    assert auto == trial                                                             # ... and the automatic rule picks them
    os.environ["LS_ND_ORDER"] = "0"
    try:
        assert factor(opt) == trial                                                  # explicit argument wins over the environment
        opt.ordering = -1
        assert factor(opt) == longest                                                # AUTO: the environment overrides the rule
    finally:
        del os.environ["LS_ND_ORDER"]
    opt.ordering = 7
    with pytest.raises(ValueError, match="ordering"):
        factor(opt)
    opt.ordering = -1
    opt.struct_bytes = 0
    with pytest.raises(ValueError, match="struct_bytes"):
        factor(opt)


def test_trial_cuts_are_chosen_after_a_long_served_period(dev):
    """A remesh loop tells the library its own period (round 6): when the previous solver of the same size class served at least
    NestedDissectionSolver.AUTO_TRIAL_CUTS_AFTER solves, the next construction with ordering=None asks for the trial cuts (their longer
    constructor is earned back by then); a short period, an explicit ordering or another size class keep the library's plain rule."""
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import to_differential
    from largesteps.solvers import NestedDissectionSolver
    from largesteps import synthetic
    v, f = synthetic.icosphere(100)
    v = synthetic.perturb(v, radial=0.05, tangential=0.25, edge=1.2 / 100, seed=0)
    tv, tf = _t(v, dev), _t(f, dev)
    M = compute_matrix(tv, tf, 0.0, alpha=0.95, cotan=True)
    u = to_differential(M, tv)
    NestedDissectionSolver._served.clear()
    first = NestedDissectionSolver(M)
    assert first.ordering_used is None and first.plan_quality["ordering"] == "longest-axis"
    x0 = first.solve(u)
    first.close()                                        # one solve served: a short period
    second = NestedDissectionSolver(M)
    assert second.ordering_used is None
    second.solves_served = NestedDissectionSolver.AUTO_TRIAL_CUTS_AFTER      # ... as if a long period had been served
    second.close()
    third = NestedDissectionSolver(M)
    assert third.ordering_used == "trial-cuts" and third.plan_quality["ordering"] == "trial-cuts"
    assert third.plan_quality["words_per_vertex"] < first.plan_quality["words_per_vertex"]
    x3 = third.solve(u)
    assert float((x3 - tv).abs().max()) <= 1e-4 and float((x3 - x0).abs().max()) <= 2e-5
    third.solves_served = 10 ** 6
    third.close()
    explicit = NestedDissectionSolver(M, ordering="longest-axis")           # an explicit choice is the caller's
    assert explicit.plan_quality["ordering"] == "longest-axis"
    explicit.close()
    NestedDissectionSolver._served.clear()


def test_a_remesh_of_a_folded_surface_skips_the_plan_that_was_rejected_last_time(dev):
    """The automatic rule builds the longest-axis plan, finds a folded sheet's separators suspect and builds the trial-cut plan as well
    (two plans + the graph distances). The next construction for the SAME surface (size class and bounding box) asks for the trial cuts at
    once -- one plan -- and gets the same dissection; a flat sheet of the same vertex count is another surface and keeps the plain rule.
    The graph distances behind the trial cuts are breadth-first sweeps on the device (nd_embed_device, round 6): LS_ND_HOST_EMBED=1 runs the
    host's sweeps instead and must give the same plan."""
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import to_differential
    from largesteps.solvers import NestedDissectionSolver
    from largesteps import synthetic
    v, f = synthetic.folded_sheet(300)
    tv, tf = _t(v, dev), _t(f, dev)
    M = compute_matrix(tv, tf, 19.0)
    u = to_differential(M, tv)
    first = NestedDissectionSolver(M)
    q1 = first.plan_quality
    assert first.ordering_used is None and q1["ordering"] == "trial-cuts" and q1["words_per_vertex_other"] > q1["words_per_vertex"]
    x1 = first.solve(u)
    first.close()
    second = NestedDissectionSolver(M)
    q2 = second.plan_quality
    assert second.ordering_used == "trial-cuts" and q2["ordering"] == "trial-cuts" and q2["words_per_vertex_other"] == 0.0
    assert q2["words_per_vertex"] == q1["words_per_vertex"]
    assert torch.equal(second.solve(u), x1)
    second.close()
    os.environ["LS_ND_HOST_EMBED"] = "1"
    try:
        third = NestedDissectionSolver(M)
    finally:
        del os.environ["LS_ND_HOST_EMBED"]
    assert third.plan_quality["words_per_vertex"] == q1["words_per_vertex"] and torch.equal(third.solve(u), x1)
    third.close()
    vf, ff = synthetic.plane(300)
    Mf = compute_matrix(_t(vf, dev), _t(ff, dev), 19.0)
    flat = NestedDissectionSolver(Mf)
    assert flat.ordering_used is None and flat.plan_quality["ordering"] == "longest-axis"
