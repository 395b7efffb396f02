"""
Vertex / face normals: oracle vs the reference-generated fixture (CPU), HIP kernels vs
oracle and fixture (-m gpu). Fixture: tests/golden/reference_normals.npz = outputs AND torch-autograd gradients of the
reference's own scripts/geometry.py (tests/golden/make_golden_normals.py).
"""
import os

import numpy as np
import pytest
import torch

from oracle import normals as on

HERE = os.path.dirname(os.path.abspath(__file__))
MESHES = ["tetra", "quad", "ico3", "ico8_noisy", "plane9", "unreferenced"]


@pytest.fixture(scope="module")
def ref():
    return np.load(os.path.join(HERE, "golden", "reference_normals.npz"))


def close(a, b, atol):
    ok = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), ok), "NaN pattern differs from the reference"
    assert np.abs(a[ok] - b[ok]).max(initial=0.0) <= atol


@pytest.mark.parametrize("name", MESHES)
def test_oracle_vs_reference(ref, name):
    v, f = ref[f"{name}/verts"], ref[f"{name}/faces"]
    fn = on.face_normals(v, f)
    close(fn, ref[f"{name}/face_normals"], 3e-7)
    close(on.vertex_normals(v, f, fn), ref[f"{name}/vertex_normals"], 3e-7)
    gscale = max(np.abs(ref[f"{name}/grad_all"]).max(), 1e-3)
    close(on.face_normals_backward(v, f, ref[f"{name}/w_f"]), ref[f"{name}/grad_face"], 1e-6 * max(np.abs(ref[f"{name}/grad_face"]).max(), 1.0))
    gv, gfn = on.vertex_normals_backward(v, f, ref[f"{name}/face_normals"].astype(np.float64), ref[f"{name}/w_v"])
    close(gv, ref[f"{name}/grad_vn_verts"], 2e-6 * gscale)
    close(gfn, ref[f"{name}/grad_vn_fn"], 1e-6 * max(np.abs(ref[f"{name}/grad_vn_fn"]).max(), 1.0))
    close(gv + on.face_normals_backward(v, f, gfn), ref[f"{name}/grad_all"], 2e-6 * gscale)


def test_oracle_gradients_are_derivatives():
    """central differences of the oracle's own forward, incl. the global-norm terms"""
    from largesteps import synthetic
    v, f = synthetic.icosphere(2)
    v = (v * (1.0 + 0.1 * np.random.default_rng(0).standard_normal((v.shape[0], 1)))).astype(np.float64)
    w = np.random.default_rng(1).standard_normal(v.shape)
    loss = lambda x: float((on.vertex_normals(x, f, on.face_normals(x, f)) * w).sum())    # noqa: E731
    fn = on.face_normals(v, f)
    gv, gfn = on.vertex_normals_backward(v, f, fn, w)
    g = gv + on.face_normals_backward(v, f, gfn)
    rng = np.random.default_rng(2)
    for _ in range(6):
        d = rng.standard_normal(v.shape)
        h = 1e-6
        fd = (loss(v + h * d) - loss(v - h * d)) / (2 * h)
        assert abs(fd - (g * d).sum()) <= 1e-6 * max(1.0, abs(fd))


# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from largesteps import _native
    _native.lib()
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("idx", [np.int64, np.int32])
@pytest.mark.parametrize("name", MESHES)
def test_hip_vs_reference(ref, dev, name, idx):
    from largesteps.normals import compute_face_normals, compute_vertex_normals
    v, f = ref[f"{name}/verts"], ref[f"{name}/faces"].astype(idx)
    tv = _t(v, dev).requires_grad_(True)
    tf = _t(f, dev)
    fn = compute_face_normals(tv, tf)
    vn = compute_vertex_normals(tv, tf, fn)
    assert fn.shape == (3, f.shape[0]) and vn.shape == v.shape and fn.dtype == torch.float32
    # fp32 kernels vs the reference's fp32 torch ops: a few ulp (different summation order in the scatter / norms)
    close(fn.detach().cpu().numpy(), ref[f"{name}/face_normals"], 1e-6)
    close(vn.detach().cpu().numpy(), ref[f"{name}/vertex_normals"], 2e-6)
    gscale = max(np.abs(ref[f"{name}/grad_all"]).max(), 1e-3)
    g_all, = torch.autograd.grad((vn * _t(ref[f"{name}/w_v"], dev)).sum(), tv, retain_graph=True)
    close(g_all.cpu().numpy(), ref[f"{name}/grad_all"], 2e-5 * gscale)
    g_face, = torch.autograd.grad((fn * _t(ref[f"{name}/w_f"], dev)).sum(), tv, retain_graph=True)
    close(g_face.cpu().numpy(), ref[f"{name}/grad_face"], 1e-5 * max(np.abs(ref[f"{name}/grad_face"]).max(), 1.0))
    # face normals as an independent input (the reference's signature allows it)
    fn_c = _t(ref[f"{name}/face_normals"], dev).requires_grad_(True)
    tv2 = _t(v, dev).requires_grad_(True)
    vn2 = compute_vertex_normals(tv2, tf, fn_c)
    gv, gfn = torch.autograd.grad((vn2 * _t(ref[f"{name}/w_v"], dev)).sum(), (tv2, fn_c))
    close(gv.cpu().numpy(), ref[f"{name}/grad_vn_verts"], 2e-5 * gscale)
    close(gfn.cpu().numpy(), ref[f"{name}/grad_vn_fn"], 1e-5 * max(np.abs(ref[f"{name}/grad_vn_fn"]).max(), 1.0))


@pytest.mark.gpu
@pytest.mark.parametrize("name", MESHES)
def test_hip_pair_paths_vs_reference(ref, dev, name, monkeypatch):
    """compute_face_normals -> compute_vertex_normals on one mesh share passes (one corner buffer in the backward, the
    vertex gradient finished by the face-normal node). Every way the two nodes can meet in a backward call must give the
    reference's gradients: both outputs used, a backward that stops at the face normals followed by one that does not, two
    vertex-normal nodes on one face-normal tensor, and the general path (LARGESTEPS_NORMALS_PAIR=0) next to the pair."""
    from largesteps.normals import compute_face_normals, compute_vertex_normals
    v, f = ref[f"{name}/verts"], ref[f"{name}/faces"]
    tv, tf = _t(v, dev).requires_grad_(True), _t(f, dev)
    w_v, w_f = _t(ref[f"{name}/w_v"], dev), _t(ref[f"{name}/w_f"], dev)
    g_all, g_face = ref[f"{name}/grad_all"], ref[f"{name}/grad_face"]
    tol = 2e-5 * max(np.abs(g_all).max(), 1e-3) + 1e-5 * max(np.abs(g_face).max(), 1.0)
    fn = compute_face_normals(tv, tf)
    vn = compute_vertex_normals(tv, tf, fn)
    assert getattr(fn, "_largesteps_pair", None) is not None
    both, = torch.autograd.grad((vn * w_v).sum() + (fn * w_f).sum(), tv, retain_graph=True)
    close(both.cpu().numpy(), g_all + g_face, tol)
    # a backward that ends at the face normals leaves nothing behind for the next one
    g_fn, = torch.autograd.grad((vn * w_v).sum(), fn, retain_graph=True)
    close(g_fn.cpu().numpy(), ref[f"{name}/grad_vn_fn"], 1e-5 * max(np.abs(ref[f"{name}/grad_vn_fn"]).max(), 1.0))
    only_face, = torch.autograd.grad((fn * w_f).sum(), tv, retain_graph=True)
    close(only_face.cpu().numpy(), g_face, 1e-5 * max(np.abs(g_face).max(), 1.0))
    # two vertex-normal nodes on the same face normals
    vn_b = compute_vertex_normals(tv, tf, fn)
    twice, = torch.autograd.grad((vn * w_v).sum() + (vn_b * w_v).sum(), tv, retain_graph=True)
    close(twice.cpu().numpy(), 2 * g_all, 2 * tol)
    # the pair and the general path agree (values: the same arithmetic; gradients: a different order of a few additions)
    g_pair, = torch.autograd.grad((vn * w_v).sum(), tv)
    monkeypatch.setenv("LARGESTEPS_NORMALS_PAIR", "0")
    fn2 = compute_face_normals(tv, tf)
    vn2 = compute_vertex_normals(tv, tf, fn2)
    assert torch.equal(fn2, fn)
    close(vn2.detach().cpu().numpy(), vn.detach().cpu().numpy(), 1e-6)
    g_gen, = torch.autograd.grad((vn2 * w_v).sum(), tv)
    close(g_gen.cpu().numpy(), g_all, tol)
    close(g_pair.cpu().numpy(), g_all, tol)
    # face normals computed without a graph: a constant for the vertex normals, in the pair's kernels as well
    monkeypatch.delenv("LARGESTEPS_NORMALS_PAIR")
    with torch.no_grad():
        fn3 = compute_face_normals(tv, tf)
    g_const, = torch.autograd.grad((compute_vertex_normals(tv, tf, fn3) * w_v).sum(), tv)
    close(g_const.cpu().numpy(), ref[f"{name}/grad_vn_verts"], 2e-5 * max(np.abs(g_all).max(), 1e-3))
    # ... and switched to requires_grad afterwards: the tensor still carries the tag (same object, same version) but NO face-normal node
    # will run in the backward -- the vertex-normal node must finish the vertices' gradient itself (advisor's finding, round 3: the
    # hand-over was dropped and the gradient came back None)
    fn3.requires_grad_(True)
    assert fn3.grad_fn is None and getattr(fn3, "_largesteps_pair", None) is not None
    g_v, g_f = torch.autograd.grad((compute_vertex_normals(tv, tf, fn3) * w_v).sum(), (tv, fn3))
    close(g_v.cpu().numpy(), ref[f"{name}/grad_vn_verts"], 2e-5 * max(np.abs(g_all).max(), 1e-3))
    close(g_f.cpu().numpy(), ref[f"{name}/grad_vn_fn"], 1e-5 * max(np.abs(ref[f"{name}/grad_vn_fn"]).max(), 1.0))


@pytest.mark.gpu
@pytest.mark.parametrize("idx", [np.int64, np.int32])
@pytest.mark.parametrize("name", MESHES + ["cfg2_bunny70k", "spike"])
def test_vertex_major_forward_equals_the_corner_buffer_forward(ref, dev, name, idx):
    """ls_vertex_normals_gathered (a thread per vertex recomputes the contributions of its corners in rank order; what the pair's forward
    runs) against ls_vertex_normals_from_norms (corner buffer written per face, summed per vertex), both through the C ABI on the same
    norms: the same bits in `raw` and `out`, NaN rows of unreferenced vertices included; a vertex of valence 40 walks its corners in
    several trips."""
    import ctypes
    from largesteps import _native, normals, synthetic
    if name == "spike":                                   # a fan of 40 triangles around vertex 0 + an unreferenced vertex
        k = 40
        ang = np.linspace(0, 2 * np.pi, k, endpoint=False)
        v = np.concatenate([[[0, 0, 0.3]], np.stack([np.cos(ang), np.sin(ang), 0.05 * np.cos(3 * ang)], 1), [[5, 5, 5]]]).astype(np.float32)
        f = np.stack([np.zeros(k, np.int64), 1 + np.arange(k), 1 + (np.arange(k) + 1) % k], 1)
    elif name.startswith("cfg"):
        v, f, _ = synthetic.config_mesh(name)
    else:
        v, f = ref[f"{name}/verts"], ref[f"{name}/faces"]
    tv, tf = _t(v.astype(np.float32), dev), _t(f.astype(idx), dev)
    vv, ff, vptr, vcorner, order = normals._prep(tv, tf)
    assert torch.equal(order[vcorner.long()].cpu(), torch.arange(3 * f.shape[0], dtype=torch.int32))
    F, V = ff.shape[0], vv.shape[0]
    lib = _native.lib()
    ws = normals._workspace(F, V, dev)
    fn = torch.empty((3, F), device=dev)
    norms = torch.empty(3, device=dev)
    _native.check(lib.ls_face_normals_with_norms(_native.ptr(vv), _native.ptr(ff), ff.element_size(), F, V, _native.ptr(fn), _native.ptr(norms),
                                                 _native.ptr(ws), ws.numel(), dev.index, _native.stream_of(dev)))
    out_a, raw_a, out_b, raw_b = (torch.empty_like(vv) for _ in range(4))
    _native.check(lib.ls_vertex_normals_from_norms(_native.ptr(vv), _native.ptr(ff), ff.element_size(), F, V, _native.ptr(vptr), _native.ptr(vcorner),
                                                   _native.ptr(norms), _native.ptr(out_a), _native.ptr(raw_a), _native.ptr(ws), ws.numel(),
                                                   dev.index, _native.stream_of(dev)))
    _native.check(lib.ls_vertex_normals_gathered(_native.ptr(vv), _native.ptr(ff), ff.element_size(), F, V, _native.ptr(vptr), _native.ptr(order),
                                                 _native.ptr(norms), _native.ptr(out_b), _native.ptr(raw_b), dev.index, _native.stream_of(dev)))
    torch.cuda.synchronize()
    for a, b in ((raw_a, raw_b), (out_a, out_b)):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert lib.ls_vertex_normals_gathered(_native.ptr(vv), _native.ptr(ff), ff.element_size(), F, V, _native.ptr(vptr), None, _native.ptr(norms),
                                          _native.ptr(out_b), _native.ptr(raw_b), dev.index, _native.stream_of(dev)) == _native.LS_E_INVALID


@pytest.mark.gpu
def test_hip_large_mesh_vs_oracle_and_errors(dev):
    from largesteps import synthetic
    from largesteps.normals import compute_face_normals, compute_vertex_normals
    v, f, _ = synthetic.config_mesh("cfg2_bunny70k")
    tv, tf = _t(v, dev).requires_grad_(True), _t(f, dev)
    fn = compute_face_normals(tv, tf)
    vn = compute_vertex_normals(tv, tf, fn)
    fn64 = on.face_normals(v, f)
    vn64 = on.vertex_normals(v, f, fn64)
    assert np.abs(fn.detach().cpu().numpy() - fn64).max() <= 2e-6
    assert np.abs(vn.detach().cpu().numpy() - vn64).max() <= 5e-6
    w = np.random.default_rng(0).standard_normal(v.shape).astype(np.float32)
    g, = torch.autograd.grad((vn * _t(w, dev)).sum(), tv)
    gv, gfn = on.vertex_normals_backward(v, f, fn64, w)
    g64 = gv + on.face_normals_backward(v, f, gfn)
    assert np.abs(g.cpu().numpy() - g64).max() <= 2e-4 * np.abs(g64).max()
    # no atomics anywhere: forward and backward are bitwise reproducible
    vn_b = compute_vertex_normals(tv, tf, compute_face_normals(tv, tf))
    g_b, = torch.autograd.grad((vn_b * _t(w, dev)).sum(), tv)
    assert torch.equal(vn_b, vn) and torch.equal(g_b, g)
    # size-independent property: unit length wherever a vertex is referenced
    assert float((vn.detach().norm(dim=1) - 1).abs().max()) <= 1e-5
    with pytest.raises(IndexError):
        compute_face_normals(tv, _t(np.array([[0, 1, v.shape[0]]]), dev))
    with pytest.raises(RuntimeError):
        compute_face_normals(torch.from_numpy(v), torch.from_numpy(f))
    with pytest.raises(ValueError):
        compute_vertex_normals(tv, tf, fn[:, :-1])
    with pytest.raises(TypeError):
        compute_face_normals(tv, tf.to(torch.int16))


@pytest.mark.gpu
def test_plan_cache_is_tied_to_the_face_tensor():
    """Two different connectivities of identical shape, the first freed before the second is created (the caching allocator
    hands the second the same address): the corner ranking must be rebuilt, not reused."""
    import gc
    import torch
    from largesteps import synthetic
    from largesteps.normals import compute_face_normals, compute_vertex_normals
    from oracle import normals as on
    dev = torch.device("cuda:0")
    v, f = synthetic.icosphere(6)
    rng = np.random.default_rng(0)
    tv = torch.from_numpy(v).to(dev)
    outs = []
    for trial in range(3):
        fp = f[rng.permutation(f.shape[0])][:, rng.permutation(3)] if trial else f
        # keep the orientation: a cyclic shift only
        fp = np.roll(f[rng.permutation(f.shape[0])], trial, axis=1)
        tf = torch.from_numpy(fp).to(dev)
        n = compute_vertex_normals(tv, tf, compute_face_normals(tv, tf)).cpu().numpy()
        ref = on.vertex_normals(v.astype(np.float64), fp, on.face_normals(v.astype(np.float64), fp))
        assert np.abs(n - ref).max() <= 1e-5
        outs.append(tf.data_ptr())
        del tf
        gc.collect()
