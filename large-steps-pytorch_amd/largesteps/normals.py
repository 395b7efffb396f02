"""
Per-face and per-vertex normals on the MI355X, differentiable.

Drop-in for the two functions every optimisation step calls right after `from_differential`
(rgl-epfl/large-steps-pytorch scripts/main.py:178-179; definitions scripts/geometry.py:91-110 and :115-147): same names,
same argument order, same shapes ((3, F) face normals, (V, 3) vertex normals), same quirks -- the corner weights use
the Frobenius norm of the whole edge matrix (`d0 / torch.norm(d0)`, geometry.py:138-141), a degenerate face or an
unreferenced vertex yields NaN. The reference runs ~40 stock torch kernels per call and lets autograd replay them; here
forward and backward are a handful of hand-written HIP kernels each (csrc/normals.hip) behind two autograd Functions.
There is no CPU path.
"""
import ctypes
import weakref

import torch
from torch.autograd import Function

from . import _native


# Per face TENSOR OBJECT: range check of the indices (syncs the host once) and the vertex-major ranking of the corners.
# One-off integer plumbing per mesh connectivity (ls_corner_ranks: native radix sort): the 3 F corners grouped by
# vertex, in ascending corner id (deterministic sums); cpos is the inverse permutation (corner -> rank), vptr the ranks
# each vertex owns. An entry is valid only for the very tensor it was built from (weak reference + version counter):
# a storage address is recycled by the caching allocator as soon as a face tensor dies, so it cannot be the identity.
_plans = {}


def _plan(f, V):
    key = id(f)
    hit = _plans.get(key)
    if hit is not None:
        ref, version, shape, dtype, ptr, v_count, plan = hit
        if ref() is f and version == f._version and shape == tuple(f.shape) and dtype == f.dtype and ptr == f.data_ptr() \
                and v_count == V:
            return plan
        del _plans[key]
    F = f.shape[0]
    n = ctypes.c_size_t(0)
    _native.check(_native.lib().ls_corner_ranks_workspace_bytes(F, V, ctypes.byref(n)))
    ws = torch.empty(n.value, dtype=torch.uint8, device=f.device)
    vcorner = torch.empty(max(3 * F, 1), dtype=torch.int32, device=f.device)[: 3 * F]     # corner -> rank ("cpos" of the C ABI)
    vptr = torch.empty(V + 1, dtype=torch.int32, device=f.device)
    with torch.cuda.device(f.device):                  # range check + counting + stable radix sort by vertex id, all native
        _native.check(_native.lib().ls_corner_ranks(_native.ptr(f), f.element_size(), F, V, _native.ptr(vptr), _native.ptr(vcorner), _native.ptr(ws),
                                                    ws.numel(), f.device.index, _native.stream_of(f.device)))
    for k in [k for k, h in _plans.items() if h[0]() is None]:      # entries of dead tensors
        del _plans[k]
    if len(_plans) >= 8:
        _plans.clear()
    # the kernels read the connectivity six times per step: 4-byte indices (validated above) halve that traffic
    narrow = f.to(torch.int32) if (f.dtype == torch.int64 and V < 2 ** 31) else f
    plan = (vptr, vcorner, narrow)
    try:
        _plans[key] = (weakref.ref(f), f._version, tuple(f.shape), f.dtype, f.data_ptr(), V, plan)
    except TypeError:           # an object that cannot be weakly referenced: do not cache
        pass
    return plan


def _prep(verts, faces):
    _native.require_device(verts, "verts")
    _native.require_device(faces, "faces")
    if verts.dim() != 2 or verts.shape[1] != 3:
        raise ValueError(f"verts must be (V, 3), got {tuple(verts.shape)}")
    if faces.dim() != 2 or faces.shape[1] != 3:
        raise ValueError(f"faces must be (F, 3), got {tuple(faces.shape)}")
    if faces.dtype not in (torch.int32, torch.int64):
        raise TypeError(f"faces must be int32 or int64, got {faces.dtype}")
    v = verts.detach()
    if v.dtype != torch.float32 or not v.is_contiguous():
        v = v.to(torch.float32).contiguous()
    f = faces if faces.is_contiguous() else faces.contiguous()
    vptr, vcorner, narrow = _plan(f, v.shape[0])
    return v, narrow, vptr, vcorner


def _workspace(F, V, dev):
    n = ctypes.c_size_t(0)
    _native.check(_native.lib().ls_normals_workspace_bytes(F, V, ctypes.byref(n)))
    return torch.empty(n.value, dtype=torch.uint8, device=dev)


class _FaceNormals(Function):
    @staticmethod
    def forward(ctx, verts, faces):
        v, f, vptr, vcorner = _prep(verts, faces)
        F, V, dev = f.shape[0], v.shape[0], v.device
        fn = torch.empty((3, F), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _native.check(_native.lib().ls_face_normals(_native.ptr(v), _native.ptr(f), f.element_size(), F, V, _native.ptr(fn),
                                                        dev.index, _native.stream_of(dev)))
        ctx.save_for_backward(v, f, vptr, vcorner)
        return fn

    @staticmethod
    def backward(ctx, g):
        v, f, vptr, vcorner = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None
        g = g.contiguous().to(torch.float32)
        dev = v.device
        gv = torch.empty_like(v)
        ws = _workspace(f.shape[0], v.shape[0], dev)
        with torch.cuda.device(dev):
            _native.check(_native.lib().ls_face_normals_backward(_native.ptr(v), _native.ptr(f), f.element_size(), f.shape[0], v.shape[0],
                                                                 _native.ptr(vptr), _native.ptr(vcorner), _native.ptr(g), _native.ptr(gv),
                                                                 _native.ptr(ws), ws.numel(), dev.index, _native.stream_of(dev)))
        return gv, None


class _VertexNormals(Function):
    @staticmethod
    def forward(ctx, verts, faces, face_normals):
        v, f, vptr, vcorner = _prep(verts, faces)
        F, V, dev = f.shape[0], v.shape[0], v.device
        _native.require_device(face_normals, "face_normals")
        if tuple(face_normals.shape) != (3, F):
            raise ValueError(f"face_normals must be (3, {F}), got {tuple(face_normals.shape)}")
        fn = face_normals.detach().to(torch.float32).contiguous()
        out = torch.empty((V, 3), dtype=torch.float32, device=dev)
        raw = torch.empty((V, 3), dtype=torch.float32, device=dev)
        norms = torch.empty(3, dtype=torch.float32, device=dev)
        ws = _workspace(F, V, dev)
        with torch.cuda.device(dev):
            _native.check(_native.lib().ls_vertex_normals(_native.ptr(v), _native.ptr(f), f.element_size(), F, V, _native.ptr(vptr),
                                                          _native.ptr(vcorner), _native.ptr(fn), _native.ptr(out), _native.ptr(raw),
                                                          _native.ptr(norms), _native.ptr(ws), ws.numel(), dev.index, _native.stream_of(dev)))
        ctx.save_for_backward(v, f, fn, raw, norms, vptr, vcorner)
        return out

    @staticmethod
    def backward(ctx, g):
        v, f, fn, raw, norms, vptr, vcorner = ctx.saved_tensors
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[2]):
            return None, None, None
        g = g.contiguous().to(torch.float32)
        F, V, dev = f.shape[0], v.shape[0], v.device
        gv = torch.empty_like(v)
        gfn = torch.empty_like(fn)
        ws = _workspace(F, V, dev)
        with torch.cuda.device(dev):
            _native.check(_native.lib().ls_vertex_normals_backward(_native.ptr(v), _native.ptr(f), f.element_size(), F, V, _native.ptr(vptr),
                                                                   _native.ptr(vcorner), _native.ptr(fn),
                                                                   _native.ptr(raw), _native.ptr(norms), _native.ptr(g), _native.ptr(gv),
                                                                   _native.ptr(gfn), _native.ptr(ws), ws.numel(), dev.index,
                                                                   _native.stream_of(dev)))
        return (gv if ctx.needs_input_grad[0] else None), None, (gfn if ctx.needs_input_grad[2] else None)


def compute_face_normals(verts, faces):
    """
    Compute per-face normals (scripts/geometry.py:91-110). Returns a (3, F) tensor.

    Parameters
    ----------
    verts : torch.Tensor
        Vertex positions (V, 3), fp32, on a HIP device
    faces : torch.Tensor
        Triangle faces (F, 3), int32 or int64
    """
    return _FaceNormals.apply(verts, faces)


def compute_vertex_normals(verts, faces, face_normals):
    """
    Compute per-vertex normals from face normals (scripts/geometry.py:115-147). Returns a (V, 3) tensor.

    Parameters
    ----------
    verts : torch.Tensor
        Vertex positions (V, 3)
    faces : torch.Tensor
        Triangle faces (F, 3)
    face_normals : torch.Tensor
        Per-face normals (3, F), normally the output of `compute_face_normals`
    """
    return _VertexNormals.apply(verts, faces, face_normals)
