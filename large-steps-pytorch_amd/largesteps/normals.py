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
import os
import threading
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
    # rank -> corner, the inverse of vcorner: what the vertex-major passes of the pair walk (one-off integer plumbing, like `narrow`)
    order = torch.empty_like(vcorner)
    order[vcorner.long()] = torch.arange(3 * F, dtype=torch.int32, device=f.device)
    plan = (vptr, vcorner, narrow, order)
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
    vptr, vcorner, narrow, order = _plan(f, v.shape[0])
    return v, narrow, vptr, vcorner, order


_ws_bytes = {}


def _workspace(F, V, dev):
    n = _ws_bytes.get((F, V))
    if n is None:
        c = ctypes.c_size_t(0)
        _native.check(_native.lib().ls_normals_workspace_bytes(F, V, ctypes.byref(c)))
        if len(_ws_bytes) > 64:
            _ws_bytes.clear()
        n = _ws_bytes[(F, V)] = c.value
    return torch.empty(n, dtype=torch.uint8, device=dev)


class _Here:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_here = _Here()


def _on(dev):
    """torch's device context only when `dev` is not the current device already (the library selects the device it is handed
    itself; entering torch's context costs several microseconds per call on the host, and an eager step is host-bound)"""
    return _here if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


# ---- the pair on one mesh ------------------------------------------------------------------------------------------------
# scripts/main.py:178-179 calls compute_face_normals and hands its result straight to compute_vertex_normals: the two are then
# one function of the vertices, and csrc/normals.hip has passes for exactly that case (one reduction less in the forward, the
# corner buffer written and gathered once in the backward instead of twice). The reference's two-function signature stays; the
# face-normal tensor carries a tag that says which vertex / face tensors (objects and versions) it was computed from, and
# compute_vertex_normals takes the shared passes only when it is handed that very tensor together with the same vertices and
# faces -- a detached copy, a view, face normals from elsewhere or an in-place edit take the general path (face normals as an
# independent input), exactly as before.
#
# Backward of the pair: autograd runs the vertex-normal node first. It returns the gradient that reaches the face normals
# (other consumers of them may add theirs) and NO gradient for the vertices: that part is handed, through the tag, to the
# face-normal node, whose single pass over the faces produces the whole vertex gradient. The hand-over is labelled with the
# id of the running backward call (torch._C._current_graph_task_id): the face-normal node only takes what was left for it
# in the same call, so a backward that stops at the face normals cannot leak into a later one.
class _PairTag:
    __slots__ = ("verts", "verts_version", "faces", "faces_version", "fn", "fn_version", "norms", "pending", "node", "__weakref__")

    def __init__(self, verts, faces, norms):
        self.verts, self.verts_version = weakref.ref(verts), verts._version
        self.faces, self.faces_version = weakref.ref(faces), faces._version
        self.fn, self.fn_version = None, 0
        self.node = None               # weak reference to the face-normal autograd node that produced the tagged tensor (None: no graph)
        self.norms = norms
        self.pending = {}              # id(ctx of a vertex-normal node) -> (graph task, g_raw, gN)

    def matches(self, verts, faces, fn):
        return (self.verts() is verts and self.verts_version == verts._version and self.faces() is faces
                and self.faces_version == faces._version and self.fn is not None and self.fn() is fn and self.fn_version == fn._version)

    def produced_by_live_node(self, fn):
        """True iff `fn` still hangs on the face-normal node that made it: only then will that node run in a backward pass and finish
        what the vertex-normal node hands to it. Face normals computed under no_grad() and switched to requires_grad afterwards (or
        detached in place) match by identity and version and have NO such node -- the hand-over would be dropped silently."""
        node = self.node() if self.node is not None else None
        return node is not None and fn.grad_fn is node


_graph_task_id = getattr(torch._C, "_current_graph_task_id", None)
_handoff = threading.local()   # .tag: the tag of the face-normal node this thread's forward just ran (picked up by compute_face_normals)


class _FaceNormals(Function):
    @staticmethod
    def forward(ctx, verts, faces):
        v, f, vptr, vcorner, _ = _prep(verts, faces)
        F, V, dev = f.shape[0], v.shape[0], v.device
        fn = torch.empty((3, F), dtype=torch.float32, device=dev)
        norms = torch.empty(3, dtype=torch.float32, device=dev)
        ws = _workspace(F, V, dev)
        with _on(dev):
            _native.check(_native.lib().ls_face_normals_with_norms(_native.ptr(v), _native.ptr(f), f.element_size(), F, V, _native.ptr(fn),
                                                                   _native.ptr(norms), _native.ptr(ws), ws.numel(), dev.index,
                                                                   _native.stream_of(dev)))
        ctx.save_for_backward(v, f, vptr, vcorner)
        ctx.set_materialize_grads(False)
        ctx.tag = _PairTag(verts, faces, norms)
        _handoff.tag = ctx.tag
        return fn

    @staticmethod
    def backward(ctx, g):
        v, f, vptr, vcorner = ctx.saved_tensors
        task = _graph_task_id() if _graph_task_id is not None else -1
        mine = [e for e in ctx.tag.pending.values() if e[0] == task and task >= 0]
        ctx.tag.pending.clear()
        if not ctx.needs_input_grad[0] or (g is None and not mine):
            return None, None
        if g is not None:
            g = g.contiguous().to(torch.float32)
        F, V, dev = f.shape[0], v.shape[0], v.device
        ws = _workspace(F, V, dev)
        lib = _native.lib()
        with _on(dev):
            if not mine:
                gv = torch.empty_like(v)
                _native.check(lib.ls_face_normals_backward(_native.ptr(v), _native.ptr(f), f.element_size(), F, V, _native.ptr(vptr),
                                                           _native.ptr(vcorner), _native.ptr(g), _native.ptr(gv), _native.ptr(ws),
                                                           ws.numel(), dev.index, _native.stream_of(dev)))
                return gv, None
            total = None
            for k, (_, g_raw, gN) in enumerate(mine):          # one entry unless several vertex-normal nodes share these face normals
                gv = torch.empty_like(v)
                _native.check(lib.ls_normals_pair_backward_verts(_native.ptr(v), _native.ptr(f), f.element_size(), F, V, _native.ptr(vptr),
                                                                 _native.ptr(vcorner), _native.ptr(ctx.tag.norms), _native.ptr(g_raw),
                                                                 _native.ptr(gN), _native.ptr(g if k == 0 else None), _native.ptr(gv),
                                                                 _native.ptr(ws), ws.numel(), dev.index, _native.stream_of(dev)))
                total = gv if total is None else total + gv
        return total, None


class _VertexNormals(Function):
    @staticmethod
    def forward(ctx, verts, faces, face_normals, tag, can_defer=False):
        ctx.can_defer = bool(can_defer)
        v, f, vptr, vcorner, order = _prep(verts, faces)
        F, V, dev = f.shape[0], v.shape[0], v.device
        _native.require_device(face_normals, "face_normals")
        if tuple(face_normals.shape) != (3, F):
            raise ValueError(f"face_normals must be (3, {F}), got {tuple(face_normals.shape)}")
        out = torch.empty((V, 3), dtype=torch.float32, device=dev)
        raw = torch.empty((V, 3), dtype=torch.float32, device=dev)
        ws = _workspace(F, V, dev)
        lib = _native.lib()
        ctx.tag = tag
        if tag is not None:                                     # the pair: norms are there, n_f is recomputed
            with _on(dev):
                # vertex-major: each vertex recomputes the contributions of its corners in rank order -- the bits of
                # ls_vertex_normals_from_norms without its corner buffer (tests/test_gpu_parity.py compares the two)
                _native.check(lib.ls_vertex_normals_gathered(_native.ptr(v), _native.ptr(f), f.element_size(), F, V, _native.ptr(vptr),
                                                             _native.ptr(order), _native.ptr(tag.norms), _native.ptr(out), _native.ptr(raw),
                                                             dev.index, _native.stream_of(dev)))
            ctx.save_for_backward(v, f, raw, vptr, vcorner)
            return out
        fn = face_normals.detach().to(torch.float32).contiguous()
        norms = torch.empty(3, dtype=torch.float32, device=dev)
        with _on(dev):
            _native.check(lib.ls_vertex_normals(_native.ptr(v), _native.ptr(f), f.element_size(), F, V, _native.ptr(vptr),
                                                _native.ptr(vcorner), _native.ptr(fn), _native.ptr(out), _native.ptr(raw),
                                                _native.ptr(norms), _native.ptr(ws), ws.numel(), dev.index, _native.stream_of(dev)))
        ctx.save_for_backward(v, f, fn, raw, norms, vptr, vcorner)
        return out

    @staticmethod
    def backward(ctx, g):
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[2]):
            return None, None, None, None, None
        g = g.contiguous().to(torch.float32)
        lib = _native.lib()
        tag = ctx.tag
        if tag is not None:
            v, f, raw, vptr, vcorner = ctx.saved_tensors
            F, V, dev = f.shape[0], v.shape[0], v.device
            ws = _workspace(F, V, dev)
            g_raw = torch.empty_like(v)
            gN = torch.empty(4, dtype=torch.float32, device=dev)
            gfn = torch.empty((3, F), dtype=torch.float32, device=dev)
            task = _graph_task_id() if _graph_task_id is not None else -1
            # the face-normal node runs later in this backward call exactly when its output needs a gradient here and the
            # vertices need one: then it finishes the job (one corner buffer). Otherwise both halves run now.
            # (and only if the face normals really hang on that node: ctx.can_defer, decided in compute_vertex_normals)
            defer = task >= 0 and ctx.needs_input_grad[0] and ctx.needs_input_grad[2] and ctx.can_defer
            with _on(dev):
                _native.check(lib.ls_normals_pair_backward_faces(_native.ptr(v), _native.ptr(f), f.element_size(), F, V, _native.ptr(raw),
                                                                 _native.ptr(tag.norms), _native.ptr(g), _native.ptr(g_raw), _native.ptr(gN),
                                                                 _native.ptr(gfn), _native.ptr(ws), ws.numel(), dev.index,
                                                                 _native.stream_of(dev)))
                if defer:
                    tag.pending[id(ctx)] = (task, g_raw, gN)
                    return None, None, gfn, None, None
                gv = None
                if ctx.needs_input_grad[0]:
                    gv = torch.empty_like(v)
                    _native.check(lib.ls_normals_pair_backward_verts(_native.ptr(v), _native.ptr(f), f.element_size(), F, V, _native.ptr(vptr),
                                                                     _native.ptr(vcorner), _native.ptr(tag.norms), _native.ptr(g_raw),
                                                                     _native.ptr(gN), _native.ptr(None), _native.ptr(gv), _native.ptr(ws),
                                                                     ws.numel(), dev.index, _native.stream_of(dev)))
            return gv, None, (gfn if ctx.needs_input_grad[2] else None), None, None
        v, f, fn, raw, norms, vptr, vcorner = ctx.saved_tensors
        F, V, dev = f.shape[0], v.shape[0], v.device
        gv = torch.empty_like(v)
        gfn = torch.empty_like(fn)
        ws = _workspace(F, V, dev)
        with _on(dev):
            _native.check(lib.ls_vertex_normals_backward(_native.ptr(v), _native.ptr(f), f.element_size(), F, V, _native.ptr(vptr),
                                                         _native.ptr(vcorner), _native.ptr(fn),
                                                         _native.ptr(raw), _native.ptr(norms), _native.ptr(g), _native.ptr(gv),
                                                         _native.ptr(gfn), _native.ptr(ws), ws.numel(), dev.index,
                                                         _native.stream_of(dev)))
        return (gv if ctx.needs_input_grad[0] else None), None, (gfn if ctx.needs_input_grad[2] else None), None, None


@_native.retry_on_oom
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
    fn = _FaceNormals.apply(verts, faces)
    tag, _handoff.tag = getattr(_handoff, "tag", None), None
    if tag is not None:                            # tag the result: compute_vertex_normals recognises the pair by it
        tag.fn, tag.fn_version = weakref.ref(fn), fn._version
        tag.node = weakref.ref(fn.grad_fn) if fn.grad_fn is not None else None
        fn._largesteps_pair = tag
    return fn


@_native.retry_on_oom
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
    tag = getattr(face_normals, "_largesteps_pair", None)
    if tag is not None and not (tag.matches(verts, faces, face_normals) and os.environ.get("LARGESTEPS_NORMALS_PAIR", "1") != "0"):
        tag = None
    return _VertexNormals.apply(verts, faces, face_normals, tag, tag is not None and tag.produced_by_live_node(face_normals))
