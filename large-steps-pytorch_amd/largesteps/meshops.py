"""
Mesh bookkeeping the optimisation loop needs around a remesh, on the MI355X.

`remove_duplicates` is the reference's scripts/geometry.py:3-11 with the same name, argument order and return values
(unique vertices in the order torch.unique(dim=0) gives them, re-indexed faces, inverse map) -- one native call
(`ls_remove_duplicates`: hand-written radix sort + compaction) instead of torch.unique's library sort. There is no CPU path.
"""
import ctypes

import torch

from . import _native


def remove_duplicates(v, f):
    """
    Generate a mesh representation with no duplicates and
    return it along with the mapping to the original mesh layout.

    Returns (unique_verts (U, 3), new_faces (F, 3) int64, inverse (V,) int64) with v == unique_verts[inverse].
    """
    _native.require_device(v, "v")
    _native.require_device(f, "f")
    if v.dim() != 2 or v.shape[1] != 3:
        raise ValueError(f"v must be (V, 3), got {tuple(v.shape)}")
    if f.dim() != 2 or f.shape[1] != 3:
        raise ValueError(f"f must be (F, 3), got {tuple(f.shape)}")
    if v.dtype != torch.float32:
        raise TypeError(f"v must be float32, got {v.dtype}")
    if f.dtype not in (torch.int32, torch.int64):
        raise TypeError(f"f must be int32 or int64, got {f.dtype}")
    if f.device != v.device:
        raise RuntimeError(f"v ({v.device}) and f ({f.device}) must be on the same device")
    vc, fc = v.detach().contiguous(), f.contiguous()
    V, F, dev = vc.shape[0], fc.shape[0], vc.device
    lib = _native.lib()
    n = ctypes.c_size_t(0)
    _native.check(lib.ls_remove_duplicates_workspace_bytes(V, ctypes.byref(n)))
    ws = torch.empty(n.value, dtype=torch.uint8, device=dev)
    unique = torch.empty((V, 3), dtype=torch.float32, device=dev)
    inverse = torch.empty(V, dtype=torch.int64, device=dev)
    new_faces = torch.empty((F, 3), dtype=torch.int64, device=dev)
    nu = ctypes.c_int64(0)
    with torch.cuda.device(dev):
        _native.check(lib.ls_remove_duplicates(_native.ptr(vc), V, _native.ptr(fc), fc.element_size(), F, _native.ptr(unique), _native.ptr(inverse),
                                               _native.ptr(new_faces), ctypes.byref(nu), _native.ptr(ws), ws.numel(), dev.index,
                                               _native.stream_of(dev)))
    return unique[: nu.value], new_faces, inverse

