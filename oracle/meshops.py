"""
oracle.meshops -- CPU restatement of the reference's scripts/geometry.py:3-11 (TEST INFRASTRUCTURE, see oracle/__init__.py).
"""
import numpy as np


def remove_duplicates(v, f):
    """scripts/geometry.py:3-11: `unique_verts, inverse = torch.unique(v, dim=0, return_inverse=True)`; `new_faces =
    inverse[f.long()]`. torch.unique(dim=0) returns the distinct rows in lexicographic order of their values (-0.0 == 0.0),
    which is np.unique(axis=0)."""
    uv, inv = np.unique(np.asarray(v), axis=0, return_inverse=True)
    inv = inv.reshape(-1).astype(np.int64)
    return uv, inv[np.asarray(f).astype(np.int64)], inv
