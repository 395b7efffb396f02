"""
Many independent small meshes through ONE set of launches.

The reference handles one mesh per solver (largesteps/parameterize.py:32-61); a loop over B small meshes pays B times the
launch chain of the re-solve, which at <= 100k vertices is pure latency (~0.1 ms per mesh on the MI355X whatever its size).
Here the B systems are one block-diagonal system: the meshes are concatenated (vertex ids offset), `compute_matrix`
assembles the union in one pass, and the direct solver's nested dissection is fed positions that lay the meshes SIDE BY
SIDE -- so its first bisections fall between meshes, where no edge is cut: those tree nodes are empty and cost nothing, and
below them every mesh is dissected as it would be alone. Factorisation, every tree level of the re-solve, the normals and
the optimizer then run once for all meshes.
"""
import torch

from . import _native
from .geometry import compute_matrix


class MeshBatch:
    """Concatenation of B meshes. verts (sum V_i, 3), faces (sum F_i, 3) with offset vertex ids, vertex_ptr / face_ptr (B + 1)."""

    def __init__(self, verts_list, faces_list):
        if len(verts_list) != len(faces_list) or not verts_list:
            raise ValueError("MeshBatch needs the same non-zero number of vertex and face tensors")
        dev = verts_list[0].device
        vp, fp = [0], [0]
        for v, f in zip(verts_list, faces_list):
            _native.require_device(v, "verts")
            _native.require_device(f, "faces")
            if v.dim() != 2 or v.shape[1] != 3 or f.dim() != 2 or f.shape[1] != 3:
                raise ValueError("every mesh needs (V, 3) vertices and (F, 3) faces")
            if v.device != dev or f.device != dev:
                raise RuntimeError("all meshes of a batch must live on the same device")
            vp.append(vp[-1] + v.shape[0])
            fp.append(fp[-1] + f.shape[0])
        self.vertex_ptr, self.face_ptr = vp, fp
        self.verts = torch.cat([v.detach().to(torch.float32) for v in verts_list], 0).contiguous()
        self.faces = torch.cat([f.long() + o for f, o in zip(faces_list, vp[:-1])], 0).contiguous()
        # layout for the dissection: mesh i shifted along x by the sum of the extents before it (+ 25 % gaps)
        lo = torch.stack([v.detach().min(0).values for v in verts_list]).to(torch.float32)
        hi = torch.stack([v.detach().max(0).values for v in verts_list]).to(torch.float32)
        width = (hi[:, 0] - lo[:, 0]).clamp_min(1e-6) * 1.25
        start = torch.cumsum(width, 0) - width
        shift = torch.zeros((len(verts_list), 3), dtype=torch.float32, device=dev)
        shift[:, 0] = start - lo[:, 0]
        counts = torch.tensor([b - a for a, b in zip(vp[:-1], vp[1:])], device=dev)
        self.layout = (self.verts + torch.repeat_interleave(shift, counts, dim=0)).contiguous()

    def __len__(self):
        return len(self.vertex_ptr) - 1

    def split(self, x):
        """per-mesh views of a per-vertex tensor of the batch"""
        return [x[a:b] for a, b in zip(self.vertex_ptr[:-1], self.vertex_ptr[1:])]


def compute_matrix_batched(batch, lambda_, alpha=None, cotan=False):
    """`compute_matrix` (geometry.py:96-133) for all meshes of the batch at once: the block-diagonal system matrix of the union
    mesh. `to_differential` / `from_differential` / the solvers take it like any other matrix, on (sum V_i, k) tensors."""
    M = compute_matrix(batch.verts, batch.faces, lambda_, alpha=alpha, cotan=cotan)
    _native.csr_of(M).positions = batch.layout        # the dissection sees the meshes side by side
    return M
