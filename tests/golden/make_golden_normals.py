#!/usr/bin/env python3
"""
Golden fixtures for the vertex-normal row: EXECUTES the reference's scripts/geometry.py
(compute_face_normals :91-110, compute_vertex_normals :115-147) on CPU tensors in the dev container and records
outputs and autograd gradients.         python tests/golden/make_golden_normals.py

The file is pure torch (no device literals), so it is imported unmodified from /root/reference/scripts. The reference
cannot travel to the GPU box, hence the committed fixture tests/golden/reference_normals.npz.
Noteworthy reference behaviour that the fixture pins: `d0 / torch.norm(d0)` divides by the Frobenius norm of the WHOLE
(3, F) edge matrix, not per face -- the "angle" weights are acos(e_a . e_b / (||E_a||_F ||E_b||_F)).
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "large-steps-pytorch_amd", "largesteps"))
import synthetic  # noqa: E402


def main():
    spec = importlib.util.spec_from_file_location("ref_scripts_geometry", "/root/reference/scripts/geometry.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    meshes = {
        "tetra": (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32), np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]], np.int64)),
        "quad": (np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2], [0, 2, 3]], np.int64)),
        "ico3": synthetic.icosphere(3),
        "ico8_noisy": (synthetic.perturb(synthetic.icosphere(8)[0], radial=0.05, tangential=0.2, edge=0.15, seed=3), synthetic.icosphere(8)[1]),
        "plane9": synthetic.plane(9),
        "unreferenced": (np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [5, 5, 5]], np.float32), np.array([[0, 1, 2], [0, 2, 3]], np.int64)),
    }
    out = {}
    rng = np.random.default_rng(11)
    for name, (v, f) in meshes.items():
        tv = torch.from_numpy(v.astype(np.float32)).requires_grad_(True)
        tf = torch.from_numpy(f.astype(np.int64))
        fn = ref.compute_face_normals(tv, tf)                    # (3, F)
        vn = ref.compute_vertex_normals(tv, tf, fn)              # (V, 3)
        w_v = torch.from_numpy(rng.standard_normal(vn.shape).astype(np.float32))
        w_f = torch.from_numpy(rng.standard_normal(fn.shape).astype(np.float32))
        # gradient through both functions (what the optimisation loop back-propagates)
        g_all, = torch.autograd.grad((vn * w_v).sum() + 0.0 * fn.sum(), tv, retain_graph=True)
        # gradient of the face normals alone
        g_face, = torch.autograd.grad((fn * w_f).sum(), tv, retain_graph=True)
        # gradient of the vertex normals with the face normals treated as a constant input
        fn_c = fn.detach().requires_grad_(True)
        tv2 = tv.detach().clone().requires_grad_(True)
        vn2 = ref.compute_vertex_normals(tv2, tf, fn_c)
        g_v_only, g_fn = torch.autograd.grad((vn2 * w_v).sum(), (tv2, fn_c))
        out.update({f"{name}/verts": v.astype(np.float32), f"{name}/faces": f.astype(np.int64),
                    f"{name}/face_normals": fn.detach().numpy(), f"{name}/vertex_normals": vn.detach().numpy(),
                    f"{name}/w_v": w_v.numpy(), f"{name}/w_f": w_f.numpy(), f"{name}/grad_all": g_all.numpy(),
                    f"{name}/grad_face": g_face.numpy(), f"{name}/grad_vn_verts": g_v_only.numpy(), f"{name}/grad_vn_fn": g_fn.numpy()})
    np.savez_compressed(os.path.join(HERE, "reference_normals.npz"), **out)
    print("wrote", len(out), "arrays:", sorted({k.split('/')[0] for k in out}))


if __name__ == "__main__":
    main()
