#!/usr/bin/env python3
"""
Golden fixtures for remove_duplicates: EXECUTES the reference's scripts/geometry.py:3-11
(`torch.unique(v, dim=0, return_inverse=True)`; `inverse[f.long()]`) on CPU tensors in the dev container.
        python tests/golden/make_golden_dedup.py
The reference cannot travel to the GPU box, hence the committed fixture tests/golden/reference_dedup.npz. Pinned behaviour:
unique rows come out in lexicographic order of (x, y, z) as VALUES (-0.0 == 0.0), the inverse map is int64.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "large-steps-pytorch_amd", "largesteps"))
import synthetic  # noqa: E402


def soup(v, f):
    """every face gets its own three vertices (what a remesher / an OBJ with split normals hands over)"""
    return v[f.reshape(-1)].copy(), np.arange(f.size, dtype=np.int64).reshape(-1, 3)


def cases():
    rng = np.random.default_rng(5)
    out = {}
    v, f = synthetic.icosphere(4)
    out["ico4_soup"] = soup(v, f)
    v, f = synthetic.plane(12)
    out["plane12_soup"] = soup(v, f)
    out["plane12_unique"] = (v, f)
    v, f = synthetic.icosphere(6)
    sv, sf = soup(synthetic.perturb(v, radial=0.05, seed=1), f)
    p = rng.permutation(sv.shape[0])                      # shuffled storage order
    inv = np.empty_like(p)
    inv[p] = np.arange(p.shape[0])
    out["ico6_soup_shuffled"] = (sv[p], inv[sf])
    z = np.array([[0.0, 1.0, -0.0], [-0.0, 1.0, 0.0], [0.0, -1.0, 2.0], [-1.0, 5.0, 5.0], [0.0, -1.0, 2.0], [-1.0, -5.0, 5.0], [3.0, 0.0, 0.0]], np.float32)
    out["signed_zero_negatives"] = (z, np.array([[0, 2, 3], [1, 4, 5], [6, 0, 1]], np.int64))
    out["int32_faces"] = (out["ico4_soup"][0], out["ico4_soup"][1].astype(np.int32))
    return out


def main():
    spec = importlib.util.spec_from_file_location("ref_scripts_geometry", "/root/reference/scripts/geometry.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {}
    for name, (v, f) in cases().items():
        uv, nf, inv = ref.remove_duplicates(torch.from_numpy(v.astype(np.float32)), torch.from_numpy(f))
        out[f"{name}/v"], out[f"{name}/f"] = v.astype(np.float32), f
        out[f"{name}/unique"], out[f"{name}/new_faces"], out[f"{name}/inverse"] = uv.numpy(), nf.numpy(), inv.numpy()
        assert nf.dtype == torch.int64 and inv.dtype == torch.int64
        print(name, v.shape[0], "->", uv.shape[0])
    np.savez_compressed(os.path.join(HERE, "reference_dedup.npz"), **out)


if __name__ == "__main__":
    main()
