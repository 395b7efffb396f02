#!/usr/bin/env python3
"""
Golden fixture for the whole optimisation step (SURVEY.md §8 rows a5-a10, f3 chained the way the reference chains them):
EXECUTES the reference's loop body -- scripts/main.py:172-208 without the renderer, which is out of scope (nvdiffrast) --
    v = from_differential(M, u, 'Cholesky')                      largesteps/parameterize.py:32-61
    n = compute_vertex_normals(v, f, compute_face_normals(v, f)) scripts/geometry.py:91-147
    loss = (v - v_target)^2.mean() + (n - n_target)^2.mean() + reg * (L @ v)^2.mean()      (image loss replaced by an L2 loss on
                                                                  positions and normals; the regulariser is main.py:193)
    loss.backward(); AdamUniform.step()                           largesteps/optimize.py:18-41
for 5 steps on two small meshes (uniform lambda-form and cotangent alpha-form), and records u, v and the loss after every step.
        python tests/golden/make_golden_step.py      (dev container; the reference cannot travel to the GPU box)

'Cholesky' runs through the scipy stand-in for cholespy of make_golden.py (fp64 SuperLU): to fp32 round-off the same x any
correct Cholesky returns; everything else is the reference's own code.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402
from make_golden import synthetic  # noqa: E402


def main():
    geometry, solvers, parameterize, optimize = make_golden.load_reference()
    spec = importlib.util.spec_from_file_location("ref_scripts_geometry", "/root/reference/scripts/geometry.py")
    sg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sg)
    torch.manual_seed(0)
    out = {}
    cases = {
        "ico5_uni": (synthetic.icosphere(5), dict(lambda_=8.0, alpha=None, cotan=False)),
        "ico6_cot": ((synthetic.perturb(synthetic.icosphere(6)[0], radial=0.03, tangential=0.1, edge=0.1, seed=5), synthetic.icosphere(6)[1]),
                     dict(lambda_=0.0, alpha=0.9, cotan=True)),
    }
    for name, ((v, f), par) in cases.items():
        tv = torch.from_numpy(v.astype(np.float32))
        tf = torch.from_numpy(f.astype(np.int64))
        M = geometry.compute_matrix(tv, tf, par["lambda_"], alpha=par["alpha"], cotan=par["cotan"])
        L = geometry.laplacian_uniform(tv, tf)
        # target: the mesh inflated and sheared a little; its normals
        target_v = tv * torch.tensor([1.08, 0.95, 1.02]) + 0.05 * tv[:, [1, 2, 0]]
        target_n = sg.compute_vertex_normals(target_v, tf, sg.compute_face_normals(target_v, tf)).detach()
        u = parameterize.to_differential(M, tv).clone().requires_grad_(True)
        opt = optimize.AdamUniform([u], 1e-2)
        reg = 1e-3
        us, vs, losses = [], [], []
        for _ in range(5):
            x = parameterize.from_differential(M, u, 'Cholesky')
            n = sg.compute_vertex_normals(x, tf, sg.compute_face_normals(x, tf))
            loss = (x - target_v).square().mean() + (n - target_n).square().mean() + reg * (L @ x).square().mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
            us.append(u.detach().numpy().copy()); vs.append(x.detach().numpy().copy()); losses.append(float(loss))
        out.update({f"{name}/verts": v.astype(np.float32), f"{name}/faces": f.astype(np.int64), f"{name}/target_v": target_v.numpy(),
                    f"{name}/target_n": target_n.numpy(), f"{name}/lambda": np.float64(par["lambda_"]),
                    f"{name}/alpha": np.float64(-1.0 if par["alpha"] is None else par["alpha"]), f"{name}/cotan": np.int64(par["cotan"]),
                    f"{name}/u_steps": np.stack(us), f"{name}/v_steps": np.stack(vs), f"{name}/losses": np.array(losses, np.float64),
                    f"{name}/lr": np.float64(1e-2), f"{name}/reg": np.float64(reg)})
        print(name, "V", v.shape[0], "losses", ["%.6f" % x for x in losses])
    np.savez_compressed(os.path.join(HERE, "reference_step.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
