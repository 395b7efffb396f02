#!/usr/bin/env python3
"""
Generate the golden fixtures of tests/golden/*.npz by EXECUTING the reference
(/root/reference/largesteps/{geometry,solvers,parameterize}.py) in the dev container.

Run from the repo root:   python tests/golden/make_golden.py

* geometry.py hard-codes device='cuda' (geometry.py:60,83,125,126). There is no GPU in the dev
  container, so the module source is read, the four literals are replaced by 'cpu' IN MEMORY and the
  result exec'd -- no reference source is written into this repo.
* solvers.py imports `cholespy` (solvers.py:3), which is not installable here. A stand-in module
  with the two imported names is put in sys.modules so that the file imports; the stand-in
  CholeskySolverF is an fp64 scipy SuperLU solve (so 'Cholesky' fixtures are NOT cholespy outputs,
  they are recorded as `*_direct` and only the CG / autograd fixtures are genuine reference output).
* parameterize.py does `from largesteps.solvers import ...`; the reference modules are registered
  under the names `largesteps`, `largesteps.solvers` etc. in this process only.

The reference cannot travel to the GPU box, hence the committed fixtures.
"""
import os
import sys
import types
import importlib.util

import numpy as np
import torch

REF = "/root/reference/largesteps"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "large-steps-pytorch_amd", "largesteps"))
import synthetic  # noqa: E402  (numpy-only mesh generators; imported as a plain module on purpose)


def load_reference():
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    # --- cholespy stand-in (test-only) ------------------------------------------------------
    chol = types.ModuleType("cholespy")

    class MatrixType:
        COO = 0

    class CholeskySolverF:
        def __init__(self, n, ii, jj, x, mtype):
            A = sp.csc_matrix((x.double().numpy(), (ii.numpy(), jj.numpy())), shape=(n, n))
            self.lu = spla.splu(A, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0,
                                options=dict(SymmetricMode=True))

        def solve(self, b, x):
            x.copy_(torch.from_numpy(self.lu.solve(b.double().numpy())).to(x.dtype))

    chol.MatrixType = MatrixType
    chol.CholeskySolverF = CholeskySolverF
    sys.modules["cholespy"] = chol

    pkg = types.ModuleType("largesteps")
    pkg.__path__ = [REF]
    sys.modules["largesteps"] = pkg

    src = open(os.path.join(REF, "geometry.py")).read()
    assert src.count("'cuda'") == 4
    geometry = types.ModuleType("largesteps.geometry")
    exec(compile(src.replace("'cuda'", "'cpu'"), "<reference geometry.py, device patched>", "exec"), geometry.__dict__)
    sys.modules["largesteps.geometry"] = geometry

    def load(name):
        spec = importlib.util.spec_from_file_location(f"largesteps.{name}", os.path.join(REF, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"largesteps.{name}"] = mod
        spec.loader.exec_module(mod)
        return mod

    solvers = load("solvers")
    parameterize = load("parameterize")
    optimize = load("optimize")
    return geometry, solvers, parameterize, optimize


def small_meshes():
    m = {}
    # G1 octahedron
    m["octahedron"] = (
        np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32),
        np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]], np.int64))
    # G2 corner tetrahedron
    m["tetra"] = (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32),
                  np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]], np.int64))
    # G3 open quad
    m["quad"] = (np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32),
                 np.array([[0, 1, 2], [0, 2, 3]], np.int64))
    # G4 degenerate collinear triangle (Heron clamp path)
    m["collinear"] = (np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0]], np.float32), np.array([[0, 1, 2]], np.int64))
    # G5 unreferenced vertex
    m["unreferenced"] = (np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [5, 5, 5]], np.float32),
                         np.array([[0, 1, 2], [0, 2, 3]], np.int64))
    # non-manifold fan: three triangles sharing edge (0,1); inconsistent orientation on the third
    m["nonmanifold"] = (np.array([[0, 0, 0], [1, 0, 0], [0.5, 1, 0], [0.5, -1, 0.2], [0.5, 0.1, 1]], np.float32),
                        np.array([[0, 1, 2], [1, 0, 3], [0, 1, 4]], np.int64))
    # duplicated face + face with a repeated vertex (degenerate index pattern)
    m["dupface"] = (np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0.3]], np.float32),
                    np.array([[0, 1, 2], [0, 1, 2], [1, 3, 2], [1, 1, 3]], np.int64))
    v, f = synthetic.icosphere(3)
    m["ico3"] = (synthetic.perturb(v, radial=0.05, tangential=0.2, edge=0.4, seed=3), f)
    v, f = synthetic.plane(12)
    rng = np.random.default_rng(7)
    v = v + (rng.uniform(-0.3, 0.3, v.shape) * np.array([1 / 11, 1 / 11, 0.02])).astype(np.float32)
    m["plane12"] = (v, f)
    v, f = synthetic.icosphere(6)
    m["ico6"] = (synthetic.perturb(v, radial=0.05, seed=1), f)
    return m


MATRIX_CASES = [
    ("uni_l10", dict(lambda_=10.0, alpha=None, cotan=False)),
    ("uni_l0p3", dict(lambda_=0.3, alpha=None, cotan=False)),
    ("uni_a0p95", dict(lambda_=123.0, alpha=0.95, cotan=False)),
    ("cot_l2", dict(lambda_=2.0, alpha=None, cotan=True)),
    ("cot_a0p9", dict(lambda_=0.0, alpha=0.9, cotan=True)),
]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    geometry, solvers, parameterize, optimize = load_reference()
    out = {}
    meta = []
    for name, (v, f) in small_meshes().items():
        tv, tf = torch.from_numpy(v), torch.from_numpy(f)
        out[f"{name}/verts"] = v
        out[f"{name}/faces"] = f
        L = geometry.laplacian_uniform(tv, tf)
        out[f"{name}/Luni_idx"] = L.indices().numpy()
        out[f"{name}/Luni_val"] = L.values().numpy()
        Lc = geometry.laplacian_cot(tv, tf).coalesce()
        out[f"{name}/Lcot_idx"] = Lc.indices().numpy()
        out[f"{name}/Lcot_val"] = Lc.values().numpy()
        for cname, kw in MATRIX_CASES:
            M = geometry.compute_matrix(tv, tf, **kw)
            assert M.is_coalesced()
            out[f"{name}/{cname}/idx"] = M.indices().numpy()
            out[f"{name}/{cname}/val"] = M.values().numpy()
            u = parameterize.to_differential(M, tv)
            out[f"{name}/{cname}/u"] = u.numpy()
            meta.append((name, cname))
        # int32 faces: uniform works, cot raises
        M32 = geometry.compute_matrix(tv, tf.int(), 10.0)
        assert torch.equal(M32.indices(), geometry.compute_matrix(tv, tf, 10.0).indices())

    # solves: only on the well-posed meshes (CG of the reference has no iteration cap)
    for name in ["octahedron", "tetra", "ico3", "plane12", "ico6"]:
        v, f = small_meshes()[name]
        tv, tf = torch.from_numpy(v), torch.from_numpy(f)
        for cname, kw in [("uni_l10", dict(lambda_=10.0)), ("cot_a0p9", dict(lambda_=0.0, alpha=0.9, cotan=True))]:
            M = geometry.compute_matrix(tv, tf, **kw)
            u = parameterize.to_differential(M, tv).clone().requires_grad_(True)
            x = parameterize.from_differential(M, u, "CG")
            w = torch.from_numpy(np.random.default_rng(5).standard_normal(v.shape).astype(np.float32))
            (x * w).sum().backward()
            out[f"{name}/{cname}/cg_x"] = x.detach().numpy()
            out[f"{name}/{cname}/cg_w"] = w.numpy()
            out[f"{name}/{cname}/cg_grad_u"] = u.grad.numpy()
            # second call: warm start from the previous solution (solvers.py:102-124)
            x2 = parameterize.from_differential(M, u.detach() * 1.01, "CG")
            out[f"{name}/{cname}/cg_x_warm"] = x2.detach().numpy()
            xd = parameterize.from_differential(M, u.detach(), "Cholesky")
            out[f"{name}/{cname}/direct_x"] = xd.numpy()

    # error strings (geometry.py:130-131, parameterize.py:55, solvers.py:112-113)
    v, f = small_meshes()["quad"]
    tv, tf = torch.from_numpy(v), torch.from_numpy(f)
    errs = {}
    for a in (1.0, -0.1, 1.5):
        try:
            geometry.compute_matrix(tv, tf, 1.0, alpha=a)
        except ValueError as e:
            errs[f"alpha={a}"] = str(e)
    M = geometry.compute_matrix(tv, tf, 1.0)
    try:
        parameterize.from_differential(M, tv, "LU")
    except ValueError as e:
        errs["method"] = str(e)
    try:
        solvers.ConjugateGradientSolver(M).solve(tv[:, 0])
    except ValueError as e:
        errs["cg_shape"] = str(e)
    try:
        solvers.Solver(M).solve(tv)
    except NotImplementedError as e:
        errs["base"] = repr(e)

    # AdamUniform trajectory (optimize.py:18-41), 5 steps on a fixed quadratic
    p = torch.nn.Parameter(torch.from_numpy(np.random.default_rng(11).standard_normal((7, 3)).astype(np.float32)))
    tgt = torch.from_numpy(np.random.default_rng(12).standard_normal((7, 3)).astype(np.float32))
    out["adam/p0"] = p.detach().numpy().copy()
    out["adam/target"] = tgt.numpy()
    opt = optimize.AdamUniform([p], lr=0.05, betas=(0.9, 0.999))
    traj = []
    for _ in range(5):
        opt.zero_grad()
        ((p - tgt) ** 2).sum().backward()
        opt.step()
        traj.append(p.detach().numpy().copy())
    out["adam/traj"] = np.stack(traj)

    np.savez_compressed(os.path.join(HERE, "reference_golden.npz"), **out)
    with open(os.path.join(HERE, "reference_errors.txt"), "w") as fh:
        for k, s in errs.items():
            fh.write(f"{k}\t{s}\n")
    print("wrote", len(out), "arrays;", os.path.getsize(os.path.join(HERE, "reference_golden.npz")) // 1024, "KiB")
    for k, s in errs.items():
        print(k, "->", s)


if __name__ == "__main__":
    main()
