"""
Parity cases of the persistent upper-level launch (csrc/nd_span.h), an EXPERIMENT that is compiled only into the -DLS_ND_EXPERIMENTS
build of the library. Not collected by the suite (the file name does not match test_*.py): tests/test_gpu_parity.py runs it in a
process whose LARGESTEPS_HIP_LIB points at tools/build/liblargesteps_hip_exp.so.
"""
import numpy as np
import pytest
import torch

from oracle import solve as osv

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda", 0)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("grid", [0, 8, 24])
@pytest.mark.parametrize("mesh,arity,leaf,k", [("plane120", 4, 64, 3), ("plane120", 2, 24, 1), ("plane120", 4, 6, 4), ("ico30cot", 4, 24, 3),
                                               ("ico30cot", 8, 16, 2), ("plane300", 4, 64, 3)])
def test_persistent_upper_levels(dev, monkeypatch, mesh, arity, leaf, k, grid):
    """The levels above the tier as ONE persistent launch (csrc/nd_span.h, "persist" = 1: tree-local barriers between the
    phases, write-through hand-offs): same answer as one launch per level, vs the fp64 oracle at the solver's tolerance,
    bitwise reproducible. grid 8 / 24: fewer workgroups than tree nodes (several jobs per workgroup and phase, ranges shared
    by siblings) and a grid that does not divide evenly."""
    from largesteps.geometry import compute_matrix
    from largesteps.solvers import NestedDissectionSolver
    from largesteps import synthetic
    if mesh.startswith("plane"):
        v, f = synthetic.plane(int(mesh[5:]))
        M = compute_matrix(_t(v, dev), _t(f, dev), 25.0)
    else:
        v, f = synthetic.icosphere(30)
        v = synthetic.perturb(v, radial=0.05, tangential=0.1, edge=0.05, seed=2)
        M = compute_matrix(_t(v, dev), _t(f, dev), 0.0, alpha=0.9, cotan=True)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    b = np.random.default_rng(7).standard_normal((v.shape[0], k)).astype(np.float32)
    x64 = osv.from_differential(idx[0], idx[1], val, b)
    if grid:
        monkeypatch.setenv("LS_ND_SPAN_GRID", str(grid))
    s = NestedDissectionSolver(M, leaf_size=leaf, arity=arity)
    x0 = s.solve(_t(b, dev))
    n0 = s.info()["launches"]
    s.set_option("persist", 1)
    assert s.info()["launches"] == 3 < n0, "tier up, the persistent launch, tier down"
    x1 = s.solve(_t(b, dev))
    assert np.abs(x1.cpu().numpy() - x64).max() <= 2e-5 * np.abs(x64).max()
    assert float((x1 - x0).abs().max()) <= 2e-5 * np.abs(x64).max()
    for _ in range(3):
        assert torch.equal(x1, s.solve(_t(b, dev))), "fixed reduction order, no atomics on data: bitwise reproducible"
    s.set_option("persist", 0)
    assert torch.equal(x0, s.solve(_t(b, dev)))
