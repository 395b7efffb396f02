import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_PARENT = os.path.join(ROOT, "large-steps-pytorch_amd")
for p in (ROOT, PKG_PARENT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests must never silently pass without a device: skip them (visibly) only when
    the marker expression did not ask for them."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """tests/golden/reference_golden.npz: outputs of the reference executed on CPU
    (tests/golden/make_golden.py)."""

    def __init__(self):
        self.z = np.load(os.path.join(ROOT, "tests", "golden", "reference_golden.npz"))
        self.meshes = sorted({k.split("/")[0] for k in self.z.files if k.endswith("/verts")})
        self.cases = ["uni_l10", "uni_l0p3", "uni_a0p95", "cot_l2", "cot_a0p9"]
        self.params = {
            "uni_l10": dict(lambda_=10.0, alpha=None, cotan=False),
            "uni_l0p3": dict(lambda_=0.3, alpha=None, cotan=False),
            "uni_a0p95": dict(lambda_=123.0, alpha=0.95, cotan=False),
            "cot_l2": dict(lambda_=2.0, alpha=None, cotan=True),
            "cot_a0p9": dict(lambda_=0.0, alpha=0.9, cotan=True),
        }
        self.solve_meshes = ["octahedron", "tetra", "ico3", "plane12", "ico6"]
        self.solve_cases = ["uni_l10", "cot_a0p9"]

    def __getitem__(self, k):
        return self.z[k]

    def errors(self):
        out = {}
        with open(os.path.join(ROOT, "tests", "golden", "reference_errors.txt")) as fh:
            for line in fh:
                k, s = line.rstrip("\n").split("\t", 1)
                out[k] = s
        return out


@pytest.fixture(scope="session")
def golden():
    return Golden()


@pytest.fixture(autouse=True)
def _forget_solver_history():
    """The direct solver remembers, per size class / surface, how long the previous solver served and whether the automatic rule ended up with
    the trial cuts (largesteps/solvers.py: a remesh loop tells the library its own period). Process-global heuristics: every test starts
    without them."""
    try:
        from largesteps.solvers import NestedDissectionSolver
        NestedDissectionSolver._served.clear()
        NestedDissectionSolver._suspect.clear()
    except Exception:
        pass
    yield
