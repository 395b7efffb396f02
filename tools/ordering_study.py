"""Symbolic cost of the nested-dissection plan on surfaces that are folded in space (host only, no GPU):
flat sheet, folded sheet, scrolls, concentric shells, the cfg3 stand-in, a strip -- with the caller's positions, with the
graph-distance embedding, and with what the library picks (mode -1). Prints factor words per vertex and the largest front."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "large-steps-pytorch_amd"), os.path.join(ROOT, "tests")]
from largesteps import synthetic  # noqa: E402


def csr_of(f, V):
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    e = np.concatenate([e, e[:, ::-1], np.stack([np.arange(V), np.arange(V)], 1)])
    key = np.unique(e[:, 0].astype(np.int64) * V + e[:, 1])
    r, c = key // V, key % V
    rowptr = np.zeros(V + 1, np.int64)
    np.add.at(rowptr, r + 1, 1)
    return np.cumsum(rowptr).astype(np.int32), c.astype(np.int32)


def meshes(n=500):
    out = {"flat": synthetic.plane(n), "folded": synthetic.folded_sheet(n)}
    for T in (3, 5, 10):
        out[f"scroll{T}"] = synthetic.scroll(n, T)
    out["shells"] = synthetic.shells(int(round((n * n / 20.0) ** 0.5)))
    cv, cf, _ = synthetic.config_mesh("cfg3_dragon250k" if n >= 400 else "cfg2_bunny70k")
    out["rough"] = (cv, cf)
    out["strip"] = strip(max(8, n // 10), n * 10)
    return out


def strip(nx=50, ny=5000):
    xs, ys = np.arange(nx), np.arange(ny)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    v = np.stack([X, Y, 0 * X], -1).reshape(-1, 3).astype(np.float32)
    i = (ys[:-1, None] * nx + xs[None, :-1]).reshape(-1)
    f = np.concatenate([np.stack([i, i + 1, i + nx + 1], 1), np.stack([i, i + nx + 1, i + nx], 1)])
    return v, f


if __name__ == "__main__":
    from native_plan import native_plan
    n = int(os.environ.get("ORDERING_N", "500"))
    names = sys.argv[1:]
    print(f"# leaf 64, arity 4; factor numbers per vertex (spread), largest front, seconds on {os.cpu_count()} host cores")
    for name, (v, f) in meshes(n).items():
        if names and name not in names:
            continue
        rowptr, col = csr_of(f, v.shape[0])
        row = [f"{name:9s} V={v.shape[0]:7d}"]
        for label, pos, ordering in (("positions, longest axis", v, 0), ("graph distances", None, 0), ("six trial cuts", v, 1), ("automatic", v, -1)):
            t = time.time()
            p = native_plan(rowptr, col, pos, 64, 4, ordering=ordering)
            took = "" if ordering >= 0 else f" took {'six trial cuts' if p.ordering == 1 else 'longest axis'}"
            row.append(f"{label}: {p.words_per_vertex:7.1f} ({p.spread:5.2f}) front {int((p.s + p.b).max()):6d} {time.time() - t:5.2f} s{took}")
        print(" | ".join(row), flush=True)
