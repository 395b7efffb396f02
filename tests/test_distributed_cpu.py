"""
N > 1 path on CPU: the host logic of largesteps.distributed (ShardPlan: block partition, local column ids,
halo / send lists) and the iteration driver ShardedPCG on real multi-process collectives (gloo, world size
2 and 3, 127.0.0.1). The local kernels are replaced by their numpy statement (tests/dist_worker.py); the
HIP kernels on shards are covered by tests/test_distributed_gpu.py (-m gpu).
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dist_worker  # noqa: E402
from largesteps.distributed import ShardPlan, block_bounds  # noqa: E402
from oracle import solve as osv  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("depth", [1, 3])
@pytest.mark.parametrize("mesh", ["plane40", "ico12cot"])
@pytest.mark.parametrize("P", [1, 2, 3, 4, 8])
def test_shard_plan(mesh, P, depth):
    v, rowptr, col, val = dist_worker.test_matrix(mesh)
    V = v.shape[0]
    A = sp.csr_matrix((val.astype(np.float64), col, rowptr), shape=(V, V))
    plans = [ShardPlan.build(rowptr, col, val, V, P, r, depth=depth) for r in range(P)]
    bounds = block_bounds(V, P)
    assert bounds[0] == 0 and bounds[-1] == V and sum(p.n_own for p in plans) == V
    x = np.random.default_rng(0).standard_normal((V, 3))
    y = A @ x
    for p in plans:
        assert p.lo == bounds[p.rank] and p.hi == bounds[p.rank + 1]
        assert p.rowptr.dtype == np.int32 and p.col.dtype == np.int32 and p.col.max(initial=0) < p.n_cols
        # local SpMV on [owned | ghosts] reproduces the global one on EVERY computed row (owned + layers < depth)
        glob = np.concatenate([np.arange(p.lo, p.hi), p.halo_global])
        assert np.unique(glob).shape[0] == glob.shape[0]
        x_ext = x[glob]
        A_loc = sp.csr_matrix((p.val.astype(np.float64), p.col, p.rowptr), shape=(p.n_rows, p.n_cols))
        np.testing.assert_allclose(A_loc @ x_ext, y[glob[:p.n_rows]], rtol=1e-12, atol=1e-12)
        # the diagonal stays at local id == row
        assert np.allclose(A_loc.diagonal(), A.diagonal()[glob[:p.n_rows]])
        assert p.n_rows == p.n_own if depth == 1 else p.n_rows >= p.n_own
        # ghosts are disjoint from the block, and the recv list tiles them by (group, owner)
        assert not ((p.halo_global >= p.lo) & (p.halo_global < p.hi)).any()
        assert sum(c for _, _, c in p.recv) == p.n_halo
        msgs_from = {}
        for q, off, cnt in p.recv:
            seg = p.halo_global[off:off + cnt]
            assert (seg >= bounds[q]).all() and (seg < bounds[q + 1]).all() and (np.diff(seg) > 0).all()
            msgs_from.setdefault(q, []).append(seg)
        # what I receive from q is exactly what q sends to me, message by message, in the same order
        for q, segs in msgs_from.items():
            sent = [idx.astype(np.int64) + plans[q].lo for dst, idx in plans[q].send if dst == p.rank]
            assert len(sent) == len(segs) and all(np.array_equal(a, b) for a, b in zip(sent, segs))
        for dst, idx in p.send:
            assert any(src == p.rank for src, _, _ in plans[dst].recv)
    if P == 1:
        assert plans[0].n_halo == 0 and not plans[0].send and not plans[0].recv
    with pytest.raises(ValueError):
        ShardPlan.build(rowptr, col, val, V, P, P)
    # the native analysis (csrc/shard_plan.cpp) against its numpy statement, array for array
    import shard_plan_statement as sps
    for p in plans:
        q = sps.ShardPlan.build(rowptr, col, val, V, P, p.rank, depth=depth)
        assert (p.lo, p.hi, p.n_rows, p.n_cols, p.n_halo) == (q.lo, q.hi, q.n_rows, q.n_cols, q.n_halo)
        for name in ("rowptr", "col", "val", "halo_global"):
            assert np.array_equal(getattr(p, name), getattr(q, name)), name
        assert p.recv == q.recv and len(p.send) == len(q.send)
        assert all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(p.send, q.send))


def run_world(tmp_path, world, mesh, k=3, ops="numpy", backend="gloo", timeout=300, solver="pcg", depth=1):
    port = free_port()
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "--rank", str(r), "--world", str(world),
                               "--port", str(port), "--out", str(tmp_path), "--mesh", mesh, "--k", str(k), "--ops", ops,
                               "--backend", backend, "--solver", solver, "--depth", str(depth)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    xs = [np.load(os.path.join(tmp_path, f"x_{r}.npy")) for r in range(world)]
    x = xs if solver in ("cols", "direct") else np.concatenate(xs)
    its = [np.load(os.path.join(tmp_path, f"it_{r}.npy")) for r in range(world)]
    return x, its


def reference_solution(mesh, k=3):
    v, rowptr, col, val = dist_worker.test_matrix(mesh)
    V = v.shape[0]
    r = np.repeat(np.arange(V), np.diff(rowptr))
    b = (sp.csr_matrix((val.astype(np.float64), col, rowptr)) @ v.astype(np.float64)).astype(np.float32)
    if k != 3:
        b = np.random.default_rng(3).standard_normal((V, k)).astype(np.float32)
    return osv.from_differential(r, col, val, b)


@pytest.mark.parametrize("mesh,world", [("plane40", 2), ("ico12cot", 2), ("plane40", 3)])
def test_sharded_pcg_gloo(tmp_path, mesh, world):
    x64 = reference_solution(mesh)
    x, its = run_world(tmp_path, world, mesh)
    assert all(int(i[1]) == 1 for i in its), "every rank reports convergence"
    assert len({int(i[0]) for i in its}) == 1, "every rank stops at the same iteration"
    assert np.abs(x - x64).max() <= 1e-4 * np.abs(x64).max()
    # same iteration count (+-2) as the unsharded run of the same statement
    single = tmp_path / "single"
    single.mkdir()
    x1, its1 = run_world(single, 1, mesh)
    assert abs(int(its1[0][0]) - int(its[0][0])) <= 2
    assert np.abs(x - x1).max() <= 2e-5 * np.abs(x64).max()


@pytest.mark.parametrize("mesh,world,depth", [("plane40", 2, 1), ("plane40", 2, 4), ("ico12cot", 2, 3), ("plane40", 3, 5)])
def test_sharded_chebyshev_gloo(tmp_path, mesh, world, depth):
    """Chebyshev shards: halo exchange only, once per `depth` iterations; same answer as the unsharded run."""
    x64 = reference_solution(mesh)
    x, its = run_world(tmp_path, world, mesh, solver="cheb", depth=depth)
    assert all(int(i[1]) == 1 for i in its) and len({int(i[0]) for i in its}) == 1
    assert np.abs(x - x64).max() <= 1e-4 * np.abs(x64).max()
    single = tmp_path / "single"
    single.mkdir()
    x1, its1 = run_world(single, 1, mesh, solver="cheb", depth=1)
    assert int(its1[0][0]) == int(its[0][0]), "the schedule depends on the global spectrum only"
    assert np.abs(x - x1).max() <= 1e-5 * np.abs(x64).max()


@pytest.mark.parametrize("world,k", [(2, 3), (3, 3), (4, 3), (2, 1), (3, 5)])
def test_column_sharded_gloo(tmp_path, world, k):
    """Right-hand-side sharding: rank r solves the columns c = r (mod min(P, k)), one all-gather returns the full
    solution to every rank -- also when ranks idle (P > k) or hold several columns (k > P)."""
    x64 = reference_solution("plane40", k)
    xs, cols = run_world(tmp_path, world, "plane40", k=k, solver="cols")
    owned = sorted(int(c) for cl in cols for c in cl[:-1])
    assert owned == list(range(k)), "every column is solved exactly once"
    assert sum(1 for cl in cols if len(cl) > 1) == min(world, k)
    for x in xs:
        assert x.shape == x64.shape and np.array_equal(x, xs[0])
        assert np.abs(x - x64).max() <= 2e-6 * np.abs(x64).max()
