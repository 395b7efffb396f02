"""
The HIP kernels on vertex-block shards (-m gpu). The test box has ONE GPU, so the ranks are separate
processes that share cuda:0 and talk through gloo (all-reduce of device tensors; halo rows staged through the
host = the loopback transport of SURVEY.md §8e). What this covers that the CPU tests cannot: the rectangular
shard matrices in SELL-64, halo packing (ls_gather_rows), ls_solver_phase / bind / poll, and that the sharded
result equals the single-GPU one. RCCL itself only replaces the transport underneath torch.distributed.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_distributed_cpu import run_world, reference_solution  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mesh,world,k", [("plane40", 1, 3), ("plane40", 2, 3), ("ico12cot", 2, 3), ("plane40", 4, 2)])
def test_sharded_hip(tmp_path, mesh, world, k):
    x64 = reference_solution(mesh, k)
    x, its = run_world(tmp_path, world, mesh, k=k, ops="hip", timeout=600)
    assert all(int(i[1]) == 1 for i in its)
    assert len({int(i[0]) for i in its}) == 1
    assert np.abs(x - x64).max() <= 1e-4 * np.abs(x64).max()


@pytest.mark.parametrize("mesh,world,depth", [("plane40", 1, 1), ("plane40", 2, 4), ("ico12cot", 2, 2), ("plane40", 4, 3)])
def test_sharded_chebyshev_hip(tmp_path, mesh, world, depth):
    x64 = reference_solution(mesh)
    x, its = run_world(tmp_path, world, mesh, ops="hip", timeout=600, solver="cheb", depth=depth)
    assert all(int(i[1]) == 1 for i in its) and len({int(i[0]) for i in its}) == 1
    assert np.abs(x - x64).max() <= 1e-4 * np.abs(x64).max()


@pytest.mark.parametrize("world,k", [(2, 3), (3, 3), (4, 2)])
def test_column_sharded_hip(tmp_path, world, k):
    """Right-hand-side sharding with the HIP solver as the local solver (loopback: gloo all-gather staged through the host)."""
    x64 = reference_solution("plane40", k)
    xs, cols = run_world(tmp_path, world, "plane40", k=k, ops="hip", timeout=600, solver="cols")
    assert sorted(int(c) for cl in cols for c in cl[:-1]) == list(range(k))
    for x in xs:
        assert np.array_equal(x, xs[0])
        assert np.abs(x - x64).max() <= 1e-4 * np.abs(x64).max()


def test_shard_from_matrix_single_rank():
    """shard_from_matrix with no process group (P = 1) must reproduce from_differential bit for bit."""
    import torch
    from largesteps import synthetic
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import to_differential, from_differential
    from largesteps.distributed import shard_from_matrix
    dev = torch.device("cuda:0")
    v, f = synthetic.plane(300)
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, 50.0)
    u = to_differential(M, tv)
    plan, solver = shard_from_matrix(M, method="pcg")
    assert plan.n_own == v.shape[0] and plan.n_halo == 0
    x = solver.solve(u)
    from largesteps.solvers import PCGSolver
    ref_solver = PCGSolver(M, rtol=1e-6)
    ref_solver.set_option("block", 256)
    ref = ref_solver.solve(u)
    assert solver.last_info["converged"]
    assert torch.equal(x, ref), "same kernels, same grid, same reduction order"
    # and the Chebyshev shard driver (P = 1: no exchange) against the single-GPU Chebyshev path
    plan_c, cheb = shard_from_matrix(M, method="auto")
    xc = cheb.solve(u)
    ref_c = from_differential(M, u, "Cholesky")
    assert cheb.last_info["method"] == "chebyshev" and cheb.last_info["converged"]
    # (the single-GPU path reads implicit uniform values, the shard the stored ones: same iterate up to rounding)
    assert float((xc - ref_c).abs().max()) <= 1e-5 * float(ref_c.abs().max())


@pytest.mark.parametrize("mesh,world,k", [("plane120", 1, 3), ("plane120", 2, 3), ("plane120", 4, 3), ("ico24cot", 2, 3), ("ico24cot", 4, 2),
                                          ("plane40", 3, 1)])
def test_sharded_direct_solver(tmp_path, mesh, world, k):
    """The nested-dissection direct solver sharded by subtrees over `world` processes that share the one GPU (gloo loopback):
    the HIP kernels on every rank's subtrees, ONE summed exchange per solve. Every rank ends with the same full x, equal to
    the fp64 oracle to the single-GPU solver's tolerance; ownership partitions the rows."""
    x64 = reference_solution(mesh, k)
    xs, info = run_world(tmp_path, world, mesh, k=k, ops="hip", timeout=600, solver="direct")
    for x in xs:
        assert np.array_equal(x, xs[0])
        assert np.abs(x - x64).max() <= 2e-5 * np.abs(x64).max()
    assert sum(int(i[2]) for i in info) == x64.shape[0]
    if world > 1:
        assert all(int(i[0]) >= 1 and int(i[1]) > 0 for i in info)


@pytest.mark.parametrize("arity,cut_level", [(4, 2), (8, 1)])
def test_sharded_direct_solver_at_world_eight_in_one_process(arity, cut_level):
    """N = 8 cuts the arity-4 elimination tree one level deeper than N = 2 / 4 (level 2: sixteen subtrees, two per rank; levels 0 and 1
    replicated); the arity-8 tree is cut at level 1 (eight subtrees, one per rank, only the root replicated). Eight shard handles in ONE process stand in for the eight ranks: part 0 on each, the exchange buffers summed
    (what the all-reduce does), part 1 on each, x stitched by row ownership -- against the fp64 oracle at the solver's
    tolerance and against the unsharded solver."""
    import ctypes
    import torch
    from largesteps import synthetic, _native
    from largesteps.geometry import compute_matrix
    from largesteps.solvers import NestedDissectionSolver
    from oracle import solve as osv
    dev = torch.device("cuda:0")
    v, f = synthetic.plane(300)
    M = compute_matrix(torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev), 25.0)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    k, world = 3, 8
    b_np = np.random.default_rng(2).standard_normal((v.shape[0], k)).astype(np.float32)
    x64 = osv.from_differential(idx[0], idx[1], val, b_np)
    b = torch.from_numpy(b_np).to(dev)
    lib = _native.lib()
    ranks = [NestedDissectionSolver(M, leaf_size=64, arity=arity, shard=(r, world)) for r in range(world)]
    owned, per_col, cuts = [], None, set()
    for s in ranks:
        rk, cnt, cut, pc = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int64(0)
        mask = np.zeros(v.shape[0], dtype=np.uint8)
        _native.check(lib.ls_direct_shard_info(s._direct._h, ctypes.byref(rk), ctypes.byref(cnt), ctypes.byref(cut), ctypes.byref(pc),
                                               mask.ctypes.data_as(ctypes.c_void_p)))
        assert cnt.value == world
        owned.append(mask.astype(bool)); cuts.add(cut.value); per_col = pc.value
    assert cuts == {cut_level}, "eight ranks cut the arity-4 tree at level 2, the arity-8 tree at level 1"
    assert np.array_equal(np.sum(owned, axis=0), np.ones(v.shape[0])), "ownership partitions the rows"
    ex = [torch.zeros(per_col * k, dtype=torch.float32, device=dev) for _ in ranks]
    xs = [torch.zeros_like(b) for _ in ranks]
    st = _native.stream_of(dev)
    for s, e, x in zip(ranks, ex, xs):
        _native.check(lib.ls_direct_solve_part(s._direct._h, _native.ptr(b), _native.ptr(x), k, 0, _native.ptr(e), st))
    total = torch.stack(ex).sum(0)
    # every entry of the exchange has exactly one non-zero contributor: the sum is exact and order independent
    assert int((torch.stack(ex) != 0).sum(0).max()) <= 1
    for s, x in zip(ranks, xs):
        e = total.clone()
        _native.check(lib.ls_direct_solve_part(s._direct._h, _native.ptr(b), _native.ptr(x), k, 1, _native.ptr(e), st))
    x = torch.zeros_like(b)
    for o, xr in zip(owned, xs):
        m = torch.from_numpy(o).to(dev)
        x[m] = xr[m]
    assert np.abs(x.cpu().numpy() - x64).max() <= 2e-5 * np.abs(x64).max()
    x1 = NestedDissectionSolver(M).solve(b)
    assert float((x - x1).abs().max()) <= 2e-5 * np.abs(x64).max()


def test_single_process_multi_device_mode(monkeypatch):
    """LARGESTEPS_DEVICES: the reference's call sites use the subtree-sharded solver from ONE process, unchanged (SURVEY.md 8e
    process model). On the 1-GPU box the device list repeats device 0 (loopback: the exchange goes through peer copies instead
    of RCCL): from_differential and its backward equal the single-device solver's to the solver's tolerance."""
    import torch
    from largesteps import synthetic, parameterize
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import from_differential, to_differential
    from oracle import solve as osv
    dev = torch.device("cuda:0")
    v, f = synthetic.plane(200)
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, 25.0)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    u = to_differential(M, tv)
    x64 = osv.from_differential(idx[0], idx[1], val, u.cpu().numpy())
    monkeypatch.setenv("LARGESTEPS_DEVICES", "0,0,0,0")
    ur = u.clone().requires_grad_(True)
    x = from_differential(M, ur, "Cholesky")
    solver = parameterize._cache[(id(M), "Cholesky")][0]
    assert solver.method == "nested-dissection" and solver.last_info["devices"] == [0, 0, 0, 0]
    assert np.abs(x.detach().cpu().numpy() - x64).max() <= 2e-5 * np.abs(x64).max()
    w = torch.from_numpy(np.random.default_rng(1).standard_normal(v.shape).astype(np.float32)).to(dev)
    (x * w).sum().backward()
    g64 = osv.from_differential(idx[0], idx[1], val, w.cpu().numpy())
    assert np.abs(ur.grad.cpu().numpy() - g64).max() <= 2e-5 * np.abs(g64).max()


def test_native_collective_at_world_one():
    """ls_dist_* (the sharded solve's all-reduce behind the C ABI, RCCL looked up at run time) on the real device with a
    communicator of ONE rank -- all a 1-GPU box can run: unique id, ncclCommInitRank, an in-place all-reduce on the solve's
    stream, and ls_dist_direct_solve == ls_direct_solve. In a subprocess: RCCL state should not leak into the test process."""
    import subprocess
    code = (
        "import ctypes, numpy as np, torch\n"
        "from largesteps import _native, synthetic\n"
        "from largesteps.geometry import compute_matrix\n"
        "from largesteps.solvers import NestedDissectionSolver\n"
        "dev = torch.device('cuda', 0); lib = _native.lib()\n"
        "buf = (ctypes.c_ubyte * 128)()\n"
        "_native.check(lib.ls_dist_unique_id(buf))\n"
        "c = ctypes.c_void_p(None)\n"
        "_native.check(lib.ls_dist_create(buf, 0, 1, 0, ctypes.byref(c)))\n"
        "t = torch.arange(1000, dtype=torch.float32, device=dev)\n"
        "_native.check(lib.ls_dist_allreduce_sum(c, _native.ptr(t), t.numel(), _native.stream_of(dev)))\n"
        "torch.cuda.synchronize(); assert float(t.sum()) == 499500.0\n"
        "v, f = synthetic.plane(150)\n"
        "M = compute_matrix(torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev), 25.0)\n"
        "b = torch.from_numpy(np.random.default_rng(0).standard_normal((v.shape[0], 3)).astype(np.float32)).to(dev)\n"
        "s = NestedDissectionSolver(M, shard=(0, 1))\n"
        "x0 = s.solve(b); x1 = torch.zeros_like(b)\n"
        "_native.check(lib.ls_dist_direct_solve(c, s._direct._h, _native.ptr(b), _native.ptr(x1), 3, _native.stream_of(dev)))\n"
        "torch.cuda.synchronize(); assert torch.equal(x0, x1)\n"
        "_native.check(lib.ls_dist_destroy(c)); print('ls_dist ok')\n")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=os.pathsep.join(sys.path))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ls_dist ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_rccl_process_group_initialises():
    """backend 'nccl' (= RCCL on ROCm) with world size 1 on the real device: the branch bench.py --gpus N takes, executed once
    on hardware (one all-reduce through RCCL)."""
    import subprocess
    code = ("import os, torch, torch.distributed as dist\n"
            "os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')\n"
            "dev = torch.device('cuda', 0); torch.cuda.set_device(dev)\n"
            "dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)\n"
            "t = torch.arange(8, dtype=torch.float32, device=dev)\n"
            "dist.all_reduce(t); torch.cuda.synchronize()\n"
            "assert float(t.sum()) == 28.0\n"
            "dist.barrier(); dist.destroy_process_group(); print('rccl ok')\n")
    from test_distributed_cpu import free_port
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "rccl ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("world,launch,workload", [(2, "torchrun", "cfg2_bunny70k"), (8, "torchrun", "cfg2_bunny70k"),
                                                   (2, "plain", "cfg2_bunny70k"), (8, "plain", "cfg4_plane1m")])
def test_bench_line_of_a_multi_process_run(world, launch, workload):
    """`bench.py --gpus N` in both forms it may be started in -- as the driver launches N > 1 (torch.distributed.run, one process per rank,
    127.0.0.1) and PLAINLY (`python bench.py --gpus N`, the shape of the driver's N = 1 command: bench.py starts the ranks itself) -- in
    loopback on the one GPU (every rank on cuda:0, gloo transport): rank 0 must print ONE JSON line with the keys the driver parses, for
    the subtree-sharded direct solver, with the sharded answer correct, the ranks that took part, every rank's kernel / collective times
    and, for the headline workload, the model's prediction for that N. (The first SCALE record can only be taken on an 8-GPU node; this
    keeps the command from failing there for a reason a 1-GPU box could have shown.)"""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, LS_POOL_GB="0")
    tail = [os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--workload", workload]
    if launch == "torchrun":
        env["LS_DIST_LOOPBACK"] = "1"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + tail
    else:
        env.pop("LS_DIST_LOOPBACK", None)           # bench.py sees one device for N ranks and picks the loopback transport itself
        for name in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(name, None)
        cmd = [sys.executable] + tail
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "from_differential_solves_per_sec" and d["n_gpus"] == world and d["steps"] == 3 and d["warmup"] == 1
    assert d["scaling"] == "strong" and d["higher_is_better"] is True and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    c = d["config"]
    assert workload in c["workload"] and f"over {world} ranks" in c["workload"] and c["method"] == "nested-dissection"
    assert c["max_abs_err_vs_v"] <= 1e-4, c["max_abs_err_vs_v"]
    rk = c["ranks"]
    assert rk["world_size"] == world and rk["communicator"]["ranks"] == world and "loopback" in rk["transport"]
    assert [p["rank"] for p in rk["per_rank"]] == list(range(world))
    assert sum(p["own_rows"] for p in rk["per_rank"]) == int(c["workload"].split("V=")[1].split(",")[0])
    for p in rk["per_rank"]:
        assert p["part0_us"] > 0 and p["part1_us"] > 0 and p["collective_us"] > 0 and abs(p["kernel_us"] - p["part0_us"] - p["part1_us"]) < 1e-6
    # the run checks itself (round 6): the sharded x against ONE unsharded solve, every row owned once, the communicator's world size;
    # a failed check would have made the command exit with rc 1 above
    chk = c["shard_check"]
    assert chk["ok"] and chk["every_row_has_one_owner"] and chk["communicator_world_matches"]
    assert chk["max_abs_diff_vs_unsharded"] <= chk["tolerance_rel"] * chk["max_abs_x"]
    if workload == "cfg4_plane1m":
        m = c["model"]
        assert m["kernel_us_per_rank"] == 121.6 and m["predicted_ms_per_step"][0] < m["predicted_ms_per_step"][1]
    else:
        assert c["model"] is None
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 * world and abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-9
