"""
bench.py's byte model and JSON plumbing, on the CPU (no GPU, no solver): the algorithmic-bytes accounting must be the one
SURVEY.md section 8(d) states, and the PMC traffic lookup must combine the tree-level kernels by dispatch count.
"""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_match_survey():
    b = load_bench()
    V, nnz, k = 1000000, 6992002, 3
    pcg = b.algorithmic_bytes(V, nnz, k, 136, "pcg")
    # B_iter = 8 nnz + 144 V = 199.9 MB at config 4 (the + 4 is rowptr's V + 1-th entry)
    assert abs(pcg["iter"] - (8 * nnz + 144 * V)) <= 8 and round(pcg["iter"] / 1e6, 1) == 199.9
    assert pcg["k1"] == 8 * nnz + 4 * (V + 1) + 2 * 4 * k * V                      # B_spmv = 8 nnz + 28 V
    assert pcg["solve"] == (4 * k + 1) * 4 * V + 136 * pcg["iter"]
    cheb = b.algorithmic_bytes(V, nnz, k, 180, "chebyshev")
    assert cheb["iter"] == 8 * nnz + 4 * (V + 1) + (4 * k + 1) * 4 * V
    imp = b.algorithmic_bytes(V, nnz, k, 180, "chebyshev", implicit_values=True)
    assert imp["iter"] == 4 * (nnz - V) + 4 * (V + 1) + (4 * k + 1) * 4 * V and round(imp["iter"] / 1e6) == 80
    assert b.HBM_PEAK_GBS == 8000.0 and b.WORKLOAD == "cfg4_plane1m"


def test_pmc_traffic_lookup(tmp_path, monkeypatch):
    b = load_bench()
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / b.PMC_FILE).write_text(json.dumps(dict(workload="cfg4_plane1m", kernels={
        "ls::k_nd_down<3>": dict(dispatches=3, traffic_bytes=100.0),
        "ls::k_nd_down_b<3>": dict(dispatches=5, traffic_bytes=20.0),
        "ls::k_cheb<3, 512, false>": dict(dispatches=7, traffic_bytes=9.0)})))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    assert b.pmc_traffic("ls::k_nd_down", "cfg4_plane1m") == (3 * 100.0 + 5 * 20.0) / 8
    assert b.pmc_traffic("ls::k_cheb<3", "cfg4_plane1m") == 9.0
    assert b.pmc_traffic("ls::k_nd_down", "another_workload") is None
    assert b.pmc_traffic("ls::nothing", "cfg4_plane1m") is None
    # one kernel GROUP (what roofline.traffic of the direct solver's line reports): dispatch-weighted over its kernels only
    t, n = b.pmc_traffic_group(("ls::k_nd_down", "ls::k_cheb"), "cfg4_plane1m")
    assert n == 15 and abs(t - (3 * 100.0 + 5 * 20.0 + 7 * 9.0) / 15) < 1e-12
    assert b.pmc_traffic_group(("ls::nothing",), "cfg4_plane1m") == (None, 0)


def test_committed_profiles_are_consistent():
    """the committed bench line of the final run carries the contract's keys and agrees with itself"""
    import glob
    path = sorted((glob.glob(os.path.join(ROOT, "profiles", "r03_*bench_direct*.json")) or glob.glob(os.path.join(ROOT, "profiles", "r02_*bench_direct*.json"))) or [os.path.join(ROOT, "profiles", "r01_run18_bench_direct.json")])[-1]
    line = [ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d
    assert d["metric"] == "from_differential_solves_per_sec" and d["n_gpus"] == 1 and d["vs_baseline"] is None
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) <= 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def test_roofline_group_resolves_in_the_committed_pmc_file():
    """`roofline.traffic` is looked up by kernel-name PREFIX in the committed PMC summary: a template argument added to a kernel
    (k_nd_tier<K, UP> became <K, UP, WAVES>) once made the lookup silently drop the group's largest kernel. Every prefix of the
    group must resolve, the launches must be in the ratio one solve has (n - 1 level launches : 1 tier launch), and the traffic the
    line would print must not be BELOW the algorithmic bytes of the committed bench line."""
    import glob
    b = load_bench()
    with open(os.path.join(ROOT, "profiles", b.PMC_FILE)) as fh:
        doc = json.load(fh)
    assert doc["workload"] == b.WORKLOAD
    per_prefix = {}
    for p in b.direct_group_prefixes():
        hits = {k: v for k, v in doc["kernels"].items() if k.startswith(p)}
        assert hits, f"no kernel of {b.PMC_FILE} starts with {p!r}"
        per_prefix[p] = sum(v["dispatches"] for v in hits.values())
    levels, tier = per_prefix["ls::k_nd_down"], per_prefix["ls::k_nd_tier<3, false"]
    # the up-sweep tier kernel must not be caught by the down-sweep prefix
    assert all(k.split(",")[1].strip() == "false" for k in doc["kernels"] if k.startswith("ls::k_nd_tier<3, false"))
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", b.PMC_FILE[:3] + "_*bench_direct*.json")))
    d = json.loads([ln for ln in open(lines[-1]).read().splitlines() if ln.startswith("{")][-1])
    n_down = d["config"]["kernel_us"]["down_launches"]
    assert tier > 0 and levels == (n_down - 1) * tier, (levels, tier, n_down)
    traffic, n = b.pmc_traffic_group(b.direct_group_prefixes(), b.WORKLOAD)
    assert n == levels + tier
    assert traffic >= 0.9 * d["roofline"]["bytes_per_launch"], (traffic, d["roofline"]["bytes_per_launch"])
    assert traffic <= 1.5 * d["roofline"]["bytes_per_launch"]


def test_plain_multi_gpu_start_launches_the_ranks_itself(monkeypatch):
    """`python bench.py --gpus 8` without RANK in the environment (the shape of the driver's N = 1 command) must not die with KeyError:
    bench.py starts torch.distributed.run itself, with the driver's own arguments, and picks the loopback transport when the box has
    fewer devices than ranks."""
    import subprocess
    import sys
    import types
    import torch
    bench = load_bench()
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=0)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    for n_dev, loop in ((1, True), (8, False)):
        monkeypatch.setattr(torch.cuda, "device_count", lambda n=n_dev: n)
        monkeypatch.delenv("LS_DIST_LOOPBACK", raising=False)
        rc = bench.launch_ranks(types.SimpleNamespace(gpus=8))
        assert rc == 0
        cmd = seen["cmd"]
        assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
        assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
        assert cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"] and cmd[-7].endswith("bench.py")
        assert (seen["env"].get("LS_DIST_LOOPBACK") == "1") == loop
        assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_shard_model_is_the_committed_table():
    """the N > 1 line carries DESIGN.md section 5's prediction for its N (so that a SCALE record tests the model): the rows are the
    per-rank kernel times of the committed profiles/r06_run1_shard_rank_kernel_times.txt"""
    bench = load_bench()
    text = open(os.path.join(ROOT, "profiles", bench.SHARD_MODEL_FILE)).read()
    for wl, rows in bench.SHARD_MODEL_US.items():
        for n, (kernel_us, above) in rows.items():
            mine = [float(ln.split(":")[2].split("us")[0]) for ln in text.splitlines() if ln.startswith(f"{wl}: rank") and f" of {n}:" in ln]
            assert mine and min(mine) - 1.0 <= kernel_us <= max(mine) + 1.0, (wl, n, mine, kernel_us)
            m = bench.shard_model(wl, n)
            assert m["kernel_us_per_rank"] == kernel_us and m["predicted_ms_per_step"][0] <= m["predicted_ms_per_step"][1]
    assert bench.shard_model("cfg2_bunny70k", 2) is None and bench.shard_model("cfg4_plane1m", 3) is None
    one = bench.shard_model("cfg4_plane1m", 1)
    assert one["collective_us_assumed"] == [0.0, 0.0]
