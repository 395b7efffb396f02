#!/usr/bin/env python3
"""
bench.py -- from_differential solves/sec on the 1M-vertex plane (BASELINE.json metric, configs[3] / cfg4).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one from_differential solve  M x = u  (M = I + 50 L_uniform of the 1000x1000 plane, u = M v,
k = 3 right-hand sides, every solve from b alone), inputs resident in HBM. What the timed result meets (`config.tolerance`):
  default (direct solver): no stopping rule -- the FORWARD error against the fp64 oracle's solution of the same system,
                           <= 1e-4 relative (north_star's "stated fp32 tolerance"); the line prints the measured value and, for
                           information, the relative residual it happens to have (2e-5 at 1M: NOT below SURVEY 8d's 1e-6, which is
                           a stopping rule for iterations and is what --iterative / --pcg stop at);
  --iterative / --pcg:     ||r|| <= 1e-6 ||b|| per column, forward error reported beside it.
N = 1: the public API path (largesteps.parameterize.from_differential -> CholeskySolver -> C ABI: ls_direct_factor once,
       ls_direct_solve per step -- the nested-dissection direct solver; --iterative / --pcg time the iterative paths).
N > 1 (one rank per GPU, RCCL; largesteps.distributed), strong scaling (total work fixed), modes (--shard):
  vertex (default) : vertex blocks = subtrees of the direct solver's elimination tree: rank r runs its share of the subtrees
                     below the cut level, every rank the few levels above it; ONE all-reduce (sum) of a few hundred KB per solve;
  columns          : the 3 right-hand-side columns are independent systems -> rank r solves column r, one all-gather per solve;
  halo             : N contiguous vertex blocks of the Chebyshev / PCG iteration with ghost layers (neighbour exchanges);
  replicas         : every rank solves its own copy of the system (independent meshes), weak scaling, no communication.

Prints ONE JSON line on rank 0 (contract: see the task description): metric/value/unit/... plus
  "roofline":     HBM roofline of the dominant kernel group (the down sweep of the direct solver: its algorithmic bytes /
                  its duration from HIP events on the solve's own stream, in an extra profiled pass right after the timed
                  region, same workload)
  "cpu_baseline": the oracle's CPU "factor once / re-solve" direct solver (scipy SuperLU, fp64, 1 thread) on
                  the same 1M-vertex system, rank 0 at N = 1 only (bounded: 1 factorisation + 3 solves); "extra": the
                  reference's CG algorithm on the host (70k config) and in stock torch ops on the device (B2 / B3)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "large-steps-pytorch_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured-achievable copy rate: 6290
HBM_ACHIEVABLE_GBS = 6290.0
WORKLOAD = "cfg4_plane1m"


def algorithmic_bytes(V, nnz, k, iters, method="pcg", implicit_values=False):
    """SURVEY.md §8(d) accounting: CSR-equivalent int32/fp32 matrix (8 B per stored entry + 4 B per row), k
    interleaved fp32 right-hand sides, every array counted once per kernel that touches it.
      pcg       K1 matrix + p + Ap | K2 x,p,Ap,r,1/diag -> x,r | K3 r,1/diag,p -> p        = 8 nnz + 4 V + (11k+2) 4V
      chebyshev one kernel: matrix + x_k (gathered) + b + diag + x_{k-1} -> x_{k+1}          = 8 nnz + 4 V + (4k+1) 4V
                (uniform Laplacian, implicit values: 4 (nnz - V) + 4 V + (4k+1) 4V)
    """
    mat = 8 * nnz + 4 * (V + 1)
    if method == "chebyshev":
        # uniform Laplacian: values implicit, the kernel reads the off-diagonal neighbour ids only (4 B each)
        b_iter = (4 * (nnz - V) + 4 * (V + 1) if implicit_values else mat) + (4 * k + 1) * 4 * V
        # setup: zero x0 (write) ; final residual check: matrix + x + b
        return dict(k1=b_iter, iter=b_iter, solve=4 * k * V + iters * b_iter + mat + 2 * 4 * k * V)
    b_spmv = mat + 2 * 4 * k * V
    b_k2 = (4 * k + 1) * 4 * V + 2 * 4 * k * V
    b_k3 = (2 * k + 1) * 4 * V + 4 * k * V
    b_iter = b_spmv + b_k2 + b_k3
    return dict(k1=b_spmv, k2=b_k2, k3=b_k3, iter=b_iter, solve=(4 * k + 1) * 4 * V + iters * b_iter)


def _newest_pmc_file():
    """the newest committed PMC summary (profiles/rNN_pmc_traffic.json, written by tools/pmc_summary.py from the --pmc passes)"""
    import glob
    names = sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    return names[-1] if names else "r03_pmc_traffic.json"


PMC_FILE = _newest_pmc_file()


def direct_group_prefixes():
    """kernel-name prefixes of the launch group the direct solver's `roofline` is quoted on (tools/pmc_summary.py keys the PMC file by
    the full kernel name, template arguments included: the prefixes stop BEFORE the arguments that vary with the tree -- the tier
    kernel is k_nd_tier<K, UP, WAVES>). tests/test_bench_model.py resolves them against the committed PMC file."""
    return ("ls::k_nd_down", "ls::k_nd_tier<3, false")


def _pmc_doc(workload):
    path = os.path.join(ROOT, "profiles", PMC_FILE)
    try:
        with open(path) as fh:
            d = json.load(fh)
        return d if d.get("workload") == workload else None
    except (OSError, ValueError):
        return None


def pmc_traffic(kernel_prefix, workload):
    """HBM-side bytes per launch of a kernel from the committed PMC passes (profiles/r03_pmc_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 correction applied), or None."""
    d = _pmc_doc(workload)
    if not d:
        return None
    hits = [rec for name, rec in d["kernels"].items() if name.startswith(kernel_prefix)]
    if not hits:
        return None
    n = sum(r.get("dispatches", 1) for r in hits)
    return sum(r["traffic_bytes"] * r.get("dispatches", 1) for r in hits) / max(n, 1)


def pmc_traffic_group(prefixes, workload):
    """(bytes per launch, launches counted) over the kernels of one GROUP (every kernel whose name starts with one of the
    prefixes), dispatch weighted -- the same population `roofline.bytes_per_launch` is computed over."""
    d = _pmc_doc(workload)
    if not d:
        return None, 0
    hits = [rec for name, rec in d["kernels"].items() if any(name.startswith(p) for p in prefixes)]
    n = sum(r.get("dispatches", 1) for r in hits)
    if not n:
        return None, 0
    return sum(r["traffic_bytes"] * r.get("dispatches", 1) for r in hits) / n, n


ASSEMBLY_KERNELS = ("ls::k_count<", "ls::k_scan_chained", "ls::k_scatter<", "ls::k_row_merge<", "ls::k_tile_scan", "ls::k_rowptr", "ls::k_emit")


def pmc_traffic_per_call(prefixes, workload):
    """HBM-side bytes of ONE call that launches each of the named kernels once (compute_matrix): the sum of their per-dispatch means in
    the committed PMC file, or None when a kernel of the list is missing there."""
    d = _pmc_doc(workload)
    if not d:
        return None
    total = 0.0
    for p in prefixes:
        hits = [rec["traffic_bytes"] for name, rec in d["kernels"].items() if name.startswith(p)]
        if not hits:
            return None
        total += sum(hits) / len(hits)
    return total


def assembly_entry(tv, tf, cfg, V, F, nnz, workload, repeats=5):
    """`compute_matrix` as its own roofline entry (the first kernel group north_star names): SURVEY 8(d)'s algorithmic bytes -- the faces read
    once, the CSR arrays and the row pointers written once; the int64 COO index list the reference's matrix handle carries is listed beside
    it -- over the time of whole calls between two HIP events (host sync for nnz and torch's allocations included)."""
    from largesteps.geometry import compute_matrix
    lam = cfg["lambda_"] if cfg["lambda_"] is not None else 0.0
    for _ in range(2):
        compute_matrix(tv, tf, lam, alpha=cfg["alpha"], cotan=cfg["cotan"])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(repeats):
        compute_matrix(tv, tf, lam, alpha=cfg["alpha"], cotan=cfg["cotan"])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / repeats
    csr_bytes = tf.element_size() * 3 * F + 8 * nnz + 4 * (V + 1) + (12 * V if cfg["cotan"] else 0)
    coo_bytes = 16 * nnz
    traffic = pmc_traffic_per_call(ASSEMBLY_KERNELS, workload)
    return dict(kernel="compute_matrix: k_count (corner ranks, one atomic per vertex pair and workgroup) -> k_scan_chained (one launch) -> k_scatter -> k_row_merge (rows sorted in "
                       "registers) -> k_tile_scan / k_rowptr -> [nnz to the host] -> k_emit", us_per_call=us,
                bytes=csr_bytes, bytes_with_coo_index_list=csr_bytes + coo_bytes, achieved=csr_bytes / (us * 1e-6) / 1e9, unit="GB/s",
                frac=csr_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, frac_with_coo_index_list=(csr_bytes + coo_bytes) / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                traffic=traffic, traffic_source=f"profiles/{PMC_FILE}, sum of the per-dispatch means of the call's kernels" if traffic is not None else None)


def spmv_entry(M, tv, V, nnz, workload, repeats=20):
    """`to_differential` (ls_spmv, csrc/spmv.hip) as its own roofline entry: SURVEY 8(d)'s 8 nnz + 4 (V + 1) + 2 x 4 k V bytes over the mean of
    `repeats` back-to-back calls between two HIP events (matrix and vectors then sit in the Infinity Cache: the warm figure; the one call per
    (re)mesh of the reference's loop meets colder data -- profiles/r05_spmv.txt lists both)."""
    from largesteps.parameterize import to_differential
    k = tv.shape[1]
    for _ in range(3):
        to_differential(M, tv)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(repeats):
        to_differential(M, tv)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / repeats
    bts = 8 * nnz + 4 * (V + 1) + 2 * 4 * k * V
    # ... and COLD: the reference's loop calls to_differential once per (re)mesh, on data no cache holds. 1 GB is overwritten in front of
    # every timed call (4x the Infinity Cache), one call between two events, median of 5
    flush = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=tv.device)
    cold = []
    for _ in range(5):
        flush.fill_(1.0)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        to_differential(M, tv)
        c1.record()
        torch.cuda.synchronize()
        cold.append(c0.elapsed_time(c1) * 1e3)
    del flush
    us_cold = sorted(cold)[len(cold) // 2]
    return dict(kernel="k_spmv<3, 0> (LDS-staged CSR tile, a row's gathers batched)", us_per_call=us, bytes=bts, achieved=bts / (us * 1e-6) / 1e9, unit="GB/s",
                frac=bts / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                note="frac is the WARM figure (back-to-back calls, 84 MB of operands in the 256 MB Infinity Cache); frac_cold is one call after 1 GB of other traffic -- what "
                     "the once-per-remesh call of the reference's loop meets",
                us_per_call_cold=us_cold, frac_cold=bts / (us_cold * 1e-6) / 1e9 / HBM_PEAK_GBS, traffic=pmc_traffic("ls::k_spmv<3, 0>", workload))


def cpu_baseline(v, f, cfg, u_np, seconds_cap=120.0):
    """Oracle direct solver timed on the host: factor once (reported, not counted), then re-solves."""
    from oracle import laplacian as ol, solve as osv
    r, c, val = ol.compute_matrix(v, f, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
    t0 = time.perf_counter()
    ds = osv.DirectSolver(r, c, val, v.shape[0])
    t_factor = time.perf_counter() - t0
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        x = ds.solve(u_np)
        times.append(time.perf_counter() - t0)
        if sum(times) > seconds_cap:
            break
    err = float(np.abs(x - v).max())
    t = float(np.median(times))
    return dict(value=1.0 / t, unit="solves/s", cores=1, kind="port",
                sample=f"oracle.DirectSolver (scipy SuperLU fp64, symmetric mode, 1 thread) on the same {v.shape[0]}-vertex system: "
                       f"1 factorisation ({t_factor:.1f} s, not counted) + {len(times)} timed 3-RHS solves, median {t * 1e3:.0f} ms; "
                       f"round-trip max-abs error {err:.1e}; host logical cores {os.cpu_count()}"), x


def describe(workload, cfg, V, nnz):
    """one-line description of a synthetic config (largesteps.synthetic.CONFIGS)"""
    mesh = {"cfg4_plane1m": "1000x1000 plane", "cfg5_plane4m": "2000x2000 plane", "scroll250k": "500x500 sheet rolled up 3 turns",
            "scroll10_250k": "500x500 sheet rolled up 10 turns", "folded250k": "500x500 sheet folded once, layers 1e-3 apart",
            "shells250k": "two concentric geodesic spheres 1e-3 apart", "scroll1m": "1000x1000 sheet rolled up 3 turns",
            "folded1m": "1000x1000 sheet folded once, layers 1e-3 apart"}.get(workload, "noisy geodesic sphere (stand-in mesh)")
    if cfg["alpha"] is not None:
        mat = f"M=(1-{cfg['alpha']:g})I+{cfg['alpha']:g}*L_{'cot' if cfg['cotan'] else 'uniform'}"
    else:
        mat = f"M=I+{cfg['lambda_']:g}*L_{'cot' if cfg['cotan'] else 'uniform'}"
    return f"{workload}: {mesh}, V={V}, nnz(M)={nnz}, {mat}, u=M v, k=3"


def run_single(args):
    from largesteps.geometry import compute_matrix
    from largesteps.parameterize import to_differential, from_differential
    from largesteps.solvers import CholeskySolver, ConjugateGradientSolver
    from largesteps import parameterize, synthetic, _native

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    v, f, cfg = synthetic.config_mesh(args.workload)
    lam = cfg["lambda_"] if cfg["lambda_"] is not None else 0.0
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    M = compute_matrix(tv, tf, lam, alpha=cfg["alpha"], cotan=cfg["cotan"])
    torch.cuda.synchronize()
    t_assemble = time.perf_counter() - t0
    u = to_differential(M, tv)
    V, nnz, k = v.shape[0], M._nnz(), 3
    method_name = "CG" if args.pcg else "Cholesky"        # 'Cholesky': cold start, reduction 1e-6 (the package default)

    # construction (factor once, NOT timed), then the cyclic garbage collector once and off -- BEFORE the warm-up steps, so that the W warm-up
    # solves run right in front of the timed region (a collection between them left the device idle for tens of milliseconds; the timed region
    # is a few milliseconds of host-driven launches, and timeit switches the collector off for the same reason)
    import gc
    solver = CholeskySolver(M) if method_name == "Cholesky" else ConjugateGradientSolver(M)
    parameterize.cache_put((id(M), method_name), solver, M)           # what from_differential's first call does (parameterize.py)
    gc.collect()
    gc.disable()
    if args.pcg:                                          # A/B: the Jacobi-PCG at the same cold-start / 1e-6 setting
        solver.rtol, solver.atol, solver.warm_start, solver.chebyshev = 1e-6, 0.0, False, False
    if args.block is not None:
        solver.set_option("block", args.block)
    if args.grid:
        solver.set_option("grid", args.grid)
    if args.check_every:
        solver.set_option("check_every", args.check_every)
    x = from_differential(M, u, method_name)              # (first use of the handle: not one of the W warm-up steps, not timed)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        x = from_differential(M, u, method_name)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = from_differential(M, u, method_name)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    info = dict(solver.last_info)
    ms = elapsed / args.steps * 1e3
    method = info["method"]
    if method == "nested-dissection":
        return report_direct(args, solver, M, u, x, tv, v, f, cfg, ms, t_assemble)

    # profiled pass right after the timed region, same workload: HIP events on the solve's own stream
    # (chebyshev: two events around the n back-to-back launches; pcg: events around every kernel)
    solver.set_option("profile", 1)
    prof = np.zeros(3)
    piters = 0
    for _ in range(max(1, min(args.steps, 5))):
        from_differential(M, u, method_name)
        a, b, c, it = solver.kernel_profile()
        prof += (a, b, c)
        piters += it
    solver.set_option("profile", 0)
    k_ms = prof / max(piters, 1)
    implicit = bool(getattr(solver, "implicit_values", False)) and method == "chebyshev"
    bts = algorithmic_bytes(V, nnz, k, info["iterations"], method, implicit)
    plan = getattr(solver, "patch_plan", None) if method == "chebyshev" else None
    launches_per_solve = info["iterations"]
    patch_note = None
    if plan is not None:
        # LDS-resident s-step kernel: one launch performs `depth` Chebyshev iterations. Its algorithmic bytes are the
        # bytes of the iterations it performs (SURVEY-style per-iteration figure x depth); what it really moves is far
        # less (the patch traffic below, and `traffic` from the PMC pass) -- that is the point of keeping patches in LDS.
        T = plan.table.astype(np.int64)
        patch_bytes = int((T[:, 3] * 2 * 4 * k + T[:, 2] * (4 * k + 4) + T[:, 4] * T[:, 2] * 2 + (T[:, 3] - T[:, 1]) * 4
                           + T[:, 1] * 2 * 4 * k).sum())
        launches_per_solve = -(-info["iterations"] // plan.depth)
        k_ms = k_ms * (piters / max(1, launches_per_solve * max(1, min(args.steps, 5))))     # per launch, not per iteration
        bts["k1"] = bts["iter"] * plan.depth
        patch_note = dict(patches=plan.n_patches, depth=plan.depth, max_local_vertices=plan.max_local,
                          redundancy=plan.redundancy, bytes_moved_per_launch_model=patch_bytes,
                          hbm_gbs_of_bytes_moved=patch_bytes / (float(k_ms[0]) * 1e-3) / 1e9,
                          # what bounds the kernel instead: LDS. Per step and computed row: W neighbour slots + own
                          # cur/prev read, one slot written, 4k bytes each; peak 128 B/clk/CU x 256 CUs x 2.4 GHz
                          lds_gbs=float((T[:, 8:8 + plan.depth].sum(axis=1) * (T[:, 4] + 3) * 4 * k).sum()) / (float(k_ms[0]) * 1e-3) / 1e9,
                          lds_peak_gbs=128 * 256 * 2.4,
                          note="achieved = bytes a one-iteration-per-launch kernel needs for the `depth` iterations of one "
                               "launch / launch time: an EFFECTIVE rate, it can exceed the HBM peak because ghost-layer "
                               "recomputation in LDS replaces HBM traffic")
    k1_gbs = bts["k1"] / (k_ms[0] * 1e-3) / 1e9
    err = float((x - tv).abs().max())
    if method == "chebyshev":
        solver_desc = "HIP Chebyshev-accelerated Jacobi iteration (1 kernel/iteration, SELL-64, a-priori iteration count" + \
                      (", implicit uniform-Laplacian values)" if implicit else ")")
        kernel_desc = ("k_cheb_uniform<3,512>" if implicit else "k_cheb<3,512>") + \
                      " (x_{k+1} = x_k + c1 (x_k - x_{k-1}) + c2 D^-1 (b - M x_k), SELL-64)"
        kernel_us = dict(k_cheb=k_ms[0] * 1e3)
        if plan is not None:
            solver_desc = (f"HIP Chebyshev-accelerated Jacobi iteration, LDS-resident on {plan.n_patches} mesh patches: "
                           f"{plan.depth} iterations per launch (ghost layers recomputed), implicit uniform-Laplacian values")
            kernel_desc = f"k_patch_cheb<3,1024> ({plan.depth} Chebyshev steps per launch on LDS-resident patches)"
            kernel_us = dict(k_patch_cheb=k_ms[0] * 1e3)
    else:
        solver_desc = "HIP Jacobi-PCG (3 kernels/iteration, SELL-64)"
        kernel_desc = "k_spmv_dot<3> (K1: Ap = M p on SELL-64, partial p.Ap)"
        kernel_us = dict(k1_spmv_dot=k_ms[0] * 1e3, k2_update=k_ms[1] * 1e3, k3_direction=k_ms[2] * 1e3)
    out = dict(
        metric="from_differential_solves_per_sec", value=1e3 / ms, unit="solves/s", n_gpus=1, steps=args.steps,
        warmup=args.warmup, ms_per_step=ms, higher_is_better=True, scaling="strong", vs_baseline=None, dtype="f32",
        data="synthetic",
        config=dict(workload=describe(args.workload, cfg, V, nnz) + ", cold start, residual reduction 1e-6", solver=solver_desc, method=method,
                    iterations=info["iterations"], converged=info["converged"],
                    tolerance=dict(kind="relative residual ||r|| / ||b|| per column at which the iteration stops (SURVEY 8d)", rel=1e-6,
                                   measured=max(float(r / b) for r, b in zip(info["rnorm"], info["bnorm"])), met=bool(info["converged"])),
                    rel_residual=[float(r / b) for r, b in zip(info["rnorm"], info["bnorm"])],
                    max_abs_err_vs_v=err, assemble_ms=t_assemble * 1e3,
                    solve_bytes=bts["solve"], solve_gbs=bts["solve"] / (ms * 1e-3) / 1e9,
                    solve_frac_of_8tbs=bts["solve"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    kernel_us=kernel_us, device=torch.cuda.get_device_name(0)),
        roofline=dict(bound="hbm", kernel=kernel_desc, achieved=k1_gbs,
                      peak=HBM_PEAK_GBS, unit="GB/s", frac=k1_gbs / HBM_PEAK_GBS, frac_of_achievable=k1_gbs / HBM_ACHIEVABLE_GBS,
                      bytes_per_launch=bts["k1"], avg_launch_us=k_ms[0] * 1e3,
                      launches_timed=int(launches_per_solve * max(1, min(args.steps, 5))) if plan is not None else int(piters),
                      traffic=pmc_traffic("ls::k_patch_cheb<3" if plan is not None else
                                          ("ls::k_cheb_uniform<3, 512, false>" if implicit else
                                           ("ls::k_cheb<3, 512, false>" if method == "chebyshev" else "ls::k_spmv_dot<3")), args.workload),
                      patch=patch_note),
    )
    if not args.no_cpu_baseline:
        base, x_oracle = cpu_baseline(v, f, cfg, u.cpu().numpy())
        out["cpu_baseline"] = base
        out["config"]["max_abs_err_vs_oracle"] = float(np.abs(x.cpu().numpy() - x_oracle).max())
        out["config"]["max_abs_oracle"] = float(np.abs(x_oracle).max())
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)


def extra_baselines(args):
    """SURVEY 8(d) baselines B2 / B3, outside any timed region of the product:
      B2  the reference's own ConjugateGradientSolver algorithm (largesteps/solvers.py:58-126; oracle.solve.reference_cg, fp32,
          column by column, stop at ||r|| <= 1e-5, no preconditioner) on the host at config 2 (70k vertices);
      B3  the same algorithm written with stock torch ops on the HIP device ("naive ROCm port": torch.sparse.mm + torch
          reductions, one column at a time like the reference) on the benchmark's own system."""
    from oracle import laplacian as ol, solve as osv
    from largesteps import synthetic
    out = {}
    v2, f2, c2 = synthetic.config_mesh("cfg2_bunny70k")
    r, c, val = ol.compute_matrix(v2, f2, c2["lambda_"], alpha=c2["alpha"], cotan=c2["cotan"])
    u2 = osv.to_differential(r, c, val, v2).astype(np.float32)
    t0 = time.perf_counter()
    x2, its = osv.reference_cg(r, c, val, u2)
    t = time.perf_counter() - t0
    out["B2_reference_cg_cpu_cfg2"] = dict(solves_per_s=1.0 / t, ms_per_solve=t * 1e3, iterations_per_column=[int(i) for i in its], cores=1,
                                           max_abs_err_vs_v=float(np.abs(x2 - v2).max()),
                                           note="reference CG algorithm, numpy/scipy fp32 on the host, 70k-vertex config")
    dev = torch.device("cuda", 0)
    v, f, cfg = synthetic.config_mesh(args.workload)
    r, c, val = ol.compute_matrix(v, f, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
    V = v.shape[0]
    Mt = torch.sparse_coo_tensor(torch.from_numpy(np.stack([r, c])).to(dev), torch.from_numpy(val.astype(np.float32)).to(dev), (V, V)).coalesce()
    tv = torch.from_numpy(v).to(dev)
    b = torch.sparse.mm(Mt, tv)

    def torch_cg(bcol):          # solvers.py:58-84, stock torch ops
        x = torch.zeros_like(bcol)
        rr = -bcol.clone()
        p = -rr
        rn = torch.norm(rr)
        k = 0
        while float(rn) > 1e-5 and k < 20000:
            Ap = torch.sparse.mm(Mt, p.unsqueeze(1)).squeeze(1)
            r2 = rn * rn
            alpha = r2 / (p * Ap).sum()
            x = x + alpha * p
            rr = rr + alpha * Ap
            rn = torch.norm(rr)
            p = -rr + (rn * rn / r2) * p
            k += 1
        return x, k
    torch_cg(b[:, 0].contiguous())             # warm-up (allocator, kernel load)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cols = [torch_cg(b[:, j].contiguous()) for j in range(3)]
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    xs = torch.stack([c_[0] for c_ in cols], 1)
    out["B3_torch_cg_hip_device"] = dict(solves_per_s=1.0 / t, ms_per_solve=t * 1e3, iterations_per_column=[int(c_[1]) for c_ in cols],
                                         max_abs_err_vs_v=float((xs - tv).abs().max()),
                                         note=f"reference CG algorithm in stock torch ops on the MI355X, {args.workload}")
    return out


def report_direct(args, solver, M, u, x, tv, v, f, cfg, ms, t_assemble):
    """JSON line for the factor-once / re-solve direct solver (largesteps.solvers.NestedDissectionSolver)."""
    from largesteps.parameterize import to_differential
    V, nnz, k = v.shape[0], M._nnz(), 3
    # profiled passes right after the timed region (same workload, HIP events on the solve's own stream):
    #   "profile" 1: events around the up sweep and the down sweep;  "profile" 3: an event in front of every launch
    n_prof = max(1, min(args.steps, 5))
    solver.set_option("profile", 1)
    up_ms = down_ms = 0.0
    for _ in range(n_prof):
        solver.solve(u)
        inf = solver.info()
        up_ms += inf["up_ms"] / n_prof
        down_ms += inf["down_ms"] / n_prof
    solver.set_option("profile", 3)
    table = None
    for _ in range(n_prof):
        solver.solve(u)
        lp = solver.launch_profile()
        if table is None:
            table = [dict(levels=list(r["levels"]), sweep=r["sweep"], factor_bytes=4 * r["words"], us=0.0) for r in lp]
        for row, r in zip(table, lp):
            row["us"] += r["ms"] * 1e3 / n_prof
    solver.set_option("profile", 0)
    # the vectors' share of a launch: per sweep the b / b' / x rows of its levels' vertices and the boundary vectors of their nodes
    # (the per-sweep total below, 4k (2V + 3 n_bnd) + 4V, split by level: ls_direct_level_rows)
    lv_rows, lv_bnd = solver.level_rows()
    idx_up, idx_down = solver.level_index_bytes()
    tier_bal = solver.tier_balance() if hasattr(solver, "tier_balance") else None        # (before the handle is closed below)
    for row in table:
        lo, hi = row["levels"]
        # static index data of the launch besides perm: tile / item records, masks, parent positions, pull lists, push lists
        row["index_bytes"] = (sum(idx_up[max(lo, 0):hi + 1]) if row["sweep"] != "down" else 0) + (sum(idx_down[max(lo, 0):hi + 1]) if row["sweep"] != "up" else 0)
        nv, nb_ = sum(lv_rows[max(lo, 0):hi + 1]), sum(lv_bnd[max(lo, 0):hi + 1])
        both = 2 if row["sweep"] == "both" else 1
        # strict: the factor words + per sweep one read and one write of the level's k-column rows (b -> b', b' -> x) and its index word;
        # "solver vectors": the hand-over of boundary values between tree levels (written by one level, read by the next, re-read when
        # pushed down: 3 passes over n_bnd k-vectors) -- this solver's own structure, reported beside the strict figure, not inside it
        row["vector_bytes"] = both * (4 * 3 * 2 * nv + 4 * nv)
        row["solver_vector_bytes"] = both * (4 * 3 * 3 * nb_)
        row["bytes"] = row["factor_bytes"] + row["vector_bytes"]
        row["tb_per_s"] = row["bytes"] / (row["us"] * 1e-6) / 1e12 if row["us"] > 0 else None
        row["frac_of_8tbs"] = row["tb_per_s"] / 8.0 if row["tb_per_s"] else None
    n_down = sum(1 for r in table if r["sweep"] == "down")
    n_up = sum(1 for r in table if r["sweep"] == "up")
    n_bnd = inf["n_bnd"]
    # algorithmic bytes: the fp32 factor data each sweep reads (dense nodes: W in both sweeps, Finv in the down sweep; leaves:
    # one packed triangle per sweep + their sparse block) + the vectors once per sweep (b / b' / x rows, boundary vectors)
    # ... + b / b' / x rows once per sweep. The boundary hand-over between tree levels (3 passes over n_bnd k-vectors per sweep) is the
    # solver's own structure: listed as `solver_vector_bytes`, NOT part of the algorithmic bytes the fractions below are quoted on
    vec_bytes = 4 * k * 2 * V + 4 * V
    hand_over_bytes = 4 * k * 3 * n_bnd
    up_bytes = 4 * inf["words_up"] + vec_bytes
    down_bytes = 4 * inf["words_down"] + vec_bytes
    solve_bytes = up_bytes + down_bytes
    r = to_differential(M, x) - u
    rel_res = [float(a / b) for a, b in zip(r.norm(dim=0).tolist(), u.norm(dim=0).tolist())]
    grp_bytes, grp_ms, grp_n = down_bytes, down_ms, n_down
    grp_name = (f"down sweep: k_nd_down_b<3> x {n_down - 1} + k_nd_tier<3, false> ({n_down} launches: x_s = Finv b'_s - W^T x_bnd "
                f"per upper tree level, then the deepest {inf['tier_levels']} levels in one launch)")
    grp_prefixes = direct_group_prefixes()
    grp_gbs = grp_bytes / (grp_ms * 1e-3) / 1e9
    traffic, traffic_n = pmc_traffic_group(grp_prefixes, args.workload)
    tm = solver.timings
    assembly = assembly_entry(tv, torch.from_numpy(f).to(tv.device), cfg, V, f.shape[0], nnz, args.workload)
    # the constructor once more, after everything above (not timed, not used): the solver above was the FIRST one this process built and
    # paid the process' one-off costs (kernel code objects loaded on first launch, host thread pool, fresh heap); a remesh loop pays this
    # cycle (scripts/main.py:137-169: the old matrix and its solver are gone when the new one is built): the measured solver is closed first,
    # so that -- as in that loop -- its factor arrays are in the library's pool when the next construction asks for them
    repeat_seconds = steady_seconds = None
    steady_cycles = []
    if not args.no_extra_baselines:
        try:
            from largesteps.solvers import NestedDissectionSolver
            if hasattr(solver, "close"):
                solver.close()
            s2 = NestedDissectionSolver(M)
            repeat_seconds = s2.build_seconds
            s2.close()
            del s2
        except Exception:
            repeat_seconds = None
        try:
            # and the steady state of a remesh loop: six further constructions (the process' first four still meet pools, allocator blocks
            # and host pages they touch for the first time: profiles/r06_f4_remesh_cycle.txt); steady = their MEDIAN, all six are listed
            for _ in range(6):
                s3 = NestedDissectionSolver(M)
                steady_cycles.append(s3.build_seconds)
                s3.close()
                del s3
            steady_seconds = sorted(steady_cycles)[len(steady_cycles) // 2]
        except Exception:
            steady_seconds = None
    out = dict(
        metric="from_differential_solves_per_sec", value=1e3 / ms, unit="solves/s", n_gpus=1, steps=args.steps,
        warmup=args.warmup, ms_per_step=ms, higher_is_better=True, scaling="strong", vs_baseline=None, dtype="f32",
        data="synthetic",
        config=dict(workload=describe(args.workload, cfg, V, nnz) + ", factor once (not timed), re-solve timed",
                    solver=(f"HIP nested-dissection multifrontal direct solver behind ls_direct_factor: {inf['levels']} tree levels, "
                            f"symbolic analysis (bisection rounds on the device, tree and index lists on host threads), fp64 factorisation with hand-written kernels (once), fp32 factor "
                            f"{inf['factor_entries'] / 1e6:.1f} M words per solve; re-solve = {inf['launches']} launches (one per upper "
                            f"level and sweep, one per sweep for the deepest {inf['tier_levels']} levels), no atomics"),
                    method="nested-dissection", iterations=0,
                    # what the timed result meets: a direct solve has no stopping rule; its accuracy statement is the forward error
                    # against the fp64 oracle (filled in below from the cpu_baseline leg's solution of the same system)
                    tolerance=dict(kind="forward error vs the fp64 oracle's solution of the same system, max-abs relative to max |x|",
                                   rel=1e-4, measured=None, met=None),
                    rel_residual=rel_res,
                    rel_residual_note="measured after the timed region, for information: the direct solver never looks at a residual "
                                      "(SURVEY 8d's 1e-6 is the ITERATIONS' stopping rule: --iterative / --pcg)",
                    max_abs_err_vs_v=float((x - tv).abs().max()), assemble_ms=t_assemble * 1e3,
                    dissection=getattr(solver, "plan_quality", None),
                    # factor words of the tier's subtrees, one workgroup (one CU) each: the heaviest one is the tier launches' time
                    tier_balance=tier_bal, tier_workgroups=inf["tier_workgroups"],
                    factor_seconds=getattr(solver, "build_seconds", None), factor_seconds_second_construction=repeat_seconds, factor_seconds_steady=steady_seconds,
                    factor_seconds_cycles=steady_cycles,
                    factor_seconds_note="first: the process' first construction (code objects loaded, thread pool started, fresh heap); second: right after the measured "
                                        "solver was closed (its buffers in the pool); cycles: six further constructions back to back; steady: their median (the first "
                                        "four constructions of a process are slower than its steady state); inside a remesh loop: tools/bench_remesh.py, tools/ctor_in_loop.py",
                    factor_stages_seconds=dict(symbolic_analysis=tm["plan_seconds"], layouts_host=tm["table_seconds"], numeric_device_and_solve_tables=tm["factor_seconds"]),
                    solve_bytes=solve_bytes, solve_gbs=solve_bytes / (ms * 1e-3) / 1e9,
                    solve_frac_of_8tbs=solve_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    kernel_us=dict(up_sweep=up_ms * 1e3, down_sweep=down_ms * 1e3,
                                   up_launches=n_up, down_launches=n_down),
                    # every launch of one solve: tree levels it runs, the factor bytes it reads (ls_direct_level_words), its
                    # duration between two HIP events on the solve's stream ("profile" 3 pass; events between the launches
                    # add ~1 us each, so the rows sum to a little more than ms_per_step), bytes / time
                    launches=table, vector_bytes_per_sweep=vec_bytes, solver_vector_bytes_per_sweep=hand_over_bytes,
                    assemble=assembly, to_differential=spmv_entry(M, tv, V, nnz, args.workload),
                    device=torch.cuda.get_device_name(0)),
        roofline=dict(bound="hbm", kernel=grp_name,
                      achieved=grp_gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=grp_gbs / HBM_PEAK_GBS,
                      frac_of_achievable=grp_gbs / HBM_ACHIEVABLE_GBS, bytes_per_launch=grp_bytes / grp_n,
                      avg_launch_us=grp_ms * 1e3 / grp_n, launches_timed=grp_n * n_prof,
                      traffic=traffic,
                      traffic_source=(f"profiles/{PMC_FILE}: FETCH_SIZE x 2 + WRITE_SIZE of separate rocprofv3 --pmc passes of this command, "
                                      f"mean over the {traffic_n} launches of the SAME kernel group, per launch like bytes_per_launch; "
                                      f"NOT measured in this run") if traffic is not None else None,
                      note="every upper tree level is one dependent launch (T_stream + ~4.5 us of launch boundary, first-byte latency and "
                           "reduction tail); inside the tier launches the leaf rounds stream at ~5.5 TB/s counted in HBM-side bytes (round 4: "
                           "per-wave clock stamps, DESIGN.md section 2.3), the tier's loss against that rate is its start (three dependent "
                           "round trips and the burst of 4096 first triangles) and its two dense levels"),
    )
    if not args.no_cpu_baseline:
        base, x_oracle = cpu_baseline(v, f, cfg, u.cpu().numpy())
        out["cpu_baseline"] = base
        # parity of the timed solve's result with the oracle's fp64 solution of the same system (tolerance: 1e-4 relative)
        out["config"]["max_abs_err_vs_oracle"] = float(np.abs(x.cpu().numpy() - x_oracle).max())
        out["config"]["max_abs_oracle"] = float(np.abs(x_oracle).max())
        tol = out["config"]["tolerance"]
        tol["measured"] = out["config"]["max_abs_err_vs_oracle"] / out["config"]["max_abs_oracle"]
        tol["met"] = bool(tol["measured"] <= tol["rel"])
        if not args.no_extra_baselines:
            out["cpu_baseline"]["extra"] = extra_baselines(args)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)


def run_distributed(args):
    import torch.distributed as dist
    from largesteps import distributed as lsd
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # LS_DIST_LOOPBACK=1: functional smoke test on a 1-GPU box -- every rank uses cuda:0 and gloo moves the data
    # (RCCL refuses two ranks on one device). The numbers of such a run mean nothing.
    loopback = os.environ.get("LS_DIST_LOOPBACK") == "1"
    if loopback:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if loopback:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)
    out = lsd.bench_sharded(args.workload, dev, steps=args.steps, warmup=args.warmup, shard=args.shard)
    if rank == 0:
        ms = out["ms_per_step"]
        if out.get("solve_bytes") is not None:
            bts = dict(solve=out["solve_bytes"])
        else:
            bts = algorithmic_bytes(out["V"], out["nnz"], 3, out["iterations"], "chebyshev" if out["method"] in ("chebyshev", "iterative") else "pcg")
        n_rep = out.get("replicas", 1)              # replicas: every rank completes one solve per step
        res = dict(
            metric="from_differential_solves_per_sec", value=n_rep * 1e3 / ms, unit="solves/s", n_gpus=world, steps=args.steps,
            warmup=args.warmup, ms_per_step=ms, higher_is_better=True, scaling="weak" if n_rep > 1 else "strong", vs_baseline=None, dtype="f32",
            data="synthetic",
            config=dict(workload=f"{args.workload}: V={out['V']}, nnz(M)={out['nnz']}, k=3, every solve from b alone, "
                                 f"sharded by {out['shard']} over {world} ranks", solver=out["solver"], iterations=out["iterations"],
                        # who took part: torch.distributed's view, the library's RCCL communicator's own (ncclCommCount), the devices, and
                        # per rank the three pieces of a sharded solve from HIP events (a profiled pass after the timed region)
                        ranks=dict(world_size=dist.get_world_size(), backend=dist.get_backend(), transport="loopback: every rank on cuda:0, gloo moves the data -- "
                                   "functional run, the timings mean nothing" if loopback else "RCCL over xGMI, one process per GPU",
                                   communicator=out.get("communicator"), devices=out.get("devices"), per_rank=out.get("per_rank")),
                        model=shard_model(args.workload, world) if out["shard"] == "vertex" else None,
                        # the sharded x against one unsharded solve of the same system, the owners' cover, the communicator's world size
                        # (largesteps/distributed.py bench_sharded); a failed check fails the run: rc 1, no throughput is reported as valid
                        shard_check=out.get("shard_check"),
                        converged=out["converged"], max_abs_err_vs_v=out["err"], halo_vertices=out["halo"], method=out["method"],
                        halo_depth=out["depth"], rows_per_rank=out["rows_per_rank"],
                        solve_bytes=bts["solve"], solve_gbs=bts["solve"] / (ms * 1e-3) / 1e9),
            # N > 1: whole sharded solve (all kernels + collectives) against the aggregate HBM peak of the N GPUs
            roofline=dict(bound="hbm", kernel="whole sharded solve (all kernels on every rank + the collectives)",
                          achieved=bts["solve"] / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS * world, unit="GB/s",
                          frac=bts["solve"] / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), traffic=None),
            cpu_baseline=None,
        )
        print(json.dumps(res), flush=True)
    bad = out.get("shard_check") is not None and not out["shard_check"]["ok"]
    dist.barrier()
    dist.destroy_process_group()
    if bad:
        if rank == 0:
            print(f"bench.py: the sharded solve FAILED its self-check: {out['shard_check']}", file=sys.stderr, flush=True)
        sys.exit(1)


def launch_ranks(args):
    """`python bench.py --gpus N` started plainly (no RANK in the environment): start the N ranks ourselves, exactly as the driver's
    launcher would -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    <the same arguments>` -- and pass rank 0's JSON line through. With fewer than N visible devices the ranks share cuda:0 and gloo moves
    the data (LS_DIST_LOOPBACK=1: a functional run; the line says so in config.ranks.transport and its numbers mean nothing)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() < args.gpus:
        env["LS_DIST_LOOPBACK"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# kernels of ONE rank per solve and the part of them above the cut level, measured on one MI355X with the plan that rank gets
# (tools/shard_rank_time.py -> profiles/r06_run1_shard_rank_kernel_times.txt); the whole-job prediction adds one latency-bound all-reduce
SHARD_MODEL_FILE = "r06_run1_shard_rank_kernel_times.txt"
SHARD_MODEL_US = {
    "cfg4_plane1m": {1: (191.5, 0.0), 2: (147.4, 12.6), 4: (125.2, 19.0), 8: (121.6, 55.6)},
    "cfg5_plane4m": {1: (711.2, 0.0), 2: (405.3, 21.5), 4: (284.5, 21.5), 8: (252.1, 71.9)},
}
SHARD_MODEL_COLLECTIVE_US = (15.0, 30.0)


def shard_model(workload, world):
    """what DESIGN.md section 5 predicts for this N, so that a SCALE record is a test of the model"""
    row = SHARD_MODEL_US.get(workload, {}).get(world)
    if row is None:
        return None
    lo, hi = (0.0, 0.0) if world == 1 else SHARD_MODEL_COLLECTIVE_US
    return dict(kernel_us_per_rank=row[0], of_which_above_the_cut_us=row[1], collective_us_assumed=[lo, hi],
                predicted_ms_per_step=[(row[0] + lo) * 1e-3, (row[0] + hi) * 1e-3],
                source=f"profiles/{SHARD_MODEL_FILE} (one rank's kernels on one MI355X) + one RCCL all-reduce of the exchange region")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=WORKLOAD)
    ap.add_argument("--block", type=int, default=None, help="threads per PCG workgroup (256 or 1024; default: auto)")
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--check-every", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-baselines", action="store_true", help="skip the B2 / B3 legs (reference CG on the host / in stock torch ops)")
    ap.add_argument("--pcg", action="store_true", help="time the Jacobi-PCG instead of the default (Chebyshev) solver")
    ap.add_argument("--iterative", action="store_true",
                    help="'Cholesky' through the Chebyshev-Jacobi iteration instead of the nested-dissection direct solver (A/B)")
    ap.add_argument("--shard", default="auto", choices=["auto", "vertex", "columns", "halo", "replicas"],
                    help="N > 1: vertex blocks = subtrees of the direct solver's elimination tree (default), right-hand-side columns across "
                         "ranks, vertex blocks of the iteration with halo exchange, or independent replicas (weak scaling)")
    args = ap.parse_args()
    if args.iterative:
        os.environ["LARGESTEPS_NO_DIRECT"] = "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args))
    if args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        run_distributed(args)
    else:
        run_single(args)


if __name__ == "__main__":
    main()
