#!/usr/bin/env python3
"""A/B sweep of the PCG knobs on one GPU: matrix access variant x grid size, per-kernel HIP-event times.
Usage: python tools/sweep.py [workload ...]   (default: cfg4_plane1m cfg2_bunny70k cfg3_dragon250k)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "large-steps-pytorch_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from largesteps.geometry import compute_matrix  # noqa: E402
from largesteps.parameterize import to_differential  # noqa: E402
from largesteps.solvers import PCGSolver  # noqa: E402
from largesteps import synthetic, _native  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    names = sys.argv[1:] or ["cfg4_plane1m", "cfg2_bunny70k", "cfg3_dragon250k"]
    for name in names:
        v, f, c = synthetic.config_mesh(name)
        tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
        t_asm = timeit(lambda: compute_matrix(tv, tf, c["lambda_"] or 0.0, alpha=c["alpha"], cotan=c["cotan"]), 3)
        M = compute_matrix(tv, tf, c["lambda_"] or 0.0, alpha=c["alpha"], cotan=c["cotan"])
        u = to_differential(M, tv)
        csr = _native.csr_of(M)
        V, nnz = csr.V, csr.nnz
        print(f"== {name}: V={V} nnz={nnz} assemble={t_asm:.2f} ms")
        for variant in (0, 1):
            t = timeit(lambda: _native.spmv(csr, tv, variant), 20)
            print(f"   to_differential variant {variant}: {t * 1e3:8.1f} us  {(8 * nnz + 28 * V) / t / 1e6:8.1f} GB/s")
        s = PCGSolver(M, rtol=1e-6)
        for algo in (0,):
            for block, grids in ((512, (512, 1024)), (256, (512, 1024)), (1024, (256, 512))):
                for grid in grids:
                    s.set_option("block", block); s.set_option("grid", grid); s.set_option("check_every", 16)
                    ms = timeit(lambda: s.solve(u), 5)
                    it = s.last_info["iterations"]
                    s.set_option("profile", 1)
                    s.solve(u)
                    k1, k2, k3, pit = s.kernel_profile()
                    s.set_option("profile", 0)
                    err = float((s.solve(u) - tv).abs().max())
                    print(f"   algo {algo} block {block:4d} grid {grid:5d}: {ms:8.3f} ms/solve  iters {it:4d}  {ms * 1e3 / it:7.2f} us/iter | "
                          f"K1 {k1 / pit * 1e3:6.2f} us  K2 {k2 / pit * 1e3:6.2f} us  K3 {k3 / pit * 1e3:6.2f} us (event-bracketed) | max|x-v| {err:.1e}", flush=True)
        s.set_option("block", 0); s.set_option("grid", 0)
        sc = PCGSolver(M, rtol=1e-6, chebyshev=True)
        for block, grids in ((256, (512, 1024)), (512, (512, 1024)), (1024, (256, 512))):
            for grid in grids:
                sc.set_option("block", block); sc.set_option("grid", grid)
                ms = timeit(lambda: sc.solve(u), 5)
                xs = sc.solve(u)
                inf = sc.last_info
                print(f"   chebyshev block {block:4d} grid {grid:5d}: {ms:8.3f} ms/solve  iters {inf['iterations']:4d} ({inf['method']})  {ms * 1e3 / max(inf['iterations'], 1):7.2f} us/iter | "
                      f"max|x-v| {float((xs - tv).abs().max()):.1e}  true rel residual {[f'{r / b:.1e}' for r, b in zip(inf['rnorm'], inf['bnorm'])]}", flush=True)
        sc.set_option("block", 0); sc.set_option("grid", 0)
        for gr in (0, 1):
            sc.set_option("graph", gr)
            sc.set_option("patch", 0)
            print(f"   chebyshev one-step kernel graph={gr}: {timeit(lambda: sc.solve(u), 10):8.3f} ms/solve")
        sc.set_option("patch", 1)
        if sc.patch_plan is not None:
            pl = sc.patch_plan
            ms = timeit(lambda: sc.solve(u), 10)
            xs = sc.solve(u)
            print(f"   chebyshev PATCH kernel ({pl.n_patches} patches, depth {pl.depth}, max_local {pl.max_local}, redundancy {pl.redundancy:.2f}): "
                  f"{ms:8.3f} ms/solve  iters {sc.last_info['iterations']}  max|x-v| {float((xs - tv).abs().max()):.1e}")
        for rt in (1e-4, 1e-5, 1e-7):
            sc.rtol = rt
            xs = sc.solve(u)
            print(f"   chebyshev rtol {rt:g}: iters {sc.last_info['iterations']}  max|x-v| {float((xs - tv).abs().max()):.1e}")
        sc.rtol = 1e-6
        sc.warm_start = True
        sc.solve(u)
        ms = timeit(lambda: sc.solve(u * 1.001), 5)
        print(f"   chebyshev warm start (u*1.001): {ms:8.3f} ms/solve iters {sc.last_info['iterations']}")
        del sc
        s.set_option("block", 0); s.set_option("grid", 0)
        x = s.solve(u)
        print(f"   max|x - v| = {float((x - tv).abs().max()):.2e}  rel residual {[r / b for r, b in zip(s.last_info['rnorm'], s.last_info['bnorm'])]}")
        for ce in (4, 8, 32, 64):
            s.set_option("check_every", ce)
            print(f"   check_every {ce:3d}: {timeit(lambda: s.solve(u), 5):8.3f} ms/solve")
        # warm start from a slightly perturbed right-hand side (the optimisation loop's situation)
        s.set_option("check_every", 16)
        s.warm_start = True
        s.solve(u)
        ms = timeit(lambda: s.solve(u * 1.001), 5)
        print(f"   warm start (u*1.001): {ms:8.3f} ms/solve iters {s.last_info['iterations']}")
        del s, M


if __name__ == "__main__":
    main()
