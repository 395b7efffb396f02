"""what ONE rank of the subtree-sharded direct solver computes per solve, timed on one GPU without any communication:
   the handle of rank r of N (shard = (r, N)) runs part 0 (its subtrees upwards), the exchange buffer is copied in place of
   the all-reduce, part 1 (replicated levels, its subtrees downwards). The all-reduce itself (94 - 375 KiB at 1M) is NOT in it.
   python tools/shard_rank_time.py [workload] [solves]"""
import ctypes, os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import torch
from largesteps import synthetic, _native
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential
from largesteps.solvers import NestedDissectionSolver
workload = sys.argv[1] if len(sys.argv) > 1 else "cfg4_plane1m"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
v, f, cfg = synthetic.config_mesh(workload)
tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
u = to_differential(M, tv)
lib = _native.lib()
for N in (1, 2, 4, 8):
    for r in sorted({0, N - 1}):
        s = NestedDissectionSolver(M, shard=(r, N))
        h = s._direct._h
        per_col, cut = ctypes.c_int64(0), ctypes.c_int(0)
        _native.check(lib.ls_direct_shard_info(h, None, None, ctypes.byref(cut), ctypes.byref(per_col), None))
        ex = torch.zeros(max(1, per_col.value * 3), dtype=torch.float32, device=dev)
        x = torch.empty_like(u)
        st = _native.stream_of(dev)
        def solve():
            if N == 1:
                _native.check(lib.ls_direct_solve(h, _native.ptr(u), _native.ptr(x), 3, st))
            else:
                _native.check(lib.ls_direct_solve_part(h, _native.ptr(u), _native.ptr(x), 3, 0, _native.ptr(ex), st))
                _native.check(lib.ls_direct_solve_part(h, _native.ptr(u), _native.ptr(x), 3, 1, _native.ptr(ex), st))
        for _ in range(5): solve()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): solve()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        # where the time goes: an event in front of every launch ("profile" 3; the events add ~1 us per launch) -- the levels ABOVE the cut
        # run on every rank alike (replicated), the rest is the rank's own share
        s.set_option("profile", 3)
        rep_us = own_us = 0.0
        rows = []
        for part in ((None,) if N == 1 else (0, 1)):
            if part is None:
                _native.check(lib.ls_direct_solve(h, _native.ptr(u), _native.ptr(x), 3, st))
            else:
                _native.check(lib.ls_direct_solve_part(h, _native.ptr(u), _native.ptr(x), 3, part, _native.ptr(ex), st))
            for row in s.launch_profile():
                rows.append(row)
                if N > 1 and row["levels"][1] < cut.value:
                    rep_us += row["ms"] * 1e3
                else:
                    own_us += row["ms"] * 1e3
        s.set_option("profile", 0)
        inf = s.info()
        print(f"{workload}: rank {r} of {N}: {ms * 1e3:7.1f} us of kernels per solve (cut level {cut.value}, exchange {per_col.value * 3 * 4 / 1024:.0f} KiB, "
              f"tier workgroups {inf['tier_workgroups']}, {len(rows)} launches; by events: levels above the cut {rep_us:6.1f} us, own subtrees {own_us:6.1f} us)", flush=True)
        del s
