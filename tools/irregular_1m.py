"""The re-solve on 1M-vertex meshes that are NOT planes, 4-wave / 3-level tier against 16-wave / 4-level tier (round 6):
python tools/irregular_1m.py [solves] [--sizes]   -> profiles/r06_tier16_irregular.txt"""
import os, sys, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import numpy as np, torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential
from largesteps.solvers import NestedDissectionSolver

n_solves = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200
dev = torch.device("cuda:0")


def run(name, v, f, cfg, variants):
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
    u = to_differential(M, tv)
    for waves, ordering in variants:
        t0 = time.perf_counter()
        s = NestedDissectionSolver(M, tier_waves=waves, ordering=ordering)
        build = time.perf_counter() - t0
        for _ in range(5): x = s.solve(u)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n_solves): x = s.solve(u)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / n_solves * 1e6
        inf, q = s.info(), s.plan_quality
        err = float((x - tv).abs().max())
        s.set_option("profile", 3)
        s.solve(u); s.solve(u)
        prof = s.launch_profile()
        tier = [p for p in prof if p["levels"][1] == inf["levels"] - 1]
        tier_us = " + ".join(f"{p['ms'] * 1e3:.1f}" for p in tier)
        print(f"{name:24s} V {v.shape[0]:8d} waves {waves:2d} order {q['ordering']:12s} words/V {q['words_per_vertex']:6.1f} spread {q['spread']:.2f} "
              f"levels {inf['levels']} tier {inf['tier_levels']} ({inf['tier_workgroups']} wg) launches {inf['launches']:2d} | {us:7.1f} us/solve | tier {tier_us} us | "
              f"factor {inf['factor_entries'] * 4e-6:6.1f} MB | build {build * 1e3:6.1f} ms | round-trip err {err:.1e}", flush=True)
        if "--table" in sys.argv:
            print("      " + " | ".join(f"L{p['levels'][0]}-{p['levels'][1]} {p['sweep']} {p['ms'] * 1e3:.1f}" for p in prof), " balance", s.tier_balance()["max_over_mean"], flush=True)
        assert err < 1e-4, err
        s.close(); del s


if "--sizes" in sys.argv:
    for n in (283, 316, 380, 447):
        v, f = synthetic.icosphere(n)
        v = synthetic.perturb(v, radial=0.05, tangential=0.25, edge=1.2 / n, seed=0)
        run(f"sphere n={n} uniform", v, f, dict(lambda_=50.0, alpha=None, cotan=False), [(16, None), (4, None)])
    for n in (900, 1000, 1200, 1414):
        v, f = synthetic.scroll(n, 3)
        run(f"scroll n={n}", v, f, dict(lambda_=50.0, alpha=None, cotan=False), [(16, None), (4, None)])
elif "--quick" in sys.argv:
    for rep in range(2):
        for name in ("cfg4_plane1m", "cfg4b_sphere1m_uniform", "scroll1m"):
            v, f, cfg = synthetic.config_mesh(name)
            run(name, v, f, cfg, [(0, None)])
else:
    for name in ("cfg4_plane1m", "cfg4b_sphere1m", "cfg4b_sphere1m_uniform", "scroll1m", "folded1m"):
        v, f, cfg = synthetic.config_mesh(name)
        variants = [(16, None), (4, None)]
        if "sphere" in name: variants += [(16, "longest-axis"), (16, "trial-cuts")]
        run(name, v, f, cfg, variants)
