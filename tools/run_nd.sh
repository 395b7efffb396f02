#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python - <<'PY'
import sys, time, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "large-steps-pytorch_amd")]
import numpy as np, torch
from largesteps import synthetic
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential
from largesteps.solvers import NestedDissectionSolver, CholeskySolver
from oracle import solve as osv
dev = torch.device("cuda:0")
for name, (v, f), kw in [("plane40", synthetic.plane(40), dict(lambda_=30.0)),
                         ("ico40", synthetic.icosphere(40), dict(lambda_=19.0)),
                         ("ico20cot", (synthetic.perturb(synthetic.icosphere(20)[0], radial=0.05, tangential=0.2, edge=0.1, seed=5), synthetic.icosphere(20)[1]), dict(lambda_=0.0, alpha=0.9, cotan=True))]:
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, **kw)
    idx, val = M.indices().cpu().numpy(), M.values().cpu().numpy()
    for k in (1, 3, 4, 6):
        b = np.random.default_rng(k).standard_normal((v.shape[0], k)).astype(np.float32)
        x64 = osv.from_differential(idx[0], idx[1], val, b)
        s = NestedDissectionSolver(M)
        x = s.solve(torch.from_numpy(b).to(dev))
        x2 = s.solve(torch.from_numpy(b).to(dev))
        print(name, "k", k, "D", s.plan.D, "rel err", float(np.abs(x.cpu().numpy() - x64).max() / np.abs(x64).max()), "deterministic", bool(torch.equal(x, x2)), flush=True)
for cfg_name in ("cfg3_dragon250k", "cfg4_plane1m"):
    v, f, cfg = synthetic.config_mesh(cfg_name)
    tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
    M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
    u = to_differential(M, tv)
    for leaf in (48, 96, 24):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s = NestedDissectionSolver(M, leaf_size=leaf)
        torch.cuda.synchronize(); t_build = time.perf_counter() - t0
        for _ in range(3): x = s.solve(u)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): x = s.solve(u)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        s.set_option("profile", 1); s.solve(u); info = s.info(); s.set_option("profile", 0)
        print(f"{cfg_name} leaf={leaf} D={s.plan.D} build {t_build:.1f}s solve {dt*1e3:.3f} ms  err_vs_v {float((x - tv).abs().max()):.2e} "
              f"entries/V {info['factor_entries']/v.shape[0]:.1f} launches {info['launches']} up {info['up_ms']*1e3:.0f}us down {info['down_ms']*1e3:.0f}us perm {info['perm_ms']*1e3:.0f}us "
              f"GB/s {info['factor_entries']*4/dt/1e9:.0f}", flush=True)
        del s
    c = CholeskySolver(M)
    for _ in range(3): y = c.solve(u)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): y = c.solve(u)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{cfg_name} iterative: {dt*1e3:.3f} ms {c.last_info['method']} {c.last_info['iterations']} its err_vs_v {float((y - tv).abs().max()):.2e}", flush=True)
PY
