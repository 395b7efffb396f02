"""Per-phase timeline of the tier kernels from per-wave clock stamps (experiments build only):
    tools/build_variant.sh stamps "-DLS_TIER_STAMPS" direct.hip
    LARGESTEPS_HIP_LIB=tools/build/v_stamps/liblargesteps_hip.so python tools/tier_stamps.py [cfg] [tier_waves]
Slots (csrc/nd_tier.h): 0 wave start, 1 header + gather done, 2 + 2 ph work of phase ph done, 3 + 2 ph its barrier passed, 16 + r r-th leaf done,
24 + ph items this wave ran in phase ph. Clock: s_memrealtime, 100 MHz."""
import ctypes, os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import numpy as np, torch
from largesteps import synthetic, _native
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential
from largesteps.solvers import NestedDissectionSolver
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "cfg4_plane1m"
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
v, f, cfg = synthetic.config_mesh(cfg_name)
tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
u = to_differential(M, tv)
s = NestedDissectionSolver(M, tier_waves=waves)
inf = s.info()
for _ in range(5): s.solve(u)
s.set_option("profile", 2)
s.solve(u); s.solve(u)
torch.cuda.synchronize()
lib = ctypes.CDLL(_native.lib_path())
lib.ls_direct_tier_stamps.restype = ctypes.c_int
lib.ls_direct_tier_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
n_wg, H = inf["tier_workgroups"], inf["tier_levels"]
W = {0: None}.get(waves, waves)
for cand in ((W,) if W else (16, 8, 4)):
    n = 2 * n_wg * cand * 32
    buf = np.zeros(n, dtype=np.int64)
    if lib.ls_direct_tier_stamps(s._direct._h, buf.ctypes.data_as(ctypes.c_void_p), n) == 0:
        W = cand
        break
st = buf.reshape(2, n_wg, W, 32).astype(np.float64)
bal = s.tier_balance()
lvl_up, lvl_down = s.level_words()
idx_up, idx_down = s.level_index_bytes()
rows, bnd = s.level_rows()
L = inf["levels"]
print(f"{cfg_name}: V {v.shape[0]}, {L} levels, tier of {H} levels on {n_wg} workgroups x {W} waves; subtree words max / mean {bal['max_over_mean']:.3f}")
for sw, name in enumerate(("up", "down")):
    t = st[sw]
    t0 = t[:, :, 0].min()
    us = (t - t0) / 100.0
    print(f"{name} sweep: wave start mean {us[:, :, 0].mean():6.2f} max {us[:, :, 0].max():6.2f} us; header + gather done mean {us[:, :, 1].mean():6.2f} max {us[:, :, 1].max():6.2f}")
    prev_mean, prev_max = us[:, :, 1].mean(), us[:, :, 1].max()
    for ph in range(H):
        lv = L - 1 - ph if sw == 0 else L - H + ph
        work, bar = us[:, :, 2 + 2 * ph], us[:, :, 3 + 2 * ph]
        items = t[:, :, 24 + ph]
        wg_done = bar.max(axis=1)                   # per workgroup: when its barrier released
        words = (lvl_up if sw == 0 else lvl_down)[lv]
        ib = (idx_up if sw == 0 else idx_down)[lv]
        print(f"  phase {ph} (level {lv}, {words * 4e-6:6.1f} MB factor + {ib * 1e-6:5.1f} MB index, items per wave mean {items.mean():.2f} max {items.max():.0f}): "
              f"work done mean {work.mean():6.2f} max {work.max():6.2f} | barrier passed mean {bar.mean():6.2f} max {bar.max():6.2f} | "
              f"phase length mean {bar.mean() - prev_mean:5.2f} (slowest workgroup {wg_done.max() - prev_max:5.2f}) us")
        if ph == (0 if sw == 0 else H - 1):
            for r in range(8):
                c = t[:, :, 16 + r]
                m = c > 0
                if m.any():
                    print(f"      leaf {r} done: mean {us[:, :, 16 + r][m].mean():6.2f} max {us[:, :, 16 + r][m].max():6.2f}  ({int(m.sum())} waves)")
        prev_mean, prev_max = bar.mean(), bar.max()
    end = us[:, :, 3 + 2 * (H - 1)]
    per_wg = end.max(axis=1)
    print(f"  kernel body: last wave done {end.max():6.2f} us; per-workgroup end min {per_wg.min():6.2f} mean {per_wg.mean():6.2f} max {per_wg.max():6.2f}")
