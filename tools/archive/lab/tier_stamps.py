"""per-phase timeline of the tier kernels (profile = 2): python tools/tier_stamps.py [cfg]"""
import ctypes, os, sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_R, os.path.join(_R, "large-steps-pytorch_amd")]
import numpy as np, torch
from largesteps import synthetic, _native
from largesteps.geometry import compute_matrix
from largesteps.parameterize import to_differential
from largesteps.solvers import NestedDissectionSolver
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "cfg4_plane1m"
dev = torch.device("cuda:0")
v, f, cfg = synthetic.config_mesh(cfg_name)
tv, tf = torch.from_numpy(v).to(dev), torch.from_numpy(f).to(dev)
M = compute_matrix(tv, tf, cfg["lambda_"] if cfg["lambda_"] is not None else 0.0, alpha=cfg["alpha"], cotan=cfg["cotan"])
u = to_differential(M, tv)
s = NestedDissectionSolver(M)
for _ in range(3): s.solve(u)
s.set_option("profile", 2)
s.solve(u)
h = s._direct._h
n_wg = s.info()["tier_workgroups"]
n = 2 * n_wg * 4 * 32
buf = np.zeros(n, dtype=np.int64)
_native.check(_native.lib().ls_direct_tier_stamps(h, buf.ctypes.data_as(ctypes.c_void_p), n))
st = buf.reshape(2, n_wg, 4, 32).astype(np.float64)
for sw, name in enumerate(("up", "down")):
    t = st[sw]
    t0 = t[:, :, 0].min()
    rel = (t - t0) / 100.0          # s_memtime ticks at 100 MHz -> us
    print(f"{name}: wave start  mean {rel[:, :, 0].mean():.2f} us  max {rel[:, :, 0].max():.2f}")
    for ph in range(3):
        a, b = rel[:, :, 1 + 2 * ph], rel[:, :, 2 + 2 * ph]
        if t[:, :, 1 + 2 * ph].max() == 0: continue
        print(f"  phase {ph}: work done mean {a.mean():.2f} min {a.min():.2f} max {a.max():.2f} | barrier passed mean {b.mean():.2f} max {b.max():.2f}")
    for r in range(8):
        c = t[:, :, 16 + r]
        if c.max() == 0: continue
        m = c > 0
        print(f"  leaf {r} done: mean {rel[:, :, 16 + r][m].mean():.2f} max {rel[:, :, 16 + r][m].max():.2f}")
    own = t - t[:, :, :1]
    for slot, label in ((24, "leaf phase entered"), (25, "first item + its index loads arrived"), (26, "first leaf's data arrived"),
                        (27, "first triangle staged"), (28, "next leaf's loads issued"), (16, "first leaf done"), (29, "LAST leaf matvec done"), (19, "4th leaf done")):
        c = own[:, :, slot]
        m = t[:, :, slot] > 0
        if m.any():
            print(f"    [{slot}] {label}: mean {c[m].mean() / 100:.2f}  (x100 ticks since the wave's start)")
