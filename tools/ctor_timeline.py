"""timeline of the LAST construction in a rocprofv3 kernel trace of tools/profile_constructor.py: per group of consecutive launches of
one kernel its launches, busy time and the idle time in front of it; then busy / idle totals per kernel.
   python tools/ctor_timeline.py kernel_trace.csv"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").replace("ls::", "").replace("(anonymous namespace)::", "")
# a construction ends with a solve (k_nd_tier<.., false ..>); it starts at the first k_f32_to_f64 / memcpy after the previous solve
ends = [i for i, r in enumerate(rows) if "k_nd_tier" in r["Kernel_Name"] and "false" in r["Kernel_Name"]]
if not ends: sys.exit("no solve in the trace")
hi = ends[-1]
lo = ends[-2] + 1 if len(ends) > 1 else 0
seg = rows[lo:hi + 1]
t0 = int(seg[0]["Start_Timestamp"])
groups, prev_end = [], t0
for r in seg:
    n, st, en = short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, st - prev_end)
    if groups and groups[-1][0] == n: g = groups[-1]; g[1] += 1; g[2] += en - st; g[3] += gap
    else: groups.append([n, 1, en - st, gap, st - t0])
    prev_end = max(prev_end, en)
busy, idle = collections.Counter(), collections.Counter()
for n, c, b, g, s in groups:
    print(f"{s / 1e6:9.3f} ms  {n[:60]:60s} x{c:<4d} busy {b / 1e3:9.1f} us  idle before/inside {g / 1e3:9.1f} us")
    busy[n] += b; idle[n] += g
print("---- totals (busy, idle in front of / between its launches)")
for n, b in busy.most_common(): print(f"{n[:60]:60s} {b / 1e3:10.1f} us  idle {idle[n] / 1e3:10.1f} us")
print(f"span {(prev_end - t0) / 1e6:.3f} ms, busy {sum(busy.values()) / 1e6:.3f} ms, idle {sum(idle.values()) / 1e6:.3f} ms")
