"""every launch of the LAST construction in a rocprofv3 kernel trace of tools/profile_constructor.py, in order: start, duration, grid, workgroup.
   python tools/ctor_launches.py kernel_trace.csv [substring of the kernel name]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").replace("ls::", "").replace("(anonymous namespace)::", "")
ends = [i for i, r in enumerate(rows) if "k_nd_tier" in r["Kernel_Name"] and "false" in r["Kernel_Name"]]
lo = ends[-2] + 1 if len(ends) > 1 else 0
seg = rows[lo:ends[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"])
want = sys.argv[2] if len(sys.argv) > 2 else ""
prev = t0
for r in seg:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if want in r["Kernel_Name"]:
        print(f"{(st - t0) / 1e3:10.1f} us  +{(st - prev) / 1e3:7.1f} idle  {short(r['Kernel_Name'])[:40]:40s} dur {(en - st) / 1e3:8.1f} us  grid {r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '?')} wg {r.get('Workgroup_Size_X', '?')}")
    prev = max(prev, en)
