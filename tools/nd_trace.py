"""print the per-launch durations of the LAST solve in a rocprofv3 kernel trace csv: python tools/nd_trace.py file.csv"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_nd_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def tier_up(n):
    # k_nd_tier<K, UP, W, NT>: the second template argument is the direction
    return n.split("k_nd_tier<", 1)[1].split(",")[1].strip().startswith("true")


def is_up(n):
    return "k_nd_up" in n or ("k_nd_tier" in n and tier_up(n))


def is_down(n):
    return "k_nd_down" in n or ("k_nd_tier" in n and not tier_up(n))


# a solve starts with the first up-sweep kernel after a down-sweep kernel
starts = [i for i, r in enumerate(rows) if is_up(r["Kernel_Name"]) and (i == 0 or is_down(rows[i - 1]["Kernel_Name"]))]
last = rows[starts[-1]:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    name = r["Kernel_Name"].split("(")[0].replace("void ls::", "")
    print(f"{name:28s} start {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))}")
print("total", (int(last[-1]["End_Timestamp"]) - t0) / 1e3, "us")
