#!/usr/bin/env python3
"""Per-kernel means of SQ counters from a rocprofv3 --pmc pass: python tools/pmc_sq_summary.py <dir> [prefix]"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(acc):
    c = {k: sum(v) / len(v) for k, v in acc[name].items()}
    n = len(next(iter(acc[name].values())))
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    line = f"{name:22s} n={n:3d} waves {c.get('SQ_WAVES', 0):9.0f}  wave_cycles {wc:12.0f}"
    if wc:
        line += f"  waiting {100 * c.get('SQ_WAIT_ANY', 0) / wc:5.1f} %  wait_inst {100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:5.1f} %  issuing {100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.1f} %"
    if "SQ_INSTS_VMEM_RD" in c and c.get("SQ_WAVES"):
        line += f"  vmem_rd/wave {c['SQ_INSTS_VMEM_RD'] / c['SQ_WAVES']:6.1f}"
    print(line)
