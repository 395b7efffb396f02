#!/usr/bin/env python3
"""
Merge two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command, --kernel-trace only) into
profiles/r01_pmc_traffic.json: per kernel name the number of dispatches and the HBM-side bytes per dispatch.

    python tools/pmc_summary.py <dir with *counter_collection.csv of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <workload> <out.json>

Counter unit: KiB. gfx950 correction (MI355X_MICROARCH.md, HBM section; re-calibrated in this repo on two kernels of
known volume): FETCH_SIZE reports half the bytes of coalesced streaming reads -> doubled;
WRITE_SIZE is exact.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read(dirname, counter):
    out = defaultdict(list)
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") == counter:
                    out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    fetch_dir, write_dir, workload, out_path = sys.argv[1:5]
    fetch, write = read(fetch_dir, "FETCH_SIZE"), read(write_dir, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        if not name.startswith("ls::") and "ls::" not in name:
            continue
        f, w = fetch.get(name, []), write.get(name, [])
        n = max(len(f), len(w))
        rd = 2.0 * 1024.0 * sum(f) / max(len(f), 1)
        wr = 1024.0 * sum(w) / max(len(w), 1)
        key = name.replace("void ", "").split("(")[0]
        kernels[key] = dict(dispatches=n, fetch_size_kib_mean_raw=sum(f) / max(len(f), 1), write_size_kib_mean=sum(w) / max(len(w), 1),
                            read_bytes_corrected=rd, write_bytes=wr, traffic_bytes=rd + wr)
    doc = dict(_how=("rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate passes) of `python bench.py "
                     "--steps 2 --warmup 1 --no-cpu-baseline` on MI355X; MEAN per dispatch (the tree-level kernels differ per "
                     "level); counter unit KiB; FETCH_SIZE doubled (gfx950 reports half the bytes of coalesced streaming reads -- "
                     "MI355X_MICROARCH.md HBM section; calibrated in this repo on copy4: 12.0 MB read -> 5.73 MiB, and on k3_row: "
                     "28.0 MB -> 13.36 MiB); WRITE_SIZE needs no correction (copy4: 12.0 MB -> 11.44 MiB)."),
               workload=workload, kernels=kernels)
    with open(out_path, "w") as fh:
        json.dump(doc, fh, indent=1)
    for k, v in kernels.items():
        print(f"{k:48s} n={v['dispatches']:5d} traffic/dispatch {v['traffic_bytes'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
