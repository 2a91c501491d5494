"""Sum rocprofv3 --pmc counter_collection CSVs per kernel: python tools/summarize_pmc.py <dir> [<dir> ...] > out.json
For every kernel: dispatches, total grid size, and for every counter the sum over dispatches."""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r"(lk_kernel<[^>]*>|level_kernel|ingest_kernel|fast_detect_kernel|fast_topk_kernel|predict_flow_kernel)", name)
    return m.group(1) if m else name[:60]


def main():
    out = {}
    for d in sys.argv[1:]:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    k = short(row["Kernel_Name"])
                    e = out.setdefault(k, {"dispatches": 0, "grid_size": 0, "counters": {}})
                    key = (row["Dispatch_Id"], k)
                    if key not in seen:
                        seen.add(key)
                        e["dispatches"] += 1
                        e["grid_size"] += int(row["Grid_Size"])
                    c = row["Counter_Name"]
                    e["counters"][c] = e["counters"].get(c, 0.0) + float(row["Counter_Value"])
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
