#!/bin/bash
# Effective shader clock during the tracking kernel: GPU busy cycles of every dispatch over its duration.
set -e
out=gpurun_out/pmc_clock
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/a -o p -- python3 tools/lk_bench.py > $out/a.log 2> $out/a.err
python3 - <<'PY'
import csv, glob
for path in glob.glob("gpurun_out/pmc_clock/a/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(path)))
    print(list(rows[0].keys()))
    acc = {}
    for r in rows:
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
        k = r["Kernel_Name"].split("(")[0]
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = acc.setdefault(k, [0.0, 0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += dur; a[2] += 1
    for k, (cyc, ns, n) in acc.items():
        print(f"{k[:50]:50s} calls {n:4d} mean {ns / n / 1e3:8.1f} us  cycles/ns {cyc / max(ns, 1):.3f}")
PY
find $out -name "*.db" -delete
