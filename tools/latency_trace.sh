#!/bin/bash
# Kernel timeline of the single-stream latency loop: per-kernel durations and the gaps between consecutive kernels.
set -e
out=gpurun_out/lat_trace
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o lat -- python3 tools/latency_stages.py > $out/run.log 2> $out/err.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/lat_trace/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]          # steady state
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    n = a["Kernel_Name"].split("(")[0][-40:]
    dur[n].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
    gap[n + " -> " + b["Kernel_Name"].split("(")[0][-28:]].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
med = lambda v: sorted(v)[len(v) // 2]
for k, v in dur.items(): print(f"dur {k:42s} n={len(v):4d} median {med(v):7.2f} us")
for k, v in gap.items(): print(f"gap {k:72s} n={len(v):4d} median {med(v):7.2f} us")
PY
grep -v amdgpu.ids $out/run.log
find $out -name "*.db" -delete
