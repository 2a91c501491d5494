"""Per-kernel means of the counters in rocprofv3 counter_collection CSVs (directories given on the command line)."""
import csv, glob, os, sys
from collections import defaultdict

for d in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k].add(row["Dispatch_Id"])
    print("==", d)
    for k, c in acc.items():
        n = len(calls[k])
        waves = c.get("SQ_WAVES", 0) / n
        line = f"{k[:60]:60s} calls {n:4d} waves/call {waves:9.0f}"
        for name in sorted(c):
            if name != "SQ_WAVES":
                line += f" | {name[3:]} /wave {c[name] / n / max(waves, 1):9.1f}"
        print(line)
