"""Per-kernel LDS bank-conflict and vector-ALU busy rates from a rocprofv3 counter pass (tools/pmc_lds_valu.sh).

lds_bank_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE   (extra LDS-array cycles / all LDS-array cycles)
valu_busy         = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * device cycles of the dispatch); the device cycles are
                    GRBM_GUI_ACTIVE / 8 (rocprofv3 reports the sum over the 8 XCDs) and, as a cross-check, the
                    dispatch's own timestamps at 2.4 GHz.
All counters are sums over the counter's instances, averaged over the dispatches of a kernel."""
import csv, glob, os, sys
from collections import defaultdict

SIMDS = 256 * 4
for d in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(float))
    inst = defaultdict(lambda: defaultdict(int))
    disp = defaultdict(dict)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            inst[k][row["Counter_Name"]] += 1
            if "Start_Timestamp" in row and row.get("End_Timestamp"):
                disp[k][row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            else:
                disp[k].setdefault(row["Dispatch_Id"], 0)
    print("==", d)
    for k, c in acc.items():
        n = len(disp[k])
        waves = c.get("SQ_WAVES", 0) / n
        rows_per_disp = {name: inst[k][name] / n for name in c}
        gui = c.get("GRBM_GUI_ACTIVE", 0) / n / max(rows_per_disp.get("GRBM_GUI_ACTIVE", 1), 1) / 8
        ns = sum(disp[k].values()) / n
        valu = 4 * c.get("SQ_ACTIVE_INST_VALU", 0) / n
        conflict, idx = c.get("SQ_LDS_BANK_CONFLICT", 0) / n, c.get("SQ_LDS_IDX_ACTIVE", 0) / n
        line = f"{k[:44]:44s} calls {n:4d} waves/call {waves:8.0f} | lds_bank_conflict {100 * conflict / idx if idx else 0:6.2f} % ({conflict:12.0f} of {idx:12.0f} LDS cycles)"
        line += f" | valu_busy {100 * valu / (SIMDS * gui) if gui else 0:5.1f} % of {gui:8.0f} cycles"
        if ns:
            line += f" ({100 * valu / (SIMDS * ns * 2.4):5.1f} % of {ns / 1e3:7.1f} us at 2.4 GHz)"
        line += f" | lds_insts/wave {c.get('SQ_INSTS_LDS', 0) / n / max(waves, 1):7.1f} | active_inst_lds/wave {4 * c.get('SQ_ACTIVE_INST_LDS', 0) / n / max(waves, 1):8.1f} cyc | wave_cycles/wave {4 * c.get('SQ_WAVE_CYCLES', 0) / n / max(waves, 1):9.0f}"
        print(line)
