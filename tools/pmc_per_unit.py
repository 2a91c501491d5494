"""Per-unit HBM traffic and vector-instruction counts of the bench kernels from rocprofv3 counter passes.

    python tools/pmc_per_unit.py <workload> <streams> <dir FETCH_SIZE run> <dir WRITE_SIZE run> <dir SQ_INSTS run>

Every run is `bench.py --workload <workload> --groups 1 ...`, so every dispatch of a per-frame kernel covers <streams>
frames; the tracking and rejection kernels are counted per point (lk_kernel runs one wavefront per point: points of a
dispatch = grid size / 64, less the few wavefronts that round the grid up to a multiple of 8).  FETCH_SIZE and
WRITE_SIZE are in KiB; gfx950 reports half the bytes of 128-byte line requests in FETCH_SIZE, so it is doubled.
Prints one JSON object: {"workload", "traffic": {kind: [bytes, unit]}, "valu": {kind: [wave instructions, unit]}, "kernels": {...}}

    python tools/pmc_per_unit.py --install <tag> <workload> [<workload> ...]

copies gpurun_out/prof_<tag>/pmc_per_unit_<workload>.json to profiles/<tag>_pmc_per_unit_<workload>.json and writes their
"traffic" and "valu" tables into profiles/pmc_per_unit.json, the file bench.py reads."""
import csv
import glob
import json
import os
import re
import sys

KINDS = {"lk_track": ("lk_kernel", "lk_pair_kernel"), "pyramid": ("level_kernel<true>", "level_kernel<false>", "ingest_kernel", "ingest_linear_kernel"),
         "fast_grid": ("fast_detect_kernel", "fast_topk_kernel"), "outlier_rejection": ("ransac_kernel",)}


def short(name):
    m = re.search(r"(lk_pair_kernel|lk_kernel|level_kernel<(?:true|false)|ingest_linear_kernel|ingest_kernel|fast_detect_kernel|fast_topk_kernel|ransac_kernel|fetch_block_kernel)", name)
    if not m:
        return None
    k = m.group(1)
    return k + ">" if k.startswith("level_kernel<") else k   # level_kernel<true, false> (frame pass, no gradient images) -> level_kernel<true>


def collect(d):
    """kernel -> {"dispatches", "grid", counter: sum}"""
    out = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if not k:
                    continue
                e = out.setdefault(k, {"dispatches": 0, "grid": 0})
                if row["Dispatch_Id"] not in seen:
                    seen.add(row["Dispatch_Id"])
                    e["dispatches"] += 1
                    e["grid"] += int(row["Grid_Size"])
                e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    return out


def install(tag, workloads):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    merged = {"note": "per-unit figures of the rocprofv3 counter passes of tools/profile_r3.sh (one-group runs of bench.py, 64 streams per launch): "
                      "FETCH_SIZE and WRITE_SIZE in runs of their own, FETCH_SIZE doubled (gfx950 counts half the bytes of 128-byte line requests), "
                      "SQ_INSTS_VALU per wavefront; raw summaries: profiles/%s_pmc_per_unit_<workload>.json" % tag, "workloads": {}}
    for wl in workloads:
        with open(os.path.join(root, "gpurun_out", "prof_" + tag, "pmc_per_unit_%s.json" % wl)) as f:
            one = json.load(f)
        with open(os.path.join(root, "profiles", "%s_pmc_per_unit_%s.json" % (tag, wl)), "w") as f:
            json.dump(one, f, indent=1, sort_keys=True)
        merged["workloads"][wl] = {"traffic": one["traffic"], "valu": one["valu"]}
    with open(os.path.join(root, "profiles", "pmc_per_unit.json"), "w") as f:
        json.dump(merged, f, indent=1, sort_keys=True)


def main():
    if sys.argv[1] == "--install":
        return install(sys.argv[2], sys.argv[3:])
    workload, streams = sys.argv[1], int(sys.argv[2])
    fetch, write, insts = collect(sys.argv[3]), collect(sys.argv[4]), collect(sys.argv[5])

    def units(run, kind):
        """(number of units the kind's kernels processed in that run, unit name)"""
        if kind in ("lk_track", "outlier_rejection"):
            lk = [run[k] for k in KINDS["lk_track"] if k in run]
            # one wavefront per point (two for lk_pair_kernel), grid rounded up to a multiple of 8 workgroups: ~3.5 spare per dispatch
            pts = sum((e["grid"] / (128.0 if name == "lk_pair_kernel" else 64.0)) - 3.5 * e["dispatches"]
                      for name, e in ((k, run[k]) for k in KINDS["lk_track"] if k in run))
            return max(pts, 1.0), "point"
        # frame sets = dispatches of the detector / of the pass that writes level 0
        lead = run.get("fast_detect_kernel") if kind == "fast_grid" else (run.get("level_kernel<true>") or run.get("ingest_linear_kernel"))
        return max(lead["dispatches"] * streams, 1) if lead else 1, "frame"

    res = {"workload": workload, "streams_per_dispatch": streams, "traffic": {}, "valu": {}, "kernels": {}}
    for kind, names in KINDS.items():
        fb = sum(2.0 * 1024.0 * fetch[k].get("FETCH_SIZE", 0.0) for k in names if k in fetch)
        wb = sum(1024.0 * write[k].get("WRITE_SIZE", 0.0) for k in names if k in write)
        nf, unit = units(fetch, kind)
        nw, _ = units(write, kind)
        vi = sum(insts[k].get("SQ_INSTS_VALU", 0.0) for k in names if k in insts)
        ni, _ = units(insts, kind)
        if fb or wb:
            res["traffic"][kind] = [fb / nf + wb / nw, unit]
        if vi:
            res["valu"][kind] = [vi / ni, unit]
    for k, e in insts.items():
        waves = e.get("SQ_WAVES", 0.0)
        res["kernels"][k] = {"dispatches": e["dispatches"], "waves_per_dispatch": waves / max(1, e["dispatches"]),
                             "valu_per_wave": e.get("SQ_INSTS_VALU", 0.0) / max(1.0, waves), "salu_per_wave": e.get("SQ_INSTS_SALU", 0.0) / max(1.0, waves),
                             "lds_per_wave": e.get("SQ_INSTS_LDS", 0.0) / max(1.0, waves), "vmem_rd_per_wave": e.get("SQ_INSTS_VMEM_RD", 0.0) / max(1.0, waves)}
    for k in set(fetch) | set(write):
        res["kernels"].setdefault(k, {}).update(fetch_kib=fetch.get(k, {}).get("FETCH_SIZE"), write_kib=write.get(k, {}).get("WRITE_SIZE"),
                                                dispatches_fetch_run=fetch.get(k, {}).get("dispatches"))
    json.dump(res, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
