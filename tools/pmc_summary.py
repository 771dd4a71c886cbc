"""Summarise rocprofv3 CSV output (--output-format csv): per kernel name, launches / average duration from the kernel trace
and per-launch averages of every PMC counter from the counter-collection file.  Usage:
    python tools/pmc_summary.py <rocprofv3 output dir> [name filter] > summary.json"""
import csv
import glob
import json
import os
import sys


def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0][:80]


def main(d, flt=None):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r.get("Kernel_Name", ""))
            if flt and flt not in k:
                continue
            e = out.setdefault(k, {"launches": 0, "total_us": 0.0, "max_us": 0.0, "counters": {}})
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            e["launches"] += 1
            e["total_us"] += dur
            e["max_us"] = max(e["max_us"], dur)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = {}
        for r in csv.DictReader(open(f)):
            k = short(r.get("Kernel_Name", ""))
            if flt and flt not in k:
                continue
            e = out.setdefault(k, {"launches": 0, "total_us": 0.0, "max_us": 0.0, "counters": {}})
            c = e["counters"].setdefault(r["Counter_Name"], {"sum": 0.0, "dispatches": 0})
            c["sum"] += float(r["Counter_Value"])
            key = (k, r["Counter_Name"], r.get("Dispatch_Id"))
            if key not in seen:
                seen[key] = 1
                c["dispatches"] += 1
    for k, e in out.items():
        if e["launches"]:
            e["avg_us"] = e["total_us"] / e["launches"]
        for c in e["counters"].values():
            c["per_dispatch"] = c["sum"] / max(c["dispatches"], 1)
        cs = {n: c["per_dispatch"] for n, c in e["counters"].items()}
        if "SQ_WAVE_CYCLES" in cs and cs["SQ_WAVE_CYCLES"]:
            w = cs["SQ_WAVE_CYCLES"]
            e["derived"] = {n.replace("SQ_", "").lower() + "_frac_of_wave_cycles": cs[n] / w
                            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
                                      "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM") if n in cs}
            if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and cs.get("SQ_BUSY_CU_CYCLES"):
                e["derived"]["mfma_busy_frac_of_cu_busy"] = cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * cs["SQ_BUSY_CU_CYCLES"])
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
