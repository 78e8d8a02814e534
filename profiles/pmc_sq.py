#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 --pmc CSV (one row per dispatch and counter), plus the ratios that say what a
kernel waits on.  SQ_* cycle counters are in quad-cycles summed over waves (MI355X_MICROARCH.md, rocprofv3 PMC slots):
WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~= WAVE_CYCLES.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS ... \\
              --kernel-trace --output-format csv -d DIR -o x -- python3 bench.py ...
    python profiles/pmc_sq.py DIR/x_counter_collection.csv [out.json]"""
import csv
import json
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(float))
    dur, n, seen = defaultdict(float), defaultdict(int), set()
    with open(sys.argv[1], newline="") as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                n[k] += 1
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    out = {}
    for k, c in acc.items():
        if not k.startswith(("fc::", "void fc::")):
            continue
        e = {"launches": n[k], "avg_us": dur[k] / n[k] * 1e6}
        for name, v in sorted(c.items()):
            e[name] = v / n[k]
        wc = e.get("SQ_WAVE_CYCLES")
        if wc:
            for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                         "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM"):
                if name in e:
                    e["frac_" + name[3:].lower()] = e[name] / wc
        if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_frac"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]
        if "GRBM_GUI_ACTIVE" in e:
            cyc = e["GRBM_GUI_ACTIVE"] / 8.0
            e["clock_ghz"] = cyc / (e["avg_us"] * 1e-6) / 1e9
            if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
                e["mfma_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
            if "SQ_LDS_IDX_ACTIVE" in e:
                e["lds_array_busy"] = e["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc)
        if e.get("SQ_INSTS_MFMA"):
            for name in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_VMEM_RD"):
                if name in e:
                    e[name[9:].lower() + "_per_mfma"] = e[name] / e["SQ_INSTS_MFMA"]
        out[k] = e
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"]))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)
    for k, v in list(out.items())[:6]:
        print(k[:100])
        print("   ", {kk: (round(vv, 4) if isinstance(vv, float) and abs(vv) < 100 else (round(vv) if isinstance(vv, float) else vv)) for kk, vv in v.items()})


if __name__ == "__main__":
    main()
