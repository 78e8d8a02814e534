#!/usr/bin/env python3
"""Per-kernel matrix-pipe utilisation from one rocprofv3 counter pass:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace \\
              --output-format csv -d gpurun_out/pmc_mfma -o m -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python profiles/pmc_mfma.py gpurun_out/pmc_mfma/m_counter_collection.csv out.json

SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 matrix pipes (MI355X_MICROARCH.md: 32 per 32x32x16 16-bit MFMA);
GRBM_GUI_ACTIVE sums the active cycles of the 8 XCDs.  mfma_busy = MFMA_BUSY / (1024 * GUI_ACTIVE / 8); the clock the kernel ran
at follows from GUI_ACTIVE / 8 over the traced duration."""
import csv
import json
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(float))
    dur = defaultdict(float)
    n = defaultdict(int)
    seen = set()
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
        if not k.startswith(("fc::", "void fc::")) or c.get("GRBM_GUI_ACTIVE", 0) <= 0:
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        out[k] = {"launches": n[k], "avg_us": dur[k] / n[k] * 1e6, "clock_ghz": cyc / dur[k] / 1e9 if dur[k] else None,
                  "mfma_busy": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc),
                  "mfma_insts_per_launch": c.get("SQ_INSTS_MFMA", 0.0) / n[k], "valu_insts_per_launch": c.get("SQ_INSTS_VALU", 0.0) / n[k]}
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"]))
    for v in out.values():
        v["mfma_busy_frac"] = v["mfma_busy"]
    import datetime
    # argv[3]: build label, argv[4]: workload key as bench.py spells it ("<config> <B> x <N> + <M>"): bench.py only quotes a table of its own workload
    json.dump({"unit_note": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 pipes x GRBM_GUI_ACTIVE / 8); clock = GRBM_GUI_ACTIVE / 8 / traced duration",
               "measured": {"label": sys.argv[3] if len(sys.argv) > 3 else "", "date": datetime.date.today().isoformat(),
                            "workload": sys.argv[4] if len(sys.argv) > 4 else "c2_dgcnn_attn_spline 16 x 4096 + 4096"},
               "kernels": out}, open(sys.argv[2], "w"), indent=1)
    for k, v in list(out.items())[:8]:
        print(f"{v['mfma_busy'] * 100:5.1f} % MFMA busy  {v['clock_ghz'] or 0:4.2f} GHz  {v['avg_us']:8.1f} us  {k[:80]}")


if __name__ == "__main__":
    main()
