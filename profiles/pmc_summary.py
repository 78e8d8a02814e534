#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 counter passes (run separately: TCC fits only one of FETCH_SIZE / WRITE_SIZE per pass):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py ...
    python profiles/pmc_summary.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv out.json [label]

Counter values are KB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly half the
bytes of wide coalesced reads, so hbm_bytes_per_launch = 2 * FETCH + WRITE.  bench.py reads the JSON for `roofline.traffic`."""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                a = acc[r["Kernel_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0])[1] + write.get(k, [0, 0])[1])):
        if not k.startswith(("fc::", "void fc::")):
            continue
        fl, fs = fetch.get(k, [0, 0.0])
        wl, ws = write.get(k, [0, 0.0])
        e = {"fetch": {"launches": fl, "avg_KB_raw": fs / max(fl, 1)}, "write": {"launches": wl, "avg_KB": ws / max(wl, 1)}}
        e["hbm_bytes_per_launch"] = (2.0 * e["fetch"]["avg_KB_raw"] + e["write"]["avg_KB"]) * 1024.0
        out[k] = e
    import datetime
    json.dump({"unit_note": "FETCH raw KB doubled (gfx950), WRITE as read; per dispatch averages",
               "measured": {"label": sys.argv[4] if len(sys.argv) > 4 else "", "date": datetime.date.today().isoformat(),
                            "workload": sys.argv[5] if len(sys.argv) > 5 else "c2_dgcnn_attn_spline 16 x 4096 + 4096"},
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k, e in list(out.items())[:8]:
        print(f"{e['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch  {k[:90]}")


if __name__ == "__main__":
    main()
