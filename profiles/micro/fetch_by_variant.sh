#!/bin/bash
# HBM-side read bytes per launch of the fused spline GEMM for several knob settings (short 8-layer C2 runs):
#   bash profiles/micro/fetch_by_variant.sh "13=2" "13=3" "13=4" "13=4 2=8"
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do
  args=""; for k in $kv; do args="$args --knob $k"; done
  rm -rf /tmp/pf
  timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o p -- python3 $root/bench.py --layers 8 --steps 1 --warmup 1 --weights module --no-cpu-baseline $args > /dev/null 2> $root/gpurun_out/fetch_var.err || { echo "$kv failed"; continue; }
  python3 - "$kv" $(find /tmp/pf -name "p_counter_collection.csv" | head -1) <<'PY' | tee -a $root/gpurun_out/fetch_by_variant.txt
import csv, sys
from collections import defaultdict
tot, n = defaultdict(float), defaultdict(set)
for r in csv.DictReader(open(sys.argv[2], newline="")):
    if r["Counter_Name"] == "FETCH_SIZE" and "gemm_f32_kernel<128, 128" in r["Kernel_Name"] and ", 4, " in r["Kernel_Name"]:
        tot[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]].add(r["Dispatch_Id"])
for k in tot:
    print(sys.argv[1], k[20:60], "launches", len(n[k]), "FETCH_SIZE raw KB/launch", round(tot[k] / len(n[k])), "-> MB read (x2, gfx950)", round(2 * tot[k] / len(n[k]) / 1e3))
PY
done
