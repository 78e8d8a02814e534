#!/bin/bash
# is the small-batch forward (C1: 2 x 1024 points) bound by the kernels or by the launches?  rocprofv3 kernel durations against the step time
tag=$1
out=$PWD/gpurun_out
root=$PWD
timeout -k 10 200 python3 bench.py --config c1_dgcnn_global_affine --batch 2 --points 1024 --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --no-profile > $out/${tag}_c1_noprofile.json 2> $out/${tag}_c1_noprofile.err
python3 - <<PY
import json
j=[json.loads(l) for l in open("$out/${tag}_c1_noprofile.json") if l.startswith("{")][-1]
print("C1 without the in-library profiler:", round(j["ms_per_step"],3), "ms per step")
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c1 -o c -- python3 $root/bench.py --config c1_dgcnn_global_affine --batch 2 --points 1024 --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --no-profile > $out/${tag}_c1_under_rocprof.log 2>&1 || { echo rocprof failed; exit 1; }
cp $(find /tmp/prof_c1 -name "c_kernel_stats.csv" | head -1) $out/${tag}_c1_kernel_stats.csv
python3 - <<PY
import csv, json
rows=list(csv.DictReader(open("$out/${tag}_c1_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print(f"25 steps: {calls} launches, {tot/1e6:.2f} ms of kernel time -> {tot/1e6/25:.3f} ms and {calls/25:.0f} launches per step")
for r in rows[:8]: print(f"  {r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.2f} us")
j=[json.loads(l) for l in open("$out/${tag}_c1_under_rocprof.log") if l.startswith("{")][-1]
print("step time under rocprof:", round(j["ms_per_step"],3))
PY
