#!/bin/bash
# C1 (2 x 1024 points) under rocprofv3 --kernel-trace: true kernel durations and the gaps between dependent launches
#   bash profiles/micro/c1_trace.sh TAG   -> gpurun_out/TAG_c1_kernel_stats.csv, TAG_c1_gaps.txt
set -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c1 -o s -- python3 $root/bench.py --config c1_dgcnn_global_affine --batch 2 --points 1024 --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 > $out/${tag}_c1_under_rocprof.log 2>&1 || { echo "rocprof failed"; tail -5 $out/${tag}_c1_under_rocprof.log; exit 1; }
cp $(find /tmp/prof_c1 -name "s_kernel_stats.csv" | head -1) $out/${tag}_c1_kernel_stats.csv
python3 - $(find /tmp/prof_c1 -name "s_kernel_trace.csv" | head -1) > $out/${tag}_c1_gaps.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 20 % of the trace = timed steps
n = len(rows); rows = rows[int(n * 0.6):]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    dur[a['Kernel_Name'][:70]].append(int(a['End_Timestamp']) - int(a['Start_Timestamp']))
    gap[a['Kernel_Name'][:70]].append(int(b['Start_Timestamp']) - int(a['End_Timestamp']))
tot_d = sum(sum(v) for v in dur.values()); tot_g = sum(sum(v) for v in gap.values())
print(f"kernels {len(rows)}  busy {tot_d/1e6:.3f} ms  gaps {tot_g/1e6:.3f} ms")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d = dur[k]; g = gap[k]
    print(f"{k:70s} n {len(d):6d}  dur avg {sum(d)/len(d)/1e3:7.2f} us  gap-after avg {sum(g)/len(g)/1e3:7.2f} us  med {sorted(g)[len(g)//2]/1e3:6.2f}")
PY
cat $out/${tag}_c1_gaps.txt
