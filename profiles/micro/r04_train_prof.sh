#!/bin/bash
# rocprofv3 kernel statistics of one training step (after one warm-up step); $2... = extra bench flags
tag=$1; shift
out=$PWD/gpurun_out
root=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o t -- python3 $root/bench.py --train --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/${tag}_train_under_rocprof.log 2>&1 || { echo "rocprof failed"; tail -5 $out/${tag}_train_under_rocprof.log; exit 1; }
f=$(find /tmp/prof_t -name "*kernel_stats.csv" | head -1)
cp $f $out/${tag}_train_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/${tag}_train_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms over 3 steps", tot/1e6)
for r in rows[:16]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.1f} {float(r['AverageNs'])/1e3:8.1f} {r['Percentage']}")
PY
f2=$(find /tmp/prof_t -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv, collections
h=collections.Counter(); tot=collections.Counter()
for r in csv.DictReader(open("$f2")):
    if "spline_wide_kernel<3" in r["Kernel_Name"]:
        d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
        b=int(d//50)*50
        h[b]+=1; tot[b]+=d
print("spline_wide_kernel<3> launch durations (us bucket: count, mean)")
for b in sorted(h): print(f"  {b:5d}-{b+50:5d}: {h[b]:5d}  {tot[b]/h[b]:8.1f}")
PY
