#!/bin/bash
# Training step (bench.py --train) plain and under rocprofv3 --kernel-trace --stats:  bash profiles/micro/train_stats.sh TAG
set -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
cd $root
timeout -k 10 300 python3 bench.py --train --steps 3 --warmup 1 > $out/${tag}_train.json 2> $out/${tag}_train.err || { echo "train bench failed"; tail -5 $out/${tag}_train.err; exit 1; }
cut -c1-400 $out/${tag}_train.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_t
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o t -- python3 $root/bench.py --train --steps 2 --warmup 1 > $out/${tag}_train_under_rocprof.log 2>&1 || { echo "rocprof failed"; tail -5 $out/${tag}_train_under_rocprof.log; exit 1; }
cp $(find /tmp/prof_t -name "t_kernel_stats.csv" | head -1) $out/${tag}_train_kernel_stats.csv
head -16 $out/${tag}_train_kernel_stats.csv | cut -c1-160
