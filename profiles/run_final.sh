#!/bin/bash
# Final measurements of a build, all on ONE box:  bash profiles/run_final.sh TAG   (writes gpurun_out/TAG_*; copy what is judged into profiles/)
#   1 bench.py (default run: the headline line, with the `train` object and the CPU baseline leg; the profiled runs below leave the training leg out:
#     rocprofv3's counter collection crashes inside torch's autograd threads)
#   2 rocprofv3 --kernel-trace --stats of a short bench run (per-kernel average durations, to agree with the in-library HIP events)
#   3 separate --pmc passes: FETCH_SIZE, WRITE_SIZE (HBM bytes per launch), matrix-pipe busy cycles + clock
#   4 one bench line per BASELINE configuration at its stated size, and one for the reference's native training shape (N != M)
# afterwards: cp gpurun_out/TAG_c2_pmc_hbm_traffic.json profiles/pmc_traffic.json ; cp gpurun_out/TAG_c2_pmc_mfma_busy.json profiles/pmc_mfma_busy.json  (what bench.py quotes)
set -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
say() { echo "$(date +%T) $*" | tee -a $out/${tag}_progress.txt; }
cd $root
say "bench"
timeout -k 10 600 python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { say "bench failed"; exit 1; }
cd /tmp && export TMPDIR=/tmp
say "rocprof stats"
rm -rf /tmp/prof_s
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --train-steps 0 > $out/${tag}_bench_under_rocprof.log 2>&1 || { say "rocprof stats failed"; exit 1; }
cp $(find /tmp/prof_s -name "s_kernel_stats.csv" | head -1) $out/${tag}_c2_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  say "pmc $c"
  rm -rf /tmp/prof_$c
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_$c -o p -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-steps 0 > $out/${tag}_pmc_$c.log 2>&1 || { say "pmc $c failed"; exit 1; }
done
python3 $root/profiles/pmc_summary.py $(find /tmp/prof_FETCH_SIZE -name "p_counter_collection.csv" | head -1) $(find /tmp/prof_WRITE_SIZE -name "p_counter_collection.csv" | head -1) $out/${tag}_c2_pmc_hbm_traffic.json "build $tag" "c2_dgcnn_attn_spline 16 x 4096 + 4096" > /dev/null
say "pmc mfma"
rm -rf /tmp/prof_m
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/prof_m -o p -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-steps 0 > $out/${tag}_pmc_mfma.log 2>&1 || { say "pmc mfma failed"; exit 1; }
python3 $root/profiles/pmc_mfma.py $(find /tmp/prof_m -name "p_counter_collection.csv" | head -1) $out/${tag}_c2_pmc_mfma_busy.json "build $tag" "c2_dgcnn_attn_spline 16 x 4096 + 4096" > /dev/null
cd $root
i=1
for cfg in "c1_dgcnn_global_affine 2 1024" "c2_dgcnn_attn_spline 16 4096" "c3_paconv_attn_affine 16 4096" "c4_dgcnn_attn_extra_affine 8 4096" "c4_dgcnn_attn_extra_affine 16 16384"; do
  set -- $cfg
  say "config $i: $cfg"
  extra="--train-steps 0"; [ $i = 2 ] && extra="--no-cpu-baseline --train-steps 0"; [ $i = 3 ] && extra="--no-cpu-baseline --train-steps 0"; [ $i = 5 ] && extra="--no-cpu-baseline --train-steps 0 --steps 3 --warmup 1"
  timeout -k 10 500 python3 bench.py --config $1 --batch $2 --points $3 $extra > $out/${tag}_bench_c$i.json 2> $out/${tag}_bench_c$i.err || { say "config $i failed"; exit 1; }
  i=$((i+1))
done
say "native shape: 20 scenes x 1024 target x 1250 context points (config/dulcet-universe.yaml)"
timeout -k 10 500 python3 bench.py --config c4_dgcnn_attn_extra_affine --batch 20 --points 1024 --ctx-points 1250 --train-steps 0 > $out/${tag}_bench_native.json 2> $out/${tag}_bench_native.err || { say "native shape failed"; exit 1; }
say "done"
