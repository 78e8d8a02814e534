#!/bin/bash
# Counter passes over a short C2 run (8 layers): one rocprofv3 --pmc pass per counter group, summarised per kernel by profiles/pmc_sq.py.
#   bash profiles/micro/pmc_passes.sh OUT_PREFIX [bench.py args ...]
set -o pipefail
out=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
groups=(
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
 "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
 "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE"
 "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
 "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum GRBM_GUI_ACTIVE"
 "TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE"
)
i=0
for g in "${groups[@]}"; do
  d=/tmp/pmc_$i; rm -rf $d
  timeout -k 10 150 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $d -o x -- python3 $root/bench.py --layers 8 --steps 1 --warmup 1 --weights module --no-cpu-baseline "$@" > /dev/null 2> $root/gpurun_out/${out}_pass$i.err || { echo "pass $i failed" | tee -a $root/gpurun_out/${out}_progress.txt; grep -m1 "exceeds" $root/gpurun_out/${out}_pass$i.err; i=$((i+1)); continue; }
  f=$(find $d -name "x_counter_collection.csv" | head -1)
  python3 $root/profiles/pmc_sq.py $f $root/gpurun_out/${out}_pass$i.json > /dev/null && echo "pass $i ok" | tee -a $root/gpurun_out/${out}_progress.txt
  i=$((i+1))
done
