#!/bin/bash
# The two SQ counter passes of sq_passes.sh over an arbitrary python program (no autograd threads under --pmc: rocprofv3 crashes there):
#   bash profiles/micro/sq_passes_cmd.sh OUT_PREFIX script.py [args ...]      -> gpurun_out/OUT_PREFIX_pass_{a,b}.json
set -o pipefail
out=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
prog=$root/$1; shift
cd /tmp && export TMPDIR=/tmp
declare -A groups
groups[a]="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
groups[b]="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
for g in a b; do
  d=/tmp/sq_$g; rm -rf $d
  (cd $root && timeout -k 10 200 rocprofv3 --pmc ${groups[$g]} --kernel-trace --output-format csv -d $d -o x -- python3 $prog "$@" > /dev/null 2> $root/gpurun_out/${out}_pass_$g.err) || { echo "pass $g failed"; tail -3 $root/gpurun_out/${out}_pass_$g.err; continue; }
  f=$(find $d -name "x_counter_collection.csv" | head -1)
  python3 $root/profiles/pmc_sq.py $f $root/gpurun_out/${out}_pass_$g.json > /dev/null && echo "pass $g ok"
done
