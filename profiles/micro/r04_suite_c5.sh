#!/bin/bash
tag=$1
out=gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/${tag}_gpu_suite.log 2>&1; tail -4 $out/${tag}_gpu_suite.log
grep -E "passed|failed" $out/${tag}_gpu_suite.log | tail -1 | grep -q failed && exit 1
for cfg in "c4_dgcnn_attn_extra_affine 16 16384 c5 --steps 3 --warmup 1" "c4_dgcnn_attn_extra_affine 8 4096 c4" "c3_paconv_attn_affine 16 4096 c3"; do
  set -- $cfg
  name=$4
  timeout -k 10 400 python bench.py --config $1 --batch $2 --points $3 --train-steps 0 --no-cpu-baseline $5 $6 $7 $8 > $out/${tag}_bench_$name.json 2> $out/${tag}_bench_$name.err
  python - <<PY
import json
j=[json.loads(l) for l in open("$out/${tag}_bench_$name.json") if l.startswith("{")][-1]
a=[k for k in j["kernels"] if "attn16" in k["kernel"]]
print("$name", round(j["value"]), round(j["ms_per_step"],2), "attention", a and round(a[0]["ms_per_step"],2), "fallbacks", j["fp16_fallbacks"])
PY
done
