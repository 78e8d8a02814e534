#!/bin/bash
tag=$1
out=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" > $out/${tag}_tests_attention.log 2>&1; tail -2 $out/${tag}_tests_attention.log
for kn in "30=1" "30=0"; do
for cfg in "c2_dgcnn_attn_spline 16 4096 c2 --steps 10 --warmup 3" "c4_dgcnn_attn_extra_affine 16 16384 c5 --steps 3 --warmup 1" "c4_dgcnn_attn_extra_affine 8 4096 c4 --steps 10 --warmup 3"; do
  set -- $cfg
  name=$4
  timeout -k 10 400 python bench.py --config $1 --batch $2 --points $3 --train-steps 0 --no-cpu-baseline --knob $kn $5 $6 $7 $8 > $out/${tag}_bench_${name}_k$kn.json 2> $out/${tag}_bench_${name}_k$kn.err
  python - <<PY
import json
j=[json.loads(l) for l in open("$out/${tag}_bench_${name}_k$kn.json") if l.startswith("{")][-1]
a=[k for k in j["kernels"] if "attn16" in k["kernel"]]
print("knob $kn", "$name", round(j["value"]), round(j["ms_per_step"],2), "attention", a and round(a[0]["ms_per_step"],2), j["mean_nats"])
PY
done
done
