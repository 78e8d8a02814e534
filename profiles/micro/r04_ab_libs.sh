#!/bin/bash
# same-box A/B of two builds of the library: profiles/micro/ab/libfcflow_A.so and libfcflow_B.so (git-ignored, built here, travel with gpurun)
# usage: r04_ab_libs.sh TAG [pytest -k expression for lib B]
tag=$1
out=gpurun_out
for v in A B A B; do
  cp profiles/micro/ab/libfcflow_$v.so flowcompare_amd/libfcflow.so
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --train-steps 0 --no-cpu-baseline > $out/${tag}_${v}_bench_c2.json 2> $out/${tag}_${v}_bench_c2.err || exit 1
  python - <<PY
import json
j=[json.loads(l) for l in open("$out/${tag}_${v}_bench_c2.json") if l.startswith("{")][-1]
print("$v", round(j["value"]), round(j["ms_per_step"],2), j["mean_nats"], j["bpd"], "fallbacks", j["fp16_fallbacks"])
for k in j["kernels"][:4]: print("   ", k["kernel"][:60], k["launches"], round(k["ms_per_step"],2))
PY
done
for v in A B; do
  cp profiles/micro/ab/libfcflow_$v.so flowcompare_amd/libfcflow.so
  timeout -k 10 300 python bench.py --config c4_dgcnn_attn_extra_affine --batch 16 --points 16384 --steps 3 --warmup 1 --train-steps 0 --no-cpu-baseline > $out/${tag}_${v}_bench_c5.json 2> $out/${tag}_${v}_bench_c5.err || exit 1
  python - <<PY
import json
j=[json.loads(l) for l in open("$out/${tag}_${v}_bench_c5.json") if l.startswith("{")][-1]
print("$v C5", round(j["value"]), round(j["ms_per_step"],2))
for k in j["kernels"][:3]: print("   ", k["kernel"][:60], k["launches"], round(k["ms_per_step"],2))
PY
done
cp profiles/micro/ab/libfcflow_B.so flowcompare_amd/libfcflow.so
if [ -n "$2" ]; then timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_flow.py -x -q -s -m gpu -k "$2" > $out/${tag}_B_tests.log 2>&1; grep -E "passed|failed|ramp|Error" $out/${tag}_B_tests.log | tail -12; fi
