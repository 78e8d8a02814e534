#!/bin/bash
# one GPU call of the round-4 loop: wide Linear op tests, flow / full-size tests, a C2 bench line with the per-kernel breakdown
tag=$1
out=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -s -m gpu -k "wide_linear or mlp" > $out/${tag}_tests_wide_linear.log 2>&1; grep -E "passed|failed|Error|rows .* hidden" $out/${tag}_tests_wide_linear.log | tail
timeout -k 10 400 python -m pytest tests/test_gpu_flow.py tests/test_gpu_fullsize.py -x -q -m gpu > $out/${tag}_tests_flow.log 2>&1; tail -5 $out/${tag}_tests_flow.log
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --train-steps 0 --no-cpu-baseline > $out/${tag}_bench_c2.json 2> $out/${tag}_bench_c2.err
python - <<PY
import json
j=[json.loads(l) for l in open("$out/${tag}_bench_c2.json") if l.startswith("{")][-1]
print(round(j["value"]), round(j["ms_per_step"],2), j["mean_nats"], j["bpd"])
for k in j["kernels"]: print("   ", k["kernel"][:80], k["launches"], round(k["ms_per_step"],2), k["tflops"] and round(k["tflops"],1))
PY
