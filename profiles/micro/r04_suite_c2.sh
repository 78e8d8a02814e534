#!/bin/bash
tag=$1
out=gpurun_out
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --train-steps 0 --no-cpu-baseline > $out/${tag}_bench_c2.json 2> $out/${tag}_bench_c2.err
python - <<PY
import json
j=[json.loads(l) for l in open("$out/${tag}_bench_c2.json") if l.startswith("{")][-1]
print(round(j["value"]), round(j["ms_per_step"],2), j["mean_nats"], j["bpd"], "fallbacks", j["fp16_fallbacks"])
for k in j["kernels"][:5]: print("   ", k["kernel"][:80], k["launches"], round(k["ms_per_step"],2), k["tflops"] and round(k["tflops"],1))
PY
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/${tag}_gpu_suite.log 2>&1; tail -4 $out/${tag}_gpu_suite.log
