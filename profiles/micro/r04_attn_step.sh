#!/bin/bash
tag=$1
out=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -s -m gpu -k "attention" > $out/${tag}_tests_attention.log 2>&1; grep -E "passed|failed|operand scale|Error" $out/${tag}_tests_attention.log | tail -24
timeout -k 10 500 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "attention or flow_backward or full_training_step_matches" > $out/${tag}_tests_train_attn.log 2>&1; tail -3 $out/${tag}_tests_train_attn.log
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --train-steps 0 --no-cpu-baseline > $out/${tag}_bench_c2.json 2> $out/${tag}_bench_c2.err
python - <<PY
import json
j=[json.loads(l) for l in open("$out/${tag}_bench_c2.json") if l.startswith("{")][-1]
print(round(j["value"]), round(j["ms_per_step"],2), j["mean_nats"], j["bpd"], "fallbacks", j["fp16_fallbacks"])
for k in j["kernels"][:5]: print("   ", k["kernel"][:80], k["launches"], round(k["ms_per_step"],2), k["tflops"] and round(k["tflops"],1))
PY
