#!/bin/bash
# same-box A/B of the row-resident coupling-MLP chain (knob 23): usage  profiles/r03_ab.sh <tag> [extra bench args]
tag=$1; shift
mkdir -p gpurun_out
for v in 1 0; do
  timeout -k 10 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --knob 23=$v "$@" > gpurun_out/${tag}_rows$v.json 2> gpurun_out/${tag}_rows$v.log; echo "bench rows$v rc $?"
done
python - <<PY
import json
for f in ("rows1","rows0"):
    try:
        j=json.load(open(f"gpurun_out/${tag}_{f}.json"))
        print(f, round(j["value"]), round(j["ms_per_step"],2), j["mean_nats"], j["bpd"])
        for k in j["kernels"]: print("   ", k["kernel"][:72], k["launches"], round(k["ms_per_step"],2), k["tflops"] and round(k["tflops"],1))
    except Exception as e: print(f, "ERR", e)
PY
