#!/bin/bash
# same-box A/B of one knob: usage  profiles/r03_ab.sh <tag> <knob> <value A> <value B> [extra bench args]
tag=$1; knob=$2; va=$3; vb=$4; shift 4
mkdir -p gpurun_out
for v in $va $vb; do
  timeout -k 10 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --train-steps 0 --knob $knob=$v "$@" > gpurun_out/${tag}_k${knob}_$v.json 2> gpurun_out/${tag}_k${knob}_$v.log; echo "bench knob $knob=$v rc $?"
done
python - <<PY
import json
for v in ("$va","$vb"):
    try:
        j=[json.loads(l) for l in open(f"gpurun_out/${tag}_k${knob}_{v}.json") if l.startswith("{")][-1]
        print("knob $knob =", v, round(j["value"]), round(j["ms_per_step"],2), j["mean_nats"], j["bpd"])
        for k in j["kernels"]: print("   ", k["kernel"][:72], k["launches"], round(k["ms_per_step"],2), k["tflops"] and round(k["tflops"],1))
    except Exception as e: print(v, "ERR", e)
PY
