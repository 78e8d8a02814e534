#!/bin/bash
# training bench A/B runs on one box: $1 = tag, then pairs "label|extra flags|env"
tag=$1; shift
out=gpurun_out
for spec in "$@"; do
  IFS='|' read -r label flags envs <<< "$spec"
  env $envs timeout -k 10 300 python bench.py --train --steps 3 --warmup 1 --no-cpu-baseline $flags > $out/${tag}_train_${label}.json 2> $out/${tag}_train_${label}.err || { echo "$label failed"; tail -3 $out/${tag}_train_${label}.err; exit 1; }
  python - <<PY
import json
j=[json.loads(l) for l in open("$out/${tag}_train_${label}.json") if l.startswith("{")][-1]
print("$label", round(j["ms_per_step"],1), "ms  peak", round(j["peak_mem_GiB"],1), "GiB  loss", j["loss"], " fallbacks", j["fp16_fallbacks"])
PY
  grep "train_flow\]" $out/${tag}_train_${label}.err | tail -1
done
