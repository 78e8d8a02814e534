#!/bin/bash
# same-box A/B: spline_wide.hip with and without -fno-slp-vectorize (forward C2 step and training step)
tag=$1
out=gpurun_out
run() {
  label=$1
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --train-steps 0 --no-cpu-baseline > $out/${tag}_c2_$label.json 2> $out/${tag}_c2_$label.err
  timeout -k 10 300 python bench.py --train --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_train_$label.json 2> $out/${tag}_train_$label.err
  python - <<PY
import json
j=[json.loads(l) for l in open("$out/${tag}_c2_$label.json") if l.startswith("{")][-1]
t=[json.loads(l) for l in open("$out/${tag}_train_$label.json") if l.startswith("{")][-1]
k=[x for x in j["kernels"] if "spline_wide" in x["kernel"]][0]
print("$label: C2", round(j["ms_per_step"],2), "ms; spline_wide", round(k["ms_per_step"],2), "ms/step; train", round(t["ms_per_step"],1), "ms", j["mean_nats"])
PY
}
run slp
FC_EXTRA_FLAGS="spline_wide.hip:-fno-slp-vectorize" python -m flowcompare_amd.build > $out/${tag}_rebuild.log 2>&1 || { echo rebuild failed; tail -5 $out/${tag}_rebuild.log; exit 1; }
run noslp
