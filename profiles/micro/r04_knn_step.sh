#!/bin/bash
tag=$1
out=gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "knn" > $out/${tag}_tests_knn.log 2>&1; tail -4 $out/${tag}_tests_knn.log
timeout -k 10 400 python -m pytest tests/test_gpu_flow.py tests/test_gpu_configs.py -x -q -m gpu > $out/${tag}_tests_flow.log 2>&1; tail -3 $out/${tag}_tests_flow.log
for kn in "" "--knob 32=0"; do
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --train-steps 0 --no-cpu-baseline $kn > $out/${tag}_bench_c2.json 2> $out/${tag}_bench_c2.err
python - <<PY
import json
j=[json.loads(l) for l in open("$out/${tag}_bench_c2.json") if l.startswith("{")][-1]
print("$kn", round(j["value"]), round(j["ms_per_step"],2), j["mean_nats"], j["bpd"])
for k in j["kernels"]:
    if "knn" in k["kernel"]: print("   ", k["kernel"][:80], k["launches"], round(k["ms_per_step"],2))
PY
done
timeout -k 10 300 python bench.py --config c4_dgcnn_attn_extra_affine --batch 16 --points 16384 --steps 2 --warmup 1 --train-steps 0 --no-cpu-baseline > $out/${tag}_bench_c5.json 2> $out/${tag}_bench_c5.err
python - <<PY
import json
j=[json.loads(l) for l in open("$out/${tag}_bench_c5.json") if l.startswith("{")][-1]
print("C5", round(j["value"]), round(j["ms_per_step"],2))
for k in j["kernels"]:
    if "knn" in k["kernel"] or "attn" in k["kernel"]: print("   ", k["kernel"][:80], k["launches"], round(k["ms_per_step"],2))
PY
