#!/bin/bash
# one GPU call of the round-4 training loop: the wide-path op test, the training tests, a training bench line (and its A/B with knob 31 = 0)
tag=$1
out=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_train.py -x -q -s -m gpu -k "wide_loop or spline_forward or linear_act or mlp_at" > $out/${tag}_tests_train_wide.log 2>&1; grep -E "passed|failed|Error|wide loop|fp32-A loop|assert" $out/${tag}_tests_train_wide.log | tail -20
timeout -k 10 300 python bench.py --train --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_bench_train.json 2> $out/${tag}_bench_train.err; tail -c 1500 $out/${tag}_bench_train.json
timeout -k 10 300 python bench.py --train --steps 3 --warmup 1 --no-cpu-baseline --knob 31=0 > $out/${tag}_bench_train_k31_0.json 2> $out/${tag}_bench_train_k31_0.err; tail -c 600 $out/${tag}_bench_train_k31_0.json
