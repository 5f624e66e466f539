#!/bin/bash
mkdir -p gpurun_out/r3z
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv or amax" > gpurun_out/r3z/t.txt 2>&1; tail -2 gpurun_out/r3z/t.txt
for i in 1 2; do timeout -k 10 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-check 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['chn_train_step']['ms_per_step'], [(k, v['ms'], v['tflops']) for k, v in d['kernels'].items() if 'wino43' in k or 'wino25' in k])"; done
bash scratch/r3_w43_trace.sh 2>&1 | grep "==\|medians" | head -4
