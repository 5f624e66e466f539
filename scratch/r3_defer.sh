#!/bin/bash
mkdir -p gpurun_out/r3z
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv or amax" > gpurun_out/r3z/t.txt 2>&1; tail -2 gpurun_out/r3z/t.txt
timeout -k 10 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-check > gpurun_out/r3z/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r3z/bench.json')); print('dfpn', d['ms_per_step'], 'chn', d['chn_train_step']['ms_per_step'])
for nm, ks in (('dfpn', d['kernels']), ('chn', d['chn_train_step']['kernels'])):
    print(nm, [(k, v['ms'], v['tflops']) for k, v in ks.items() if 'wino43' in k or 'wino25' in k or 'wino16' in k])"
bash scratch/r3_w43_trace.sh
