#!/bin/bash
mkdir -p gpurun_out/r3x
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv or amax" > gpurun_out/r3x/t.txt 2>&1; tail -2 gpurun_out/r3x/t.txt
timeout -k 10 200 python scratch/r3_winox_bench.py 2>&1 | grep TF
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-check --no-dfpn > gpurun_out/r3x/chn.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r3x/chn.json')); print('chn', d['ms_per_step'], [(k, v['ms']) for k, v in d['kernels'].items() if 'winox' in k])"

