#!/bin/bash
mkdir -p gpurun_out/r3u
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -x -q > gpurun_out/r3u/amax2_tests.txt 2>&1; tail -3 gpurun_out/r3u/amax2_tests.txt
timeout -k 10 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-check > gpurun_out/r3u/amax2_bench.json 2> gpurun_out/r3u/amax2_bench.err; tail -2 gpurun_out/r3u/amax2_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r3u/amax2_bench.json')); print('dfpn', d['ms_per_step'], 'chn', d['chn_train_step']['ms_per_step'])
for nm, c in (('dfpn', d), ('chn', d['chn_train_step'])):
    ks = c['kernels'] if nm == 'chn' else d.get('kernels', {})
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1]['ms'])[:8]: print('  ', nm, k, v)
"
