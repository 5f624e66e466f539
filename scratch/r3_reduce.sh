#!/bin/bash
mkdir -p gpurun_out/r3x; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv or weight or batched" > gpurun_out/r3x/t.txt 2>&1; tail -2 gpurun_out/r3x/t.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/r3x/stats -o b --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check > gpurun_out/r3x/bench.json 2> gpurun_out/r3x/err.txt
python3 - <<'PY'
import csv, glob, json
f = glob.glob('gpurun_out/r3x/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('wgrad_reduce', 'l1_fwd', 'sobel_l1_fwd', 'absmax', 'axpby')):
        print(r['Name'][:50], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us', round(float(r['TotalDurationNs']) / 1e6, 1), 'ms')
d = json.load(open('gpurun_out/r3x/bench.json')); print('dfpn', d['ms_per_step'], 'chn', d['chn_train_step']['ms_per_step'])
PY
