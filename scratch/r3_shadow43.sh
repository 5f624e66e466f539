#!/bin/bash
# split form of the in-loop transform of the 512-thread Winograd kernels: hipcc's (packed / mixed ops share the matrix pipe),
# SHADOW = 1 (plain VALU only, 8 per pair), SHADOW = 2 (5 per pair, two of them mixed)
mkdir -p gpurun_out/r3z
run() { timeout -k 10 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-check 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['chn_train_step']['ms_per_step'], [(k, v['ms']) for k, v in d['kernels'].items() if 'wino43' in k or 'wino25' in k])"; }
run default
for v in 1 2; do
  touch master_thesis_amd/csrc/conv_halo_wino43.hip master_thesis_amd/csrc/conv_halo_wino25.hip
  MT355_CFLAGS_conv_halo_wino43="-DMT_SHADOW_DEFAULT=$v" MT355_CFLAGS_conv_halo_wino25="-DMT_SHADOW_DEFAULT=$v" python -m master_thesis_amd.build > gpurun_out/r3z/build_s.txt 2>&1 || { tail -3 gpurun_out/r3z/build_s.txt; exit 1; }
  run shadow$v
done
