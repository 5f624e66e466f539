#!/bin/bash
# round 4: border-tile staging of wgrad_pc_kernel without per-load multiplies; A/B against the previous form (git stash of the one file)
run() { timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-chn 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'], 2), [(k, v['ms'], v['tflops']) for k, v in d['kernels'].items() if 'wgrad_pc_kernel<3' in k])"; }
build() { python -m master_thesis_amd.build > gpurun_out/r4z_build_wg.txt 2>&1 || { tail -3 gpurun_out/r4z_build_wg.txt; exit 1; }; }
cp master_thesis_amd/csrc/conv_halo_wgrad.hip /tmp/wg_new.hip
run new
git show HEAD:master_thesis_amd/csrc/conv_halo_wgrad.hip > master_thesis_amd/csrc/conv_halo_wgrad.hip 2>/dev/null || cp scratch/wg_old.hip master_thesis_amd/csrc/conv_halo_wgrad.hip
build; run old
cp /tmp/wg_new.hip master_thesis_amd/csrc/conv_halo_wgrad.hip; build; run new
