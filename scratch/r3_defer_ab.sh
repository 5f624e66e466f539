#!/bin/bash
# same-box A/B: folding the amax slots in a launch of their own (grids of >= 1024 blocks) vs the in-kernel election everywhere
mkdir -p gpurun_out/r3z
run() { for i in 1 2; do timeout -k 10 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-check 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['chn_train_step']['ms_per_step'])"; done; }
run deferred
touch master_thesis_amd/csrc/conv_halo_wino.hip master_thesis_amd/csrc/conv_halo_wino25.hip master_thesis_amd/csrc/conv_halo_wino43.hip master_thesis_amd/csrc/conv_halo_planes.hip
MT355_CFLAGS="-DMT_AMAX_DEFER_BLOCKS=2000000000" python -m master_thesis_amd.build > gpurun_out/r3z/build_ab.txt 2>&1 || { tail -3 gpurun_out/r3z/build_ab.txt; exit 1; }
run in-kernel
