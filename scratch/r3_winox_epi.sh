#!/bin/bash
# two-phase epilogue of igemm_winox_kernel: correctness, layer timings, block timeline, CHN step
mkdir -p gpurun_out/r3u
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" > gpurun_out/r3u/epi_tests.txt 2>&1; tail -2 gpurun_out/r3u/epi_tests.txt
timeout -k 10 200 python scratch/r3_winox_bench.py 2>&1 | grep TF
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-check --no-dfpn > gpurun_out/r3u/epi_chn.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r3u/epi_chn.json')); print('chn', d['ms_per_step'])"
touch master_thesis_amd/csrc/conv_halo_wino.hip
MT355_CFLAGS_conv_halo_wino="-DMT_WINOX_TRACE=1" python -m master_thesis_amd.build > gpurun_out/r3u/build_epi.txt 2>&1 || { tail -5 gpurun_out/r3u/build_epi.txt; exit 1; }
timeout -k 10 200 python scratch/r3_winox_trace2.py 2>&1 | grep -v amdgpu.ids
