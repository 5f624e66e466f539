#!/bin/bash
mkdir -p gpurun_out/r3z
touch master_thesis_amd/csrc/conv_halo_wino43.hip
MT355_CFLAGS_conv_halo_wino43="-DMT_W43_TRACE=1" python -m master_thesis_amd.build > gpurun_out/r3z/build.txt 2>&1 || { tail -5 gpurun_out/r3z/build.txt; exit 1; }
timeout -k 10 300 python scratch/r3_w43_trace.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3z/w43_trace.txt
