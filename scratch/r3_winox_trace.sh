#!/bin/bash
mkdir -p gpurun_out/r3u
touch master_thesis_amd/csrc/conv_halo_wino.hip
MT355_CFLAGS_conv_halo_wino="-DMT_WINOX_TRACE=1 $WX_EXTRA" python -m master_thesis_amd.build > gpurun_out/r3u/build_trace.txt 2>&1 || { tail -5 gpurun_out/r3u/build_trace.txt; exit 1; }
timeout -k 10 200 python ${WX_SCRIPT:-scratch/r3_winox_trace.py} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3u/trace.txt
