#!/bin/bash
mkdir -p gpurun_out/r3u
for v in 0 1; do
touch master_thesis_amd/csrc/conv_halo_wino.hip
MT355_CFLAGS_conv_halo_wino="-DMT_WINOX_TRACE=1 -DMT_LEAN_NT=$v" python -m master_thesis_amd.build > gpurun_out/r3u/build_nt.txt 2>&1 || { tail -5 gpurun_out/r3u/build_nt.txt; exit 1; }
echo "#### MT_LEAN_NT=$v"
timeout -k 10 200 python scratch/r3_winox_trace2.py 2>&1 | grep -v amdgpu.ids | grep "==\|slowest 25\|epilogue:" 
timeout -k 10 200 python scratch/r3_winox_abl.py 2>&1 | grep TF
done
