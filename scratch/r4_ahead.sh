#!/bin/bash
# round 4: igemm_winox_kernel<.,1> requesting the halo TWO chunks ahead (second register set) -- possible without spills only at two blocks per CU
mkdir -p gpurun_out/r4z
run() { timeout -k 10 200 python scratch/r3_winox_abl.py 2>&1 | grep TF | sed "s/^/$1: /"; }
build() { touch master_thesis_amd/csrc/conv_halo_wino.hip; MT355_CFLAGS_conv_halo_wino="$1" python -m master_thesis_amd.build > gpurun_out/r4z/build_ahead.txt 2>&1 || { tail -3 gpurun_out/r4z/build_ahead.txt; exit 1; }; }
run shipped
build "-DMT_WINOX_OCC=2"; run occ2
build "-DMT_WINOX_OCC=2 -DMT_WINOX_AHEAD=1"; timeout -k 10 120 python -m pytest tests/test_gpu_kernels.py -q -x -k "forward_backward" 2>&1 | tail -n 1; run occ2_ahead
build "-DMT_WINOX_AHEAD=1"; run occ3_ahead_spilling
build "-DMT_WINOX_OCC=2 -DMT_WINOX_AHEAD=1 -DMT_WINOX_PIPE=1"; run occ2_ahead_pipe
build ""
