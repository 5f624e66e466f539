#!/bin/bash
# occupancy sensitivity of igemm_winox_kernel: default (3 blocks per CU for TN = 1, 2 for TN = 2) vs one block fewer
set -e
mkdir -p gpurun_out/r3u
timeout -k 10 200 python scratch/r3_winox_bench.py > gpurun_out/r3u/occ_default.txt 2>&1
touch master_thesis_amd/csrc/conv_halo_wino.hip
MT355_CFLAGS_conv_halo_wino="-DMT_WINOX_LDS_PAD=14000" python -m master_thesis_amd.build > gpurun_out/r3u/build.txt 2>&1
timeout -k 10 200 python scratch/r3_winox_bench.py > gpurun_out/r3u/occ_minus1.txt 2>&1
grep TF gpurun_out/r3u/occ_default.txt; echo ---; grep TF gpurun_out/r3u/occ_minus1.txt
