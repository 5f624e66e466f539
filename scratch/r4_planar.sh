#!/bin/bash
# round 4: chunk-major ("planar") block buffer for igemm_winox_kernel -- a timing probe before any layout work
mkdir -p gpurun_out/r4z
build() { touch master_thesis_amd/csrc/conv_halo_wino.hip; MT355_CFLAGS_conv_halo_wino="$1" python -m master_thesis_amd.build > gpurun_out/r4z/build_planar.txt 2>&1 || { tail -n 3 gpurun_out/r4z/build_planar.txt; exit 1; }; }
timeout -k 10 200 python scratch/r4_planar.py nhwc 2>&1 | grep "us "
build "-DMT_WINOX_PLANAR=1"; timeout -k 10 200 python scratch/r4_planar.py planar 2>&1 | grep "us "
build ""; timeout -k 10 200 python scratch/r4_planar.py nhwc 2>&1 | grep "us "
