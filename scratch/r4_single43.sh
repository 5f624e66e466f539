#!/bin/bash
# round 4: igemm_wino43s_kernel (one wave per SIMD, 512 registers, three taps of weight fragments in flight) against the shipped kernel
mkdir -p gpurun_out/r4z
run() { timeout -k 10 200 python scratch/one_shape.py 2>&1 | grep "fwd" | sed "s/^/$1: /"; }
build() { touch master_thesis_amd/csrc/conv_halo_wino43.hip; MT355_CFLAGS_conv_halo_wino43="$1" python -m master_thesis_amd.build > gpurun_out/r4z/build_s43.txt 2>&1 || { tail -3 gpurun_out/r4z/build_s43.txt; exit 1; }; }
run shipped
build "-DMT_W43_SINGLE=1 -DMT_W43S_ABL=4"; run single_no_transform
build "-DMT_W43_SINGLE=1 -DMT_W43S_ABL=5"; run single_mfma_and_A_only
build "-DMT_W43_SINGLE=1 -DMT_W43S_SCHED=1"; run single_nofence
build "-DMT_W43_SINGLE=1 -DMT_W43S_SCHED=2"; run single_groups
build ""; run shipped
