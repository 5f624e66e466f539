#!/bin/bash
# round 4: ablations of igemm_wino43_kernel's main loop (timing only, results are wrong): what would perfect prefetching buy?
mkdir -p gpurun_out/r4z
run() { timeout -k 10 200 python scratch/one_shape.py 2>&1 | grep "fwd" | sed "s/^/$1: /"; timeout -k 10 200 python scratch/r4_one_shape_b.py 2>&1 | grep "fwd" | sed "s/^/$1: /"; }
build() { touch master_thesis_amd/csrc/conv_halo_wino43.hip; MT355_CFLAGS_conv_halo_wino43="$1" python -m master_thesis_amd.build > gpurun_out/r4z/build_abl.txt 2>&1 || { tail -3 gpurun_out/r4z/build_abl.txt; exit 1; }; }
run shipped
build "-DMT_W43_ABL=1"; run no_B_reloads
build "-DMT_W43_ABL=2"; run no_A_reads
build "-DMT_W43_ABL=4"; run no_transform
build "-DMT_W43_ABL=3"; run no_A_no_B
build "-DMT_W43_ABL=7"; run mfma_only
build ""; run shipped
