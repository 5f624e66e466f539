#!/bin/bash
# round 4: is the multi-process corruption of the conv_tiny kernels tied to their packed-fp32 instructions (v_pk_fma_f32 runs on
# the pipe the other process's MFMAs use)?  Same repro, conv_tiny.hip built with and without SLP vectorisation.
mkdir -p gpurun_out/r4z
build() { touch master_thesis_amd/csrc/conv_tiny.hip; MT355_CFLAGS_conv_tiny="$1" python -m master_thesis_amd.build > gpurun_out/r4z/build_pk.txt 2>&1 || { tail -3 gpurun_out/r4z/build_pk.txt; exit 1; }; }
run() { for i in 1 2; do timeout -k 10 120 python scratch/r4_two_proc_fwd.py 3 40 2>&1 | grep "^rank .: " | sed "s/^/$1 /"; done; }
run packed
build "-fno-slp-vectorize"; run scalar
build ""; run packed
build "-fno-slp-vectorize"; run scalar
build ""
