#!/bin/bash
for pad in 8 32 16; do
  export MT355_CFLAGS_conv_halo_wgrad="-DMT_WGPC_PAD=$pad"
  touch master_thesis_amd/csrc/conv_halo_wgrad.hip
  python3 -c "from master_thesis_amd import build; build.build(verbose=False)" > gpurun_out/wgpad_build_$pad.log 2>&1
  TAG=pad$pad python3 scratch/wgrad_lab.py 2>&1 | grep wgrad
done
