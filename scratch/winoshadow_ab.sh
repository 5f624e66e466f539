#!/bin/bash
for sh in 0 1 0 1; do
  export MT355_CFLAGS_conv_halo_wino="-DMT_SHADOW_DEFAULT=$sh"
  touch master_thesis_amd/csrc/conv_halo_wino.hip
  python3 -c "from master_thesis_amd import build; build.build(verbose=False)" > gpurun_out/ws_build_$sh.log 2>&1
  TAG=shadow$sh python3 scratch/layer_lab.py 2>&1 | grep -E "k3" | head -4
done
