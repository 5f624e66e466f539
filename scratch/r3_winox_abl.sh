#!/bin/bash
# where the time of igemm_winox_kernel goes: ablations (lab builds; results are wrong by construction)
mkdir -p gpurun_out/r3u
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" > gpurun_out/r3u/abl_tests.txt 2>&1; tail -2 gpurun_out/r3u/abl_tests.txt
timeout -k 10 200 python scratch/r3_winox_abl.py 2>&1 | grep TF > gpurun_out/r3u/abl0.txt; echo "== default"; cat gpurun_out/r3u/abl0.txt
for a in 1 2 3 4 5; do
  touch master_thesis_amd/csrc/conv_halo_wino.hip
  MT355_CFLAGS_conv_halo_wino="-DMT_WINOX_ABL=$a" python -m master_thesis_amd.build > gpurun_out/r3u/build$a.txt 2>&1 || { echo build $a failed; tail -5 gpurun_out/r3u/build$a.txt; exit 1; }
  timeout -k 10 200 python scratch/r3_winox_abl.py 2>&1 | grep TF > gpurun_out/r3u/abl$a.txt; echo "== ablation $a"; cat gpurun_out/r3u/abl$a.txt
done
