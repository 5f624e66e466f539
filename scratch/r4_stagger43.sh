#!/bin/bash
# round 4: the two xi halves of igemm_wino43_kernel (the two waves of every SIMD) do their transform at different taps
mkdir -p gpurun_out/r4z
run() { timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-chn 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'], 2), [(k, v['ms'], v['tflops']) for k, v in d['kernels'].items() if 'wino43' in k])"; }
build() { touch master_thesis_amd/csrc/conv_halo_wino43.hip; MT355_CFLAGS_conv_halo_wino43="$1" python -m master_thesis_amd.build > gpurun_out/r4z/build_st.txt 2>&1 || { tail -3 gpurun_out/r4z/build_st.txt; exit 1; }; }
run default
build "-DMT_W43_STAGGER=1"; timeout -k 10 120 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3x3 or forward_backward" 2>&1 | tail -n 1; run stagger
build ""; run default
build "-DMT_W43_STAGGER=1"; run stagger
build ""
