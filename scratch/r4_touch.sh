#!/bin/bash
# round 4: igemm_winox_kernel<.,1> touching the halo of chunk c + 2 during chunk c (L2 prefetch, MT_WINOX_TOUCH), alternating builds
mkdir -p gpurun_out/r4z
run() { timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-dfpn 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'], 2), [(k, v['ms'], v['launches']) for k, v in d['kernels'].items() if 'winox' in k])"; }
build() { touch master_thesis_amd/csrc/conv_halo_wino.hip; MT355_CFLAGS_conv_halo_wino="$1" python -m master_thesis_amd.build > gpurun_out/r4z/build_touch.txt 2>&1 || { tail -3 gpurun_out/r4z/build_touch.txt; exit 1; }; }
run touch
build "-DMT_WINOX_TOUCH=0"; run off
build ""; timeout -k 10 120 python -m pytest tests/test_gpu_kernels.py -q -x -k "forward_backward" 2>&1 | tail -n 1; run touch
build "-DMT_WINOX_TOUCH=0"; run off
build ""
