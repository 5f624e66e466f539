#!/bin/bash
# round 4: igemm_winox_kernel<.,1> with the next step's image fragments prefetched (MT_WINOX_PIPE), alternating builds on one box
mkdir -p gpurun_out/r4z
run() { timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-dfpn 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'], 2), [(k, v['ms'], v['launches']) for k, v in d['kernels'].items() if 'winox' in k])"; }
build() { touch master_thesis_amd/csrc/conv_halo_wino.hip; MT355_CFLAGS_conv_halo_wino="$1" python -m master_thesis_amd.build > gpurun_out/r4z/build_pipe.txt 2>&1 || { tail -3 gpurun_out/r4z/build_pipe.txt; exit 1; }; }
timeout -k 10 120 python -m pytest tests/test_gpu_kernels.py -q -x -k "forward_backward or upsampled or amax" 2>&1 | tail -n 1
run pipe
build "-DMT_WINOX_PIPE=0"; run off
build ""; run pipe
build "-DMT_WINOX_PIPE=0"; run off
build ""
