#!/bin/bash
# round 4: igemm_wino43_kernel without the all-zero second transform round on one-tile images (MT_W43_SKIP_R1), alternating builds
mkdir -p gpurun_out/r4z
run() { timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-chn 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'], 2), [(k, v['ms'], v['tflops']) for k, v in d['kernels'].items() if 'wino43' in k])"; }
build() { touch master_thesis_amd/csrc/conv_halo_wino43.hip; MT355_CFLAGS_conv_halo_wino43="$1" python -m master_thesis_amd.build > gpurun_out/r4z/build_sk.txt 2>&1 || { tail -3 gpurun_out/r4z/build_sk.txt; exit 1; }; }
run skip
build "-DMT_W43_SKIP_R1=0"; run always
build ""; run skip
build "-DMT_W43_SKIP_R1=0"; run always
build ""
