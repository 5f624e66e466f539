#!/bin/bash
# round 4: the successor-tile L2 prefetch of igemm_wino43_kernel (MT_W43_PREFETCH), alternating builds on one box
mkdir -p gpurun_out/r4z
run() { timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-chn 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'], 2), [(k, v['ms'], v['tflops']) for k, v in d['kernels'].items() if 'wino43' in k])"; }
build() { touch master_thesis_amd/csrc/conv_halo_wino43.hip; MT355_CFLAGS_conv_halo_wino43="$1" python -m master_thesis_amd.build > gpurun_out/r4z/build_pf.txt 2>&1 || { tail -3 gpurun_out/r4z/build_pf.txt; exit 1; }; }
run prefetch32
build "-DMT_W43_PREFETCH=0"; run off
build "-DMT_W43_PREFETCH_AHEAD=64"; run prefetch64
build "-DMT_W43_PREFETCH=0"; run off
build ""; run prefetch32
build "-DMT_W43_PREFETCH_AHEAD=16"; run prefetch16
build ""
