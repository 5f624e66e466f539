#!/bin/bash
# round 4: 4 x 16-pixel tiles for the 32-wide xi-split kernel (twice the blocks, four per CU, half the chain per block) -- A/B on one box
mkdir -p gpurun_out/r4z
build() { touch master_thesis_amd/csrc/conv_halo_wino.hip; MT355_CFLAGS_conv_halo_wino="$1" python -m master_thesis_amd.build > gpurun_out/r4z/build_th4.txt 2>&1 || { tail -n 3 gpurun_out/r4z/build_th4.txt; exit 1; }; }
run() { timeout -k 10 200 python scratch/r4_planar.py nhwc 2>&1 | grep "us " | sed "s/^nhwc/$1/"; }
chn() { timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-dfpn 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 chn', round(d['ms_per_step'],2), [(k,v['ms']) for k,v in d['kernels'].items() if 'winox' in k])"; }
run shipped; chn shipped
build "-DMT_WINOX_TH4=1"; run th4
timeout -k 10 200 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" 2>&1 | tail -n 1
timeout -k 10 200 python -m pytest tests/test_gpu_models.py -x -q -k "chn or rrdb" 2>&1 | tail -n 1
chn th4
build "-DMT_WINOX_TH4=1 -DMT_WINOX_TH4_BELOW=600"; chn th4_below600
build ""; chn shipped
