#!/bin/bash
# round 4: the four-instruction fp16 split (scale inside the mixed-precision FMAs) in igemm_winox_kernel -- A/B on one box, bit-identity by checksum
mkdir -p gpurun_out/r4z
build() { touch master_thesis_amd/csrc/conv_halo_wino.hip; MT355_CFLAGS_conv_halo_wino="$1" python -m master_thesis_amd.build > gpurun_out/r4z/build_split3.txt 2>&1 || { tail -n 3 gpurun_out/r4z/build_split3.txt; exit 1; }; }
run() { timeout -k 10 200 python scratch/r4_planar.py nhwc 2>&1 | grep "us " | sed "s/^nhwc/$1/"; }
run shipped
build "-DMT_WINOX_SHADOW=3"; run split3
timeout -k 10 200 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" 2>&1 | tail -n 1
build ""; run shipped
build "-DMT_WINOX_SHADOW=3"; run split3
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-dfpn 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split3 chn', d['ms_per_step'], [(k,v['ms']) for k,v in d['kernels'].items() if 'winox' in k])"
build ""
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-dfpn 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shipped chn', d['ms_per_step'], [(k,v['ms']) for k,v in d['kernels'].items() if 'winox' in k])"
