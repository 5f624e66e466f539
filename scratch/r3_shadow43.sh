#!/bin/bash
# split form of the in-loop transform of the 512-thread Winograd kernels (F(4,3), F(2,5)): hipcc's own against the
# five-instruction form (conv_common.h: split4f SHADOW = 2); same box, alternating
mkdir -p gpurun_out/r3z
run() { timeout -k 10 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-check 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'], 2), round(d['chn_train_step']['ms_per_step'], 2), [(k, v['ms']) for k, v in d['kernels'].items() if 'wino43' in k or 'wino25' in k])"; }
build() { touch master_thesis_amd/csrc/conv_halo_wino43.hip master_thesis_amd/csrc/conv_halo_wino25.hip; MT355_CFLAGS_conv_halo_wino43="$1" MT355_CFLAGS_conv_halo_wino25="$1" python -m master_thesis_amd.build > gpurun_out/r3z/build_s.txt 2>&1 || { tail -3 gpurun_out/r3z/build_s.txt; exit 1; }; }
run default
build "-DMT_SHADOW_DEFAULT=2"; run shadow2
build ""; run default
build "-DMT_SHADOW_DEFAULT=2"; run shadow2
