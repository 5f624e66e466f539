#!/bin/bash
# round 4: igemm_wino43_kernel's in-loop transform + split WITHOUT the VALU instructions that share the matrix pipe
# (v_pk_*_f32, v_fma_mix*): alternating builds on one box
mkdir -p gpurun_out/r4z
run() { timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-chn 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'], 2), [(k, v['ms'], v['tflops']) for k, v in d['kernels'].items() if 'wino43' in k or 'wino25' in k])"; }
build() { touch master_thesis_amd/csrc/conv_halo_wino43.hip master_thesis_amd/csrc/conv_halo_wino25.hip; MT355_CFLAGS_conv_halo_wino43="$1" MT355_CFLAGS_conv_halo_wino25="${1//W43/W25}" python -m master_thesis_amd.build > gpurun_out/r4z/build_sc.txt 2>&1 || { tail -3 gpurun_out/r4z/build_sc.txt; exit 1; }; }
run default
build "-DMT_W43_SCALAR_XFORM=1 -DMT_W43_SPLIT=1 -fno-slp-vectorize"; run scalar_all
build "-DMT_W43_SPLIT=1"; run split1_only
build ""; run default
build "-DMT_W43_SCALAR_XFORM=1 -DMT_W43_SPLIT=1 -fno-slp-vectorize"; run scalar_all
build "-DMT_W43_SCALAR_XFORM=1 -fno-slp-vectorize"; run scalar_xform_only
build ""
