#!/bin/bash
# round 4: cache policy of igemm_wino43_kernel -- non-temporal halo loads / output stores (leave the L2 to the weight
# fragments) and n-tile-major block order inside an XCD (one n-tile's fragments in flight); alternating builds on one box
mkdir -p gpurun_out/r4z
run() { timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check --no-chn 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'], 2), [(k, v['ms'], v['tflops']) for k, v in d['kernels'].items() if 'wino43' in k])"; }
build() { touch master_thesis_amd/csrc/conv_halo_wino43.hip; MT355_CFLAGS_conv_halo_wino43="$1" python -m master_thesis_amd.build > gpurun_out/r4z/build_c43.txt 2>&1 || { tail -3 gpurun_out/r4z/build_c43.txt; exit 1; }; }
run default
build "-DMT_W43_NT_LOAD=1"; run nt_load
build "-DMT_W43_NT_LOAD=1 -DMT_W43_NT_STORE=1"; run nt_load_store
build "-DMT_W43_NT_STORE=1"; run nt_store
build "-DMT_W43_NTILE_MAJOR=1"; run ntile_major
build "-DMT_W43_NTILE_MAJOR=1 -DMT_W43_NT_LOAD=1 -DMT_W43_NT_STORE=1"; run ntile_major_nt
build ""; run default
