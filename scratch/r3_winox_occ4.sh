#!/bin/bash
# cost model of four blocks per CU for igemm_winox_kernel<.,1> (lab build with half the halo registers: results are WRONG)
mkdir -p gpurun_out/r3u
timeout -k 10 200 python scratch/r3_winox_abl.py 2>&1 | grep TF > gpurun_out/r3u/occ4_base.txt; echo "== default"; cat gpurun_out/r3u/occ4_base.txt
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-check --no-dfpn > gpurun_out/r3u/occ4_chn_base.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r3u/occ4_chn_base.json')); print('chn base', d['chn_train_step']['ms_per_step'] if 'chn_train_step' in d else d['ms_per_step'])"
touch master_thesis_amd/csrc/conv_halo_wino.hip
MT355_CFLAGS_conv_halo_wino="-DMT_WINOX_OCC1=4 -DMT_WINOX_ABL=6" python -m master_thesis_amd.build > gpurun_out/r3u/build_occ4.txt 2>&1 || { tail -5 gpurun_out/r3u/build_occ4.txt; exit 1; }
timeout -k 10 200 python scratch/r3_winox_abl.py 2>&1 | grep TF > gpurun_out/r3u/occ4_model.txt; echo "== 4 blocks per CU (model)"; cat gpurun_out/r3u/occ4_model.txt
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-check --no-dfpn > gpurun_out/r3u/occ4_chn_model.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r3u/occ4_chn_model.json')); print('chn model', d['chn_train_step']['ms_per_step'] if 'chn_train_step' in d else d['ms_per_step'])"
