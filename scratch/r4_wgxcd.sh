#!/bin/bash
# round 4: XCD-aware block order of the weight-gradient grids (the channel tiles of one pixel range on one L2) -- A/B on one box
mkdir -p gpurun_out/r4z
build() { touch master_thesis_amd/csrc/conv_halo_wgrad.hip; MT355_CFLAGS_conv_halo_wgrad="$1" python -m master_thesis_amd.build > gpurun_out/r4z/build_wgxcd.txt 2>&1 || { tail -n 3 gpurun_out/r4z/build_wgxcd.txt; exit 1; }; }
run() { timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-check > gpurun_out/r4z/wgxcd_$1.json 2>/dev/null; python - "$1" <<'PY'
import json, sys
d = json.loads(open(f'gpurun_out/r4z/wgxcd_{sys.argv[1]}.json').read().strip().splitlines()[-1])
c = d['chn_train_step']
wg = lambda k: round(sum(v['ms'] for n, v in k.items() if n.startswith('wgrad')), 2)
print(sys.argv[1], 'dfpn %.2f ms (wgrad %.2f)  chn %.2f ms (wgrad %.2f)' % (d['ms_per_step'], wg(d['kernels']), c['ms_per_step'], wg(c['kernels'])), flush=True)
PY
}
timeout -k 10 200 python -m pytest tests/test_gpu_kernels.py -x -q -k "weight_gradient or wgrad or backward" 2>&1 | tail -n 1
run xcd_a
build "-DMT_WG_XCD=0"; run plain_a
build ""; run xcd_b
build "-DMT_WG_XCD=0"; run plain_b
build ""
