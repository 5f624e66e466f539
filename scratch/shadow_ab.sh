#!/bin/bash
# A/B of the shadow-safe fp16 split per translation unit (rebuilds on the box)
run() { python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'dfpn %.2f ms chn %.2f ms' % (d['ms_per_step'], d['chn_train_step']['ms_per_step']), flush=True)"; }
run base
for tu in conv_halo conv_halo_wino conv_halo_planes; do
  export MT355_CFLAGS_$tu="-DMT_SHADOW_DEFAULT=1"
  touch master_thesis_amd/csrc/$tu.hip
  python3 -c "from master_thesis_amd import build; build.build(verbose=False)" > gpurun_out/shadow_build_$tu.log 2>&1
  run "+$tu"
done
