#!/bin/bash
run() { python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'dfpn %.2f ms chn %.2f ms' % (d['ms_per_step'], d['chn_train_step']['ms_per_step']), flush=True)"; }
for v in 1 0 1 0; do
  export MT355_CFLAGS_conv_halo="-DMT_HALO_THIN_CK=$v"
  touch master_thesis_amd/csrc/conv_halo.hip
  python3 -c "from master_thesis_amd import build; build.build(verbose=False)" > gpurun_out/thinck_build_$v.log 2>&1
  run "thin_ck=$v"
done
