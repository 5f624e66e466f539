#!/bin/bash
# batched weight gradients of the RRDB trunk: pixel ranges chosen by rounds of one block per CU vs the block target
for v in ${SPLITS_LIST:-old new old new 4 6}; do
  if [ "$v" = new ]; then unset MT355_WGRAD_BATCH_SPLITS; else export MT355_WGRAD_BATCH_SPLITS=$v; fi
  timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-check --no-dfpn 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; w=[v for n,v in k.items() if 'x60' in n or 'x30' in n]; print('$v', 'chn %.2f ms' % d['ms_per_step'], 'batched wgrad', [(x['launches'], x['ms'], x['tflops']) for x in w])"
done
