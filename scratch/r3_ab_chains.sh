#!/bin/bash
# A/B on one box: the RRDB trunk as 1 / 2 / 4 / 8 launch chains on as many streams (MT355_RRDB_CHAINS), CHN step only
for v in ${CHAINS_LIST:-1 2 4 8 2 4}; do
  for mode in "" ; do
    MT355_RRDB_CHAINS=$v python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check --no-dfpn $mode 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('chains=$v $mode', 'chn %.2f ms' % d['ms_per_step'], flush=True)"
  done
done
