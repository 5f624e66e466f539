#!/bin/bash
# usage: ab_env.sh VAR A B [bench args]: the bench with VAR=A, VAR=B, VAR=A, VAR=B on one box
var=$1; a=$2; b=$3; shift 3
for v in $a $b $a $b; do
  env $var=$v python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); c=d.get('chn_train_step'); print('$var=$v', 'dfpn %.2f ms' % d['ms_per_step'], ('chn %.2f ms' % c['ms_per_step']) if c else '', flush=True)"
done
