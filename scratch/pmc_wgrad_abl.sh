#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/pmc_wgabl
mkdir -p $out
for a in 0 1 5; do
  export MT355_WGRAD_ABL=$a
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace -d $out/a$a -o p --output-format csv -- python3 scratch/one_wgrad.py > $out/a$a.log 2>&1 || echo "pass $a failed"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$out/a*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(list); dur = []
    for r in csv.DictReader(open(f)):
        if 'wgrad_pc' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
    n = len(acc['GRBM_GUI_ACTIVE'])
    d = sum(dur) / max(1, len(dur))
    print(f.split('/')[2], 'avg us %.1f' % d, 'n', n, {k: '%.4g' % (sum(v) / len(v)) for k, v in acc.items()})
    g = sum(acc['GRBM_GUI_ACTIVE']) / n
    print('   clock GHz if /8: %.3f   mfma busy frac %.3f' % (g / 8 / d / 1e3, sum(acc['SQ_VALU_MFMA_BUSY_CYCLES']) / n / (1024 * g / 8)))
PY
