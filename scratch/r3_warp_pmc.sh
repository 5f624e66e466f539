#!/bin/bash
# PMC of the warp micro-benchmark (bench.warp_roofline): fabric bytes and L2 hit rate
export TMPDIR=/tmp
out=gpurun_out/r3_warp_pmc
mkdir -p $out
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $set --kernel-trace -d $out/$tag -o p --output-format csv -- python3 scratch/r3_warp_bench.py > $out/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$out/*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'warp_fwd' in r['Kernel_Name']:
            acc[r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
    for g, d in acc.items():
        print(f.split('/')[2], 'grid', g, {k: '%.4g' % (sum(v) / len(v)) for k, v in d.items()}, 'n', len(next(iter(d.values()))))
PY
