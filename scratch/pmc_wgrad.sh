#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/pmc_wgrad
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" "SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/p$i -o p --output-format csv -- python3 scratch/one_wgrad.py > $out/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$out/p*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(list); dur = []
    for r in csv.DictReader(open(f)):
        if 'wgrad_pc_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
    print(f.split('/')[2], 'avg us', sum(dur) / max(1, len(dur)), 'n', len(dur))
    for k, v in acc.items():
        print('  ', k, 'avg=%.4g' % (sum(v) / len(v)))
PY
