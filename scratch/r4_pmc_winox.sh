#!/bin/bash
# round 4: the same four SQ counter passes over igemm_winox_kernel<f16,1> on the dense-block shape the CHN trunk launches most
# (160 -> 32 on 16 images of 64 x 64: 512 blocks, one round at two blocks per CU).
export TMPDIR=/tmp
out=gpurun_out/pmc_winox
mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_WAVES SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_IFETCH SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/p$i -o p --output-format csv -- python3 scratch/r4_winox_shape.py > $out/p$i.log 2>&1 || echo "pass $i failed: $(tail -n 2 $out/p$i.log)"
done
python3 - <<PY
import csv, glob, collections
tot = {}
for f in sorted(glob.glob('$out/p*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(list); dur = []
    for r in csv.DictReader(open(f)):
        if 'winox' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e6)
    print(f.split('/')[2], 'launches', len(dur) // max(1, len(acc)), 'avg ms', sum(dur) / max(1, len(dur)))
    for k, v in acc.items():
        tot[k] = sum(v) / len(v)
        print('   %-32s avg=%.5g' % (k, tot[k]))
w = tot.get('SQ_WAVE_CYCLES')
if w:
    print('per wave-cycle (SQ_WAVE_CYCLES = 1):')
    for k in ('SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_MISC', 'SQ_WAIT_INST_LDS', 'SQ_VALU_MFMA_BUSY_CYCLES'):
        if k in tot:
            print('   %-28s %.3f' % (k, tot[k] / w))
PY
rm -rf $out/p1 $out/p2 $out/p3 $out/p4
