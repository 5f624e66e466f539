#!/bin/bash
# round 4: matrix-pipe busy share and sustained clock of igemm_wino43_kernel as shipped and with its feeding work ablated
export TMPDIR=/tmp
mkdir -p gpurun_out/r4z
build() { touch master_thesis_amd/csrc/conv_halo_wino43.hip; MT355_CFLAGS_conv_halo_wino43="$1" python -m master_thesis_amd.build > gpurun_out/r4z/build_ac.txt 2>&1 || { tail -3 gpurun_out/r4z/build_ac.txt; exit 1; }; }
meas() {
  rm -rf gpurun_out/r4z/pmc_ac
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/r4z/pmc_ac -o p --output-format csv -- python3 scratch/one_shape.py > gpurun_out/r4z/pmc_ac.log 2>&1
  python3 - "$1" <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/r4z/pmc_ac/**/*counter_collection.csv', recursive=True)[0]
busy = act = dur = n = 0
for r in csv.DictReader(open(f)):
    if 'wino43' not in r['Kernel_Name']:
        continue
    if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES':
        busy += float(r['Counter_Value']); dur += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e6; n += 1
    elif r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        act += float(r['Counter_Value'])
print('%-14s %.3f ms per launch, matrix pipe %.3f busy, clock %.2f GHz' % (sys.argv[1], dur / n, busy / (act / 8 * 1024), act / 8 / (dur * 1e-3) / 1e9))
PY
}
meas shipped
build "-DMT_W43_ABL=7"; meas mfma_only
build "-DMT_W43_ABL=4"; meas no_transform
build "-DMT_W43_ABL=1"; meas no_B_reloads
build ""
