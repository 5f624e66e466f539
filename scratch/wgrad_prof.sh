#!/bin/bash
# kernel-level times of the weight-gradient lab (rocprofv3 --stats)
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/wgp -o s --output-format csv -- python3 scratch/wgrad_lab.py > gpurun_out/wgp.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/wgp/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-90s %5d calls avg %9.1f us  min %9.1f max %9.1f" % (r["Name"][:90], int(r["Calls"]), float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
