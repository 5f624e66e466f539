#!/bin/bash
# usage: step_stats.sh dfpn|chn N
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/ss_$1 -o s --output-format csv -- python3 scratch/step_stats.py $1 $2 > gpurun_out/ss_$1.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/ss_$1/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
n=$2
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("$1: kernel time per step %.1f ms, launches per step %.0f" % (tot/1e6/n, sum(int(r["Calls"]) for r in rows)/n))
for r in rows[:40]:
    print("%-78s %7.1f /step %8.3f ms/step %6.2f%%" % (r["Name"][:78], int(r["Calls"])/n, float(r["TotalDurationNs"])/1e6/n, float(r["Percentage"])))
PY
