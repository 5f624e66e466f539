#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): collects the rocprofv3 evidence bench.py's numbers are
# judged against and writes compact summaries under gpurun_out/profiles_<tag>/ :
#   <tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check $only` (DFPN cfg2 only, so per-kernel averages are those of the roofline line)
#   <tag>_bench_under_rocprof.json the bench line printed in that run
#   <tag>_pmc_hbm_traffic.txt      FETCH_SIZE / WRITE_SIZE (separate passes) per kernel, DFPN cfg2 step (eager) + correlation micro-bench
#   <tag>_pmc_mfma_util.txt        matrix-pipe utilisation and sustained clock per kernel (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE)
# usage: tools/collect_profiles.sh <tag> [dfpn|chn]     (chn: the CHN step alone, files named <tag>_..._chn_...)
set -u
export TMPDIR=/tmp
tag=${1:-r04}
mode=${2:-dfpn}
if [ "$mode" = chn ]; then only=--no-dfpn; sfx=_chn; else only=--no-chn; sfx=; fi
out=gpurun_out/profiles_$tag$sfx
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats -o bench --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check $only > $out/${tag}_bench${sfx}_under_rocprof.json 2> $out/stats.err || echo "stats run failed"
cp $(find $out/stats -name '*kernel_stats.csv' | head -1) $out/${tag}_bench${sfx}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/pmc_$c -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check $only --no-graph > $out/pmc_$c.json 2> $out/pmc_$c.err || echo "pmc $c failed"
done
# matrix-pipe utilisation: MFMA busy cycles (summed over the 1024 SIMDs) against the GPU-active cycles
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_MFMA -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check $only --no-graph > $out/pmc_MFMA.json 2> $out/pmc_MFMA.err || echo "pmc MFMA failed"
python3 - "$out" "$tag" "$sfx" "$only" <<'PY'
import csv, glob, sys, collections, os
sys.path.insert(0, os.getcwd())
out, tag, sfx, only = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
what = 'CHN cfg3 step' if sfx else 'DFPN cfg2 step'
def load(counter):
    f = glob.glob(f'{out}/pmc_{counter}/**/*counter_collection.csv', recursive=True)
    acc = collections.defaultdict(list)
    if not f:
        return acc
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] == counter:
            acc[r['Kernel_Name']].append((float(r['Counter_Value']), (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e6))
    return acc
fe, wr = load('FETCH_SIZE'), load('WRITE_SIZE')
rows = []
for k in fe:
    n = len(fe[k])
    f = sum(v for v, _ in fe[k]) / n
    w = sum(v for v, _ in wr.get(k, [(0, 0)])) / max(1, len(wr.get(k, [1])))
    ms = sum(t for _, t in fe[k]) / n
    mb = (2 * f + w) * 1024 / 1e6
    rows.append((mb * n, k, n, f, w, mb, ms))
rows.sort(reverse=True)
with open(f'{out}/{tag}_pmc{sfx}_hbm_traffic.txt', 'w') as o:
    o.write('# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check ' + only + ' --no-graph\n')
    o.write(f'# ({what}s, eager launches, + the 2048-pair correlation micro-benchmark).  Units: KB per dispatch; gfx950 correction\n')
    o.write('# (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads 1/2 of a wide coalesced stream -> bytes = (2*FETCH + WRITE) * 1024\n')
    import bench
    o.write('# csrc_sha16: ' + bench.csrc_sha16() + '\n')
    o.write('kernel,launches,avg_FETCH_SIZE_KB,avg_WRITE_SIZE_KB,corrected_MB_per_launch,avg_ms_under_pmc,corrected_GB_per_s\n')
    for _, k, n, f, w, mb, ms in rows[:40]:
        name = k.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0]
        o.write(f'{name},{n},{f:.1f},{w:.1f},{mb:.1f},{ms:.3f},{mb / ms if ms else 0:.0f}\n')
print(open(f'{out}/{tag}_pmc{sfx}_hbm_traffic.txt').read()[:3000])
# ---- MFMA utilisation
f = glob.glob(f'{out}/pmc_MFMA/**/*counter_collection.csv', recursive=True)
if f:
    busy, act, dur = collections.defaultdict(list), collections.defaultdict(list), collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name']
        if r['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES':
            busy[k].append(float(r['Counter_Value']))
            dur[k].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e6)
        elif r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            act[k].append(float(r['Counter_Value']))
    rows = []
    for k in busy:
        b, a, ms = sum(busy[k]), sum(act.get(k, [0])), sum(dur[k])
        if b <= 0 or a <= 0:
            continue
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs = 1024 matrix pipes
        util = b / (a / 8 * 1024)
        clk = a / 8 / (ms * 1e-3) / 1e9
        rows.append((b, k, len(busy[k]), ms, util, clk))
    rows.sort(reverse=True)
    with open(f'{out}/{tag}_pmc{sfx}_mfma_util.txt', 'w') as o:
        o.write('# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace, bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check ' + only + ' --no-graph\n')
        o.write('# mfma_util = MFMA busy cycles / (GPU-active cycles per XCD x 1024 SIMDs); clock = active cycles per XCD / duration.\n')
        o.write('# The dense 16-bit peak (2.5 PFLOP/s) assumes util 1.0 at 2.4 GHz; on random fp16 data a pure-MFMA loop holds ~1.62 GHz (scratch/lab/mfma_peak.hip).\n')
        o.write('kernel,launches,total_ms_under_pmc,mfma_util,clock_GHz\n')
        for b, k, n, ms, util, clk in rows[:30]:
            name = k.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0]
            o.write(f'{name},{n},{ms:.3f},{util:.3f},{clk:.2f}\n')
    print(open(f'{out}/{tag}_pmc{sfx}_mfma_util.txt').read())
PY
rm -rf $out/stats $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_MFMA
ls -la $out
