import sys, json
sys.path.insert(0, '.')
import torch, bench
for size, pairs in ((256, 1024), (512, 256), (64, 16384)):
    r = bench.warp_roofline(pairs, size)
    print(size, pairs, '%.3f ms' % r['ms'], '%.0f GB/s' % r['achieved'], 'frac %.3f' % r['frac'], flush=True)
