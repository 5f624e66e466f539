import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import bench
r = bench.corr_roofline()
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
