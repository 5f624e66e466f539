import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
from bench_layer import bench
for sh in [(32,256,256,128,2,3), (32,256,256,10,128,5)]:
    for kind in ('fwd', 'bwd', 'wgrad'):
        bench(*sh, 'f16x3', True, reps=10, kind=kind)
