import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
from bench_layer import bench
for halo in (True, False):
    for kind in ('fwd', 'bwd', 'wgrad'):
        bench(32,256,256,128,2,3, 'f16x3', halo, reps=10, kind=kind)
    for kind in ('fwd', 'wgrad'):
        bench(32,256,256,10,128,5, 'f16x3', halo, reps=10, kind=kind)
