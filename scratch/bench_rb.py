import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
from bench_layer import bench
shapes = [(32,256,256,128,128,3), (8192,16,16,256,256,3), (32,256,256,128,128,5), (32,64,64,64,160,3), (32,128,128,128,128,3)]
for sh in shapes:
    for kind in ('fwd', 'bwd'):
        bench(*sh, 'f16x3', True, reps=10, kind=kind)
