import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
from bench_layer import bench
for sh in [(32,256,256,128,128,5), (32,128,128,128,128,5), (32,64,64,128,128,5), (32,32,32,128,128,5), (32,16,16,256,256,5)]:
    bench(*sh, 'f16x3', True, reps=10, kind='fwd')
    bench(*sh, 'f16x3', True, reps=10, kind='bwd')
