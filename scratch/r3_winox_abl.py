import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
from bench_layer import bench
for sh in [(32,64,64,64,32,3), (32,64,64,160,32,3), (16,64,64,160,32,3), (32,64,64,192,64,3), (32,128,128,160,32,3)]:
    bench(*sh, 'f16x3', True, reps=20, kind='fwd')
