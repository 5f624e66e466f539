import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
from bench_layer import bench
for sh in [(32,256,256,128,128,3), (8192,16,16,256,256,3), (8192,16,16,128,256,3), (32,128,128,128,128,3), (32,64,64,128,128,3), (40,256,256,64,128,3)]:
    bench(*sh, 'f16x3', True, reps=10, kind='fwd')
