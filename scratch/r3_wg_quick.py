import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
from bench_layer import bench
bench(32,256,256,128,128,3, 'f16x3', True, reps=10, kind='fwd')      # (box-speed calibrator)
for sh in [(32,256,256,128,128,3), (8192,16,16,256,256,3), (32,128,128,128,128,3), (32,256,256,128,128,5)]:
    bench(*sh, 'f16x3', True, reps=10, kind='wgrad')
