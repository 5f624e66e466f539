import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
from bench_layer import bench
bench(32, 256, 256, 128, 128, 3, 'f16x3', True, reps=20, kind='fwd')
