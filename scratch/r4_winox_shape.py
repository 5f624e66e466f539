# the dense-block layer the CHN trunk launches most: 160 -> 32 on 16 images of 64 x 64 (one launch chain's half batch), 100 launches
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
from bench_layer import bench
bench(16, 64, 64, 160, 32, 3, 'f16x3', True, reps=100, kind='fwd')
