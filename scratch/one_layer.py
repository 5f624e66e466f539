import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'scratch')
from bench_layer import bench
kind = sys.argv[1]; prec = sys.argv[2]
bench(32,256,256,128,128,3, prec, True, reps=3, kind=kind)
