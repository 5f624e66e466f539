cp master_thesis_amd/libmt355.so /tmp/base.so
for v in base wg5k5; do if [ $v = base ]; then cp /tmp/base.so master_thesis_amd/libmt355.so; else cp master_thesis_amd/libmt355_$v.so master_thesis_amd/libmt355.so; fi; echo "== $v"; python scratch/bench_layer.py wgrad 2>&1 | grep -E "k5.*x6|k5"; done
cp /tmp/base.so master_thesis_amd/libmt355.so
python scratch/bench_layer.py fwd | grep x6
