for v in khb3_occ1 khb1_occ2; do cp master_thesis_amd/libmt355_$v.so master_thesis_amd/libmt355.so; echo "== $v"; python scratch/bench_layer.py wgrad 2>&1 | grep -E "k3"; done
