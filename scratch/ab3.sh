cp master_thesis_amd/libmt355.so /tmp/base.so
for v in base abl_NOB abl_NOSYNC abl_NOBDABL_NOSYNC; do if [ $v = base ]; then cp /tmp/base.so master_thesis_amd/libmt355.so; else cp master_thesis_amd/libmt355_$v.so master_thesis_amd/libmt355.so; fi; echo "== $v"; python scratch/one_layer.py fwd bf16x6; python scratch/one_layer.py fwd bf16x3; done
cp /tmp/base.so master_thesis_amd/libmt355.so
