#!/bin/bash
# Round-5 GPU call 19: 98 304-ray default step: head / tail weight gradients beside the body's kernel when that leaves CUs free
# (R2L_DW_OVERLAP_MAX_RAYS lifts the small-step condition of the overlap; R2L_DW_WGS = persistent workgroups of r2l_dw16_kernel)
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/r05s
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
run() { label=$1; shift; env "$@" python tools/train_step_time.py "$label" 80 2>&1 | grep "ms per step" >> $OUT/ab.txt; }
for k in 1 2; do
run "default (256 dW workgroups, one stream)" A=1
run "dW on 224 workgroups, one stream" R2L_DW_WGS=224
run "dW on 192 workgroups, one stream" R2L_DW_WGS=192
run "overlap, dW on 256" R2L_DW_OVERLAP_MAX_RAYS=1000000
run "overlap, dW on 240" R2L_DW_OVERLAP_MAX_RAYS=1000000 R2L_DW_WGS=240
run "overlap, dW on 224" R2L_DW_OVERLAP_MAX_RAYS=1000000 R2L_DW_WGS=224
run "overlap, dW on 208" R2L_DW_OVERLAP_MAX_RAYS=1000000 R2L_DW_WGS=208
run "overlap, dW on 192" R2L_DW_OVERLAP_MAX_RAYS=1000000 R2L_DW_WGS=192
run "overlap, dW on 160" R2L_DW_OVERLAP_MAX_RAYS=1000000 R2L_DW_WGS=160
done
cat $OUT/ab.txt
