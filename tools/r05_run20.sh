#!/bin/bash
# Round-5 GPU call 20: as call 19, around the workgroup counts that divide the 86 x 96 = 8256 chunk units of a 98 304-ray step evenly
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/r05t
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
run() { label=$1; shift; env "$@" python tools/train_step_time.py "$label" 80 2>&1 | grep "ms per step" >> $OUT/ab.txt; }
for k in 1 2; do
run "default (256 dW workgroups, one stream)" A=1
run "dW on 192, one stream" R2L_DW_WGS=192
run "dW on 172, one stream" R2L_DW_WGS=172
run "overlap, dW on 192" R2L_DW_OVERLAP_MAX_RAYS=1000000 R2L_DW_WGS=192
run "overlap, dW on 172" R2L_DW_OVERLAP_MAX_RAYS=1000000 R2L_DW_WGS=172
run "overlap, dW on 129" R2L_DW_OVERLAP_MAX_RAYS=1000000 R2L_DW_WGS=129
run "overlap, dW on 196" R2L_DW_OVERLAP_MAX_RAYS=1000000 R2L_DW_WGS=196
run "overlap, dW on 188" R2L_DW_OVERLAP_MAX_RAYS=1000000 R2L_DW_WGS=188
done
cat $OUT/ab.txt
