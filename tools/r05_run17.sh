#!/bin/bash
# Round-5 GPU call 17: does the teacher frame care about --chunk?  (32 768 = the reference's default; 65 536; 163 840 = one launch per frame)
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/r05q
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
for k in 1 2 3; do for c in 32768 65536 163840; do timeout 200 python tools/teacher_time.py $c 2>&1 | grep "teacher frame" >> $OUT/chunk.txt; done; done
cat $OUT/chunk.txt
