#!/bin/bash
# Round-5 GPU call 21: the render kernels on operands of different entropy (tools/operand_entropy_render.py), twice
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/r05u
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
for k in 1 2; do timeout 400 python tools/operand_entropy_render.py 12 2>&1 | grep "ms per" >> $OUT/entropy.txt; echo >> $OUT/entropy.txt; done
cat $OUT/entropy.txt
