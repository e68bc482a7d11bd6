#!/bin/bash
# Round-5 GPU call 22: training step and teacher frame on operands of different entropy (tools/operand_entropy_train.py)
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/r05v
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 600 python tools/operand_entropy_train.py 40 > $OUT/entropy_train.log 2>&1
grep "ms per\|Error\|error" $OUT/entropy_train.log | cut -c1-220; tail -3 $OUT/entropy_train.log | cut -c1-300
