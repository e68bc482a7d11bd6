#!/bin/bash
# Round-5 GPU call 14: the three CLI pipelines end to end with the round's kernels (wall clock; tools/e2e_*.py)
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/r05n
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 500 python tools/e2e_create_data.py > $OUT/e2e_create_data.txt 2>&1; tail -12 $OUT/e2e_create_data.txt | cut -c1-250
timeout 500 python tools/e2e_train.py > $OUT/e2e_train.txt 2>&1; tail -12 $OUT/e2e_train.txt | cut -c1-250
timeout 500 python tools/e2e_render.py > $OUT/e2e_render.txt 2>&1; tail -12 $OUT/e2e_render.txt | cut -c1-250
