#!/bin/bash
# Round-5 GPU call 15: test-set loop with images, 200 frames (smooth / incompressible ground truth), twice
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/r05o
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
nproc > $OUT/e2e_render.txt
for k in 1 2; do
timeout 300 python tools/e2e_render.py 40 >> $OUT/e2e_render.txt 2>&1
timeout 300 python tools/e2e_render.py 200 >> $OUT/e2e_render.txt 2>&1
timeout 300 python tools/e2e_render.py 200 noise >> $OUT/e2e_render.txt 2>&1
done
grep -v amdgpu.ids $OUT/e2e_render.txt | cut -c1-200
