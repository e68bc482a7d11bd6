#!/bin/bash
# Round-5 GPU call 7: tile-major cooperative chains (two ray tiles per workgroup): parity tests, then same-box A/B against the library before the change.
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/r05g
REPO=$(pwd)
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1200 python -m pytest tests/test_forward_gpu.py tests/test_train_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "coopf or two_tiles or segmented" -x > $OUT/tests.log 2>&1
echo "pytest rc $?" >> $OUT/tests.log
tail -25 $OUT/tests.log | cut -c1-250
for r in 1 2 3; do
  timeout 120 python tools/small_step_time.py "tile-major (NT = 2) + E/O k order" >> $OUT/small_ab.txt 2>&1
  R2L_LIB_PATH=$REPO/tools/_bin/round5_before_tm/libr2l_hip.so timeout 120 python tools/small_step_time.py "before" >> $OUT/small_ab.txt 2>&1
done
cat $OUT/small_ab.txt
