#!/bin/bash
# Round-5 GPU call 1: full GPU suite, the bench line, the multi-seed training-equivalence table (run from the repo root via gpurun).
export TMPDIR=/tmp
OUT=gpurun_out/r05a
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/tests.log 2>&1
echo "pytest rc $?" >> $OUT/tests.log
tail -25 $OUT/tests.log
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc $?"; tail -c 2500 $OUT/bench.json
R2L_EQ_SEEDS=0,1,2,3 R2L_EQ_FAMILIES=0,1,3 timeout 900 python tools/train_equivalence.py 12000 16384 > $OUT/train_eq_seeds.txt 2>&1
tail -12 $OUT/train_eq_seeds.txt
