#!/bin/bash
# Round-5 GPU call 16: the training-equivalence table completed: the bf16x3 trio on seeds 0 - 3, and seeds 4 - 7 for the default trio and
# the exact-fp32 family (12 000 steps of 16 384 rays each, as profiles/r05_train_equivalence_seeds.txt)
export TMPDIR=/tmp
OUT=gpurun_out/r05p
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
R2L_EQ_SEEDS=0,1,2,3 R2L_EQ_FAMILIES=2 timeout 600 python tools/train_equivalence.py 12000 16384 > $OUT/train_eq_bf16x3.txt 2>&1
tail -4 $OUT/train_eq_bf16x3.txt | cut -c1-250
R2L_EQ_SEEDS=4,5,6,7 R2L_EQ_FAMILIES=0,3 timeout 900 python tools/train_equivalence.py 12000 16384 > $OUT/train_eq_seeds4to7.txt 2>&1
tail -6 $OUT/train_eq_seeds4to7.txt | cut -c1-250
