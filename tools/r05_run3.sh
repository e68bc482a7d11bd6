#!/bin/bash
# Round-5 GPU call 3: full GPU suite after the prune / fused Adam / emb cfg / expansion fix; small-step A/Bs; chain-trip diagnosis again.
export TMPDIR=/tmp
OUT=gpurun_out/r05c
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 300 python tools/diag_chain_trip.py > $OUT/diag_chain_trip.txt 2>&1
grep -E "step|slot|vs oracle" $OUT/diag_chain_trip.txt | cut -c1-330 | head -40
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/tests.log 2>&1
echo "pytest rc $?" >> $OUT/tests.log
tail -15 $OUT/tests.log
for r in 1 2 3; do
  python tools/small_step_time.py default >> $OUT/small_ab.txt 2>&1
  R2L_NO_ADAM_PACK=1 python tools/small_step_time.py "separate adam + packs" >> $OUT/small_ab.txt 2>&1
  R2L_HEAD_SLICE_RAYS=256 python tools/small_step_time.py "head slices >= 256 rays" >> $OUT/small_ab.txt 2>&1
  R2L_HEAD_SLICE_RAYS=64 python tools/small_step_time.py "head slices >= 64 rays" >> $OUT/small_ab.txt 2>&1
done
cat $OUT/small_ab.txt
