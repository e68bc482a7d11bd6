#!/bin/bash
# Round-5 GPU call 6: head / tail gradients beside the body's (small steps): same-box A/B; full GPU suite; bench.
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/r05f
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
for r in 1 2 3; do
  python tools/small_step_time.py "head + tail beside the body" >> $OUT/small_ab.txt 2>&1
  R2L_NO_DW_OVERLAP=1 python tools/small_step_time.py "one stream (round 4)" >> $OUT/small_ab.txt 2>&1
done
cat $OUT/small_ab.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/tests.log 2>&1
echo "pytest rc $?" >> $OUT/tests.log
tail -12 $OUT/tests.log
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc $?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05f/bench.json'))
print(json.dumps(d['summary'])[:3000])
PY
