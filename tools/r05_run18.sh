#!/bin/bash
# Round-5 GPU call 18: full GPU suite, then the round's profile evidence (tools/r05_profile.sh)
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/r05r
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > $OUT/tests.log 2>&1
echo "pytest rc $?" >> $OUT/tests.log
tail -6 $OUT/tests.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "import json; o=json.load(open(\"$OUT/bench.json\")); print(json.dumps(o[\"summary\"])[:1500])"
