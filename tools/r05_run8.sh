#!/bin/bash
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/r05h
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_teacher_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log | cut -c1-200
timeout 300 python - > $OUT/r2o.txt 2>&1 <<'PY'
import sys, json, torch
sys.path.insert(0, '.')
import bench
print(json.dumps(bench.raw2outputs_leg(torch.device('cuda', 0), 20, 3), indent=1))
PY
cat $OUT/r2o.txt | grep -E "us_per_launch|achieved|frac\"|S64|S192|at_262"
