#!/bin/bash
# Round-5 GPU call 11: register-resident sample_pdf_sort kernel: teacher-path tests, then the HBM legs of the bench (hipGraph replay, HBM-cold)
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/r05k
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_teacher_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider > $OUT/tests.log 2>&1
echo "pytest rc $?" >> $OUT/tests.log
tail -6 $OUT/tests.log | cut -c1-300
timeout 300 python - > $OUT/r2o.log 2>&1 <<'PY'
import json, torch, bench
for rep in range(3):
    r = bench.raw2outputs_leg(torch.device("cuda:0"), 20, 3)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk in ("us_per_launch", "achieved", "frac", "at_262144_rays")} for k, v in r.items() if isinstance(v, dict)}))
PY
cat $OUT/r2o.log | cut -c1-900
