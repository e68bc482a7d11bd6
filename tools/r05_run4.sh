#!/bin/bash
# Round-5 GPU call 4: fused Adam + re-pack A/B again (wider grids) with kernel stats; failing tests re-run.
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/r05d
REPO=$(pwd)
rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_driver_gpu.py tests/test_forward_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -k "adam_packed or dx_chain or three_adam or trained_weights_parity or emb_path" > $OUT/tests.log 2>&1
tail -6 $OUT/tests.log
for r in 1 2 3; do
  R2L_ADAM_PACK=1 python tools/small_step_time.py "fused adam + re-pack" >> $OUT/small_ab.txt 2>&1
  python tools/small_step_time.py "separate adam + packs" >> $OUT/small_ab.txt 2>&1
done
cat $OUT/small_ab.txt
cd /tmp
for v in fused separate; do
  if [ $v = fused ]; then export R2L_ADAM_PACK=1; else unset R2L_ADAM_PACK; fi
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$v -o kt --output-format csv -- python $REPO/tools/small_step_time.py $v 200 > $OUT/kt_$v.log 2>&1
  f=$(find $OUT/kt_$v -name "*kernel_stats.csv" | head -1)
  python - "$f" > $OUT/kernels_$v.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0]
    if 'r2l' in n or 'elementwise' in n: print('%-64s calls %6s avg %9.1f us  total %9.1f ms' % (n[:64], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
  find $OUT/kt_$v -name "*kernel_trace.csv" -delete
  echo "== $v"; head -30 $OUT/kernels_$v.txt
done
