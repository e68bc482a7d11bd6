#!/bin/bash
# per-kernel averages of library variants on one box: tools/ab_kt.sh libA.so libB.so -> gpurun_out/abkt/<lib>.txt
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/abkt
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for L in "$@"; do
  R2L_LIB_PATH=$REPO/r2l_amd/lib/$L rocprofv3 --kernel-trace --stats -d $OUT/$L -o kt --output-format csv -- python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-teacher > $OUT/$L.log 2>&1
  f=$(find $OUT/$L -name "*kernel_stats.csv" | head -1)
  python - "$f" > $OUT/$L.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0]
    if 'r2l' in n: print('%-60s calls %5s avg %10.1f us' % (n[:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  find $OUT/$L -name "*kernel_trace.csv" -delete
  echo "== $L"; cat $OUT/$L.txt
done
