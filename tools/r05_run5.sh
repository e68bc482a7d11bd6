#!/bin/bash
# Round-5 GPU call 5: what do the ride-along stash stores of the training chains cost?  Cache-policy variants and a
# no-store timing build of r2l_fwd2_kernel<SAVE> / r2l_bwd2_kernel, same box, per-kernel averages from rocprofv3 --kernel-trace.
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/r05e
REPO=$(pwd)
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for r in 1 2; do
for v in default hst_aux0 hst_aux3 hst_aux18 hst_nostore; do
  if [ $v = default ]; then unset R2L_LIB_PATH; else export R2L_LIB_PATH=$REPO/tools/_bin/$v/libr2l_hip.so; fi
  rocprofv3 --kernel-trace --stats -d $OUT/kt_${v}_$r -o kt --output-format csv -- python $REPO/tools/train_step_time.py "$v" 40 > $OUT/kt_${v}_$r.log 2>&1
  grep "ms per step" $OUT/kt_${v}_$r.log >> $OUT/stash_store_ab.txt
  f=$(find $OUT/kt_${v}_$r -name "*kernel_stats.csv" | head -1)
  python - "$f" >> $OUT/stash_store_ab.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0]
    if 'r2l_fwd2_kernel<false, true' in n or 'r2l_bwd2_kernel' in n or 'r2l_dw16_kernel' in n:
        print('      %-52s calls %4s avg %9.1f us' % (n[:52], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  rm -rf $OUT/kt_${v}_$r
done; done
cat $OUT/stash_store_ab.txt
