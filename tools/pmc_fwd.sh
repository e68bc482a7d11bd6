#!/bin/bash
# PMC passes over the render bench (forward kernel): where do the non-MFMA cycles go?
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc; mkdir -p $OUT; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $OUT/a -o a --output-format csv -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train > /dev/null 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $OUT/b -o b --output-format csv -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC -d $OUT/c -o c --output-format csv -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train > /dev/null 2>&1
cd $REPO
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc/*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'r2l_fwd' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(f.split('/')[-2], k, 'n=%d mean=%.4g' % (len(v), sum(v)/len(v)))
PY
find $OUT -name "*.csv" -size +200k -delete
