#!/bin/bash
# Run on the GPU box (via gpurun): bench + rocprofv3 kernel-trace stats + PMC passes. Summaries -> gpurun_out/prof/
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc1 -o pmc1 --output-format csv -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-teacher > $OUT/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc2 -o pmc2 --output-format csv -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-teacher > $OUT/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc3 -o pmc3 --output-format csv -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-teacher > $OUT/pmc3.log 2>&1
cd $REPO
python - <<'PY'
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob('gpurun_out/prof/pmc*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0]
        if name.startswith(('r2l_', 'void r2l_')):
            agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        for c, v in d.items():
            out.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(out, open('gpurun_out/prof/pmc_summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/kt/kt_kernel_stats.csv | cut -c1-160 | head -20
