#!/bin/bash
# Round-6 evidence, run through gpurun from the repo root: bench line, rocprofv3 kernel-trace + PMC passes of the same command,
# power / clock trace, staged-backward cost, gradient-noise diagnostic.  Summaries -> gpurun_out/r06/ (copy into profiles/).
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06
rm -rf $OUT; mkdir -p $OUT
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python $REPO/bench.py $ARGS > $OUT/kt.log 2>&1
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-teacher"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc1 -o pmc1 --output-format csv -- python $REPO/bench.py $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc2 -o pmc2 --output-format csv -- python $REPO/bench.py $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc3 -o pmc3 --output-format csv -- python $REPO/bench.py $ARGS > $OUT/pmc3.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA -d $OUT/pmc4 -o pmc4 --output-format csv -- python $REPO/bench.py $ARGS > $OUT/pmc4.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA -d $OUT/pmc5 -o pmc5 --output-format csv -- python $REPO/tools/teacher_time.py > $OUT/pmc5.log 2>&1
# the alpha-composite kernel alone (eager launches on HBM-resident inputs): traffic counters, then busy / instruction counters
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc6 -o pmc6 --output-format csv -- python $REPO/tools/r2o_time.py > $OUT/pmc6.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc7 -o pmc7 --output-format csv -- python $REPO/tools/r2o_time.py > $OUT/pmc7.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES -d $OUT/pmc8 -o pmc8 --output-format csv -- python $REPO/tools/r2o_time.py > $OUT/pmc8.log 2>&1
cd $REPO
python - "$OUT" <<'PY'
import csv, glob, collections, json, sys
out_dir = sys.argv[1]
out = {}
for f in sorted(glob.glob(out_dir + '/pmc*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0]
        if name.startswith(('r2l_', 'void r2l_')):
            agg[name + ' grid=' + r.get('Grid_Size', '?')][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        for c, v in d.items():
            out.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(out, open(out_dir + '/pmc_summary.json', 'w'), indent=1)
PY
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
cat $OUT/bench.json | cut -c1-1500
head -30 $OUT/kt/kt_kernel_stats.csv | cut -c1-150

