#!/bin/bash
# Run on the GPU box (via gpurun): bench + rocprofv3 kernel-trace stats + PMC passes. Summaries -> gpurun_out/prof/
set -x
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/kt.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $OUT/pmc1 -o pmc1 --output-format csv -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc2 -o pmc2 --output-format csv -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc3 -o pmc3 --output-format csv -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc3.log 2>&1
cd $REPO
find $OUT -name "*.csv" | head -30
for f in $(find $OUT/kt -name "*kernel_stats.csv"); do head -20 $f; done
# keep only small summaries
find $OUT -name "*kernel_trace.csv" -size +2M -delete
ls -la $OUT/*
