#!/bin/bash
# per-kernel breakdown of one small-batch training step (default 4096 rays): rocprofv3 kernel-trace stats
export TMPDIR=/tmp
REPO=$(pwd)
N=${1:-4096}
mkdir -p $REPO/gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/sp -o sp --output-format csv -- python $REPO/tools/small_prof.py $N > /tmp/sp.log 2>&1 || tail -20 /tmp/sp.log
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/sp/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-64s calls %4s avg %10.1f us  %5s %%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
