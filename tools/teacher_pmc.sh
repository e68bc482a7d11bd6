#!/bin/bash
# PMC counters of the teacher kernels: tools/teacher_pmc.sh
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d /tmp/tpm -o tpm --output-format csv -- python $REPO/bench.py --no-cpu-baseline --no-train --steps 2 --warmup 1 > /tmp/tpm.log 2>&1 || tail -20 /tmp/tpm.log
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/tpm/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0]
    if "r2l_" in n:
        agg[n + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-60s waves %7d  mfma_busy/wave %5.1f%%  wait_any %5.1f%%  wait_inst %5.1f%%  gui/8 %.0f" %
          (k[:60], m.get("SQ_WAVES", 0), 100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * wc),
           100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("GRBM_GUI_ACTIVE", 0) / 8))
PY
