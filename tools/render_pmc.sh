#!/bin/bash
# PMC counters of the render kernel: tools/render_pmc.sh
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
rm -rf /tmp/rpm
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU -d /tmp/rpm -o rpm --output-format csv -- python $REPO/bench.py --no-cpu-baseline --no-train --no-teacher --steps 3 --warmup 1 > /tmp/rpm.log 2>&1 || tail -20 /tmp/rpm.log
rm -rf /tmp/rpm2
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES -d /tmp/rpm2 -o rpm2 --output-format csv -- python $REPO/bench.py --no-cpu-baseline --no-train --no-teacher --steps 3 --warmup 1 > /tmp/rpm2.log 2>&1 || tail -20 /tmp/rpm2.log
python - <<'PY'
import csv, glob, collections
for d in ("/tmp/rpm", "/tmp/rpm2"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no counters in", d); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"].split("(")[0]
        if "fwd" in n:
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in agg.items():
        m = {c: sum(v) / len(v) for c, v in dd.items()}
        print(k[:40], {c: "%.4g" % v for c, v in m.items()})
        wc = m.get("SQ_WAVE_CYCLES", 0) or 1
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            print("   mfma_busy/wave %.1f%%  wait_any %.1f%%  wait_inst %.1f%%" % (100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * wc), 100 * m["SQ_WAIT_ANY"] / wc, 100 * m["SQ_WAIT_INST_ANY"] / wc))
        if "SQ_WAIT_INST_LDS" in m:
            print("   wait_inst_lds %.1f%%  active_inst_lds %.1f%%  bank_conflict/wavecycles %.3f" % (100 * m["SQ_WAIT_INST_LDS"] / wc, 100 * m.get("SQ_ACTIVE_INST_LDS", 0) / wc, m.get("SQ_LDS_BANK_CONFLICT", 0) / wc))
PY
