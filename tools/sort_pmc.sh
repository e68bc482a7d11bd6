#!/bin/bash
# PMC passes of tools/r2o_time.py for the sample_pdf + sort kernel (separate rocprofv3 --pmc runs, no trace domain beside them):
# VALU / LDS instruction counts, LDS bank conflicts, wave cycles -> gpurun_out/<name>/pmc_sort.txt        tools/sort_pmc.sh <name>
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-sort_pmc}; mkdir -p "$OUT"
pass() { k=$1; shift; (cd /tmp && rocprofv3 --pmc "$@" -d "$OUT/p$k" -o p --output-format csv -- python $ROOT/tools/r2o_time.py > "$OUT/p$k.log" 2>&1); }
pass 1 SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass 2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU
pass 3 SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sample_pdf" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_sort.txt", "w") as fh:
    for k, d in acc.items():
        fh.write(k + "\n")
        for c, v in sorted(d.items()):
            fh.write("  %-28s mean per launch %14.1f  (%d launches)\n" % (c, sum(v) / len(v), len(v)))
print(open(out + "/pmc_sort.txt").read())
PY
rm -rf "$OUT"/p[0-9]
