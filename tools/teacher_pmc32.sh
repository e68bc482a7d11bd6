#!/bin/bash
# PMC pass of the exact-fp32 teacher point network (r2l_teacher_mlp_kernel): MFMA busy, VALU / MFMA instruction counts.   tools/teacher_pmc32.sh
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/teacher_pmc32; mkdir -p "$OUT"
(cd /tmp && R2L_NO_FWD3=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES -d "$OUT/p" -o tp --output-format csv -- python $ROOT/tools/teacher_time.py > "$OUT/run.log" 2>&1)
tail -1 "$OUT/run.log"
python - "$OUT" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "teacher" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:40] + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    busy = 100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (m["GRBM_GUI_ACTIVE"] / 8) if m.get("GRBM_GUI_ACTIVE") else float("nan")
    print(k, {c: round(x) for c, x in m.items()}, "MFMA busy %.1f %%" % busy, "VALU per MFMA %.2f" % (m["SQ_INSTS_VALU"] / max(m["SQ_INSTS_MFMA"], 1)))
PY
rm -rf "$OUT/p"
