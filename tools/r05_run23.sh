#!/bin/bash
# Round-5 GPU call 23: effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time) and MFMA busy of the render kernels on the three
# data sets of tools/operand_entropy_render.py (default / fp16-exact / zero-body weights: 7 launches each, per family, in this order)
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/r05w; rm -rf $OUT; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $OUT/a -o a --output-format csv -- python $REPO/tools/operand_entropy_render.py 4 > $OUT/run.log 2>&1
cd $REPO
python - > $OUT/clock.txt <<'PY'
import csv, glob, collections
dur = {}
for f in glob.glob('gpurun_out/r05w/a/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        dur[int(r['Dispatch_Id'])] = (r['Kernel_Name'].split('(')[0].replace('void ', ''), int(r['End_Timestamp']) - int(r['Start_Timestamp']))
cnt = collections.defaultdict(dict)
for f in glob.glob('gpurun_out/r05w/a/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        cnt[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
groups = collections.defaultdict(list)
for did in sorted(dur):
    name, ns = dur[did]
    if ns > 5_000_000 and ('fwd2_kernel<true' in name or 'fwd3_kernel<true' in name or 'fwd_kernel<1' in name):
        groups[name].append((did, ns))
for name, rows in groups.items():
    per = len(rows) // 3
    for gi, label in enumerate(('default', 'fp16-exact', 'zero body')):
        sel = rows[gi * per:(gi + 1) * per][3:]  # (skip the three warm-up launches of the group)
        ms = sum(ns for _, ns in sel) / len(sel) / 1e6
        clk = sum(cnt[d]['GRBM_GUI_ACTIVE'] / 8 / (ns * 1e-9) / 1e9 for d, ns in sel) / len(sel)
        busy = sum(cnt[d]['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (cnt[d]['GRBM_GUI_ACTIVE'] / 8) for d, ns in sel) / len(sel)
        print('%-34s %-11s %8.2f ms (under the profiler)  clock %.2f GHz  MFMA busy %.1f %% of the active cycles' % (name[:34], label, ms, clk, 100 * busy))
PY
cat $OUT/clock.txt; tail -3 $OUT/run.log | cut -c1-200
find $OUT -name "*.csv" -delete
