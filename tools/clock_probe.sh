#!/bin/bash
# effective shader clock per kernel: GRBM_GUI_ACTIVE / kernel wall time of the same (profiled) pass
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/clk; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/a -o a --output-format csv -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $REPO
python - <<'PY'
import csv, glob, collections
dur = {}
for f in glob.glob('gpurun_out/clk/a/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        dur[r['Dispatch_Id']] = (r['Kernel_Name'].split('(')[0].replace('void ', ''), int(r['End_Timestamp']) - int(r['Start_Timestamp']))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/clk/a/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        d = dur.get(r['Dispatch_Id'])
        if d and d[1] > 100000: agg[d[0]][r['Counter_Name']].append((float(r['Counter_Value']), d[1]))  # (idle guard launches: few us)
for n, d in agg.items():
    if not any(k in n for k in ('fwd2', 'bwd2', 'fwd3', 'bwd3', 'dw_body', 'fwd_kernel', 'c16', 'teacher')): continue
    g = d.get('GRBM_GUI_ACTIVE', [])
    if not g: continue
    # GRBM_GUI_ACTIVE is summed over the XCDs (8): cycles per XCD = value / 8
    clk = [v / 8 / (ns * 1e-9) / 1e9 for v, ns in g]
    wc = d.get('SQ_WAVE_CYCLES', []); mb = d.get('SQ_VALU_MFMA_BUSY_CYCLES', [])
    busy = (sum(v for v, _ in mb) / (4 * sum(v for v, _ in wc))) if wc and mb else float('nan')
    print('%-40s wall %.3f ms  clock %.2f GHz (GUI_ACTIVE/8/wall)  mfma_busy %.1f%%' % (n[:40], sum(ns for _, ns in g) / len(g) / 1e6, sum(clk) / len(clk), 100 * busy))
PY
rm -rf $OUT
