#!/bin/bash
export TMPDIR=/tmp
REPO=$(pwd); OUT=$REPO/gpurun_out/pmct; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $OUT/a -o a --output-format csv -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-teacher > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/c -o c --output-format csv -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-teacher > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $OUT/d -o d --output-format csv -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-teacher > /dev/null 2>&1
cd $REPO
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/pmct/*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if n.startswith('r2l_') and ('fwd' in n or 'bwd' in n or 'dw_body' in n) and 'pack' not in n:
            agg[n][r['Counter_Name']].append(float(r['Counter_Value']))
    for n, d in agg.items():
        for c, v in d.items():
            tot[n][c] = sum(v) / len(v)
for n, d in tot.items():
    wc = d.get('SQ_WAVE_CYCLES', 1)
    print(n)
    print('   wave_cycles(quad) %.4g  wait_any %.1f%%  wait_inst %.1f%%  active %.1f%%  mfma_busy/(4*wc) %.1f%%' % (wc, 100*d.get('SQ_WAIT_ANY',0)/wc, 100*d.get('SQ_WAIT_INST_ANY',0)/wc, 100*d.get('SQ_ACTIVE_INST_ANY',0)/wc, 100*d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(4*wc)))
    print('   insts: valu(non-mfma) %.4g mfma %.4g vmem_rd %.4g vmem_wr %.4g salu %.4g smem %.4g  inst_cycles_vmem %.4g' % (d.get('SQ_INSTS_VALU',0)-d.get('SQ_INSTS_MFMA',0), d.get('SQ_INSTS_MFMA',0), d.get('SQ_INSTS_VMEM_RD',0), d.get('SQ_INSTS_VMEM_WR',0), d.get('SQ_INSTS_SALU',0), d.get('SQ_INSTS_SMEM',0), d.get('SQ_INST_CYCLES_VMEM',0)))
    print('   lds: insts %.4g active %.4g idx_active %.4g bank_conflict %.4g addr_conflict %.4g unaligned %.4g wait_inst_lds %.4g' % (d.get('SQ_INSTS_LDS',0), d.get('SQ_ACTIVE_INST_LDS',0), d.get('SQ_LDS_IDX_ACTIVE',0), d.get('SQ_LDS_BANK_CONFLICT',0), d.get('SQ_LDS_ADDR_CONFLICT',0), d.get('SQ_LDS_UNALIGNED_STALL',0), d.get('SQ_WAIT_INST_LDS',0)))
PY
rm -rf $OUT
