#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/prof2
rm -rf $OUT/pmc; mkdir -p $OUT
SMALL="bench.py --steps 1 --warmup 1 --res 160 --features synthetic --no-split-precision --no-cpu-baseline --no-genmesh"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $OUT/pmc -- python $SMALL > $OUT/bench.json 2> $OUT/err.log
tail -3 $OUT/err.log
python - <<'PY'
import csv, glob, collections
p = glob.glob('gpurun_out/prof2/pmc/*/*_counter_collection.csv')[0]
rows = [r for r in csv.DictReader(open(p)) if 'k_mlp_gemm' in r['Kernel_Name']]
by = collections.defaultdict(dict)
for r in rows:
    d = by[r['Dispatch_Id']]
    d[r['Counter_Name']] = float(r['Counter_Value']); d['dur'] = int(r['End_Timestamp']) - int(r['Start_Timestamp']); d['grid'] = int(r['Grid_Size'])
agg = collections.defaultdict(list)
for k, d in by.items():
    agg[d['grid'] // 256].append(d)
for nb, v in sorted(agg.items()):
    v = [d for d in v if 'GRBM_GUI_ACTIVE' in d]
    if not v: continue
    d = sorted(v, key=lambda x: x['dur'])[len(v) // 2]
    clk = d['GRBM_GUI_ACTIVE'] / 8 / d['dur']          # GHz (sum over 8 XCDs)
    util = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * d['GRBM_GUI_ACTIVE'] / 8)
    print('blocks', nb, 'n', len(v), 'dur us %.1f' % (d['dur'] / 1e3), 'clock GHz %.3f' % clk, 'mfma_busy/(1024 SIMD x cycles) %.3f' % util,
          {k: '%.3g' % x for k, x in d.items() if k not in ('dur', 'grid')})
PY
