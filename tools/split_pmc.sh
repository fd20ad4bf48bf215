#!/bin/bash
# SQ counters of the MLP GEMM kernels (fp32 and split precision) on the same launch mix: separate --pmc passes, kernel trace only
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/splitpmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for mode in 0 2; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a$mode -- python3 $R/tools/split_pmc.py $mode > $OUT/a$mode.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d $OUT/b$mode -- python3 $R/tools/split_pmc.py $mode > $OUT/b$mode.log 2>&1
done
cd $R
python3 - <<'P'
import csv, glob, collections
for tag in ('a0', 'b0', 'a2', 'b2'):
    fs = glob.glob(f'gpurun_out/splitpmc/{tag}/*/*counter_collection.csv')
    if not fs:
        print(tag, 'no counter file'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        if 'gemm' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k, d in acc.items():
        print(tag, k[:34], {n: f'{v:.4g}' for n, v in sorted(d.items())})
P
rm -rf $OUT/*/*/*kernel_trace.csv
