#!/bin/bash
# per-layer GEMM durations of k_mlp_gemm_buf under the PIFU_GEMM_DBG ablations (EXPERIMENTS build)
export TMPDIR=/tmp
for d in "$@"; do
  OUT=gpurun_out/dbg_$d; rm -rf $OUT; mkdir -p $OUT
  PIFU_GEMM_DBG=$d rocprofv3 --kernel-trace --output-format csv -d $OUT -- python tools/lab/bench_lab.py --steps 1 --warmup 1 --features synthetic --batch-rows 262144 --no-cpu-baseline --no-genmesh --no-split-precision --no-lean-pass > $OUT/line.json 2> $OUT/err.log
  python - "$OUT" "$d" <<'P'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_mlp_gemm' in r['Kernel_Name']:
        d[int(r['Grid_Size_X']) // 256].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('dbg', sys.argv[2], {k: round(sum(v) / len(v), 1) for k, v in sorted(d.items()) if len(v) > 100})
P
done
