#!/bin/bash
# per-layer GEMM durations (grouped by grid size) of two GEMM variants, 262 144-row batches, synthetic features
export TMPDIR=/tmp
for v in "$@"; do
  OUT=gpurun_out/lab_$v; rm -rf $OUT; mkdir -p $OUT
  PIFU_GEMM_VARIANT=$v rocprofv3 --kernel-trace --output-format csv -d $OUT -- python tools/lab/bench_lab.py --steps 1 --warmup 1 --features synthetic --batch-rows 262144 --no-cpu-baseline --no-genmesh --no-split-precision --no-lean-pass > $OUT/line.json 2> $OUT/err.log
  python - "$OUT" "$v" <<'P'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_mlp_gemm' in r['Kernel_Name']:
        name = r['Kernel_Name'].split('k_mlp_gemm')[1].split('(')[0][:8]
        d[(name, int(r['Grid_Size_X']) // 256)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('variant', sys.argv[2])
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print('  k_mlp_gemm%-8s blocks %6d launches %4d avg_us %8.1f total_ms %8.2f' % (k[0], k[1], len(v), sum(v) / len(v), sum(v) / 1e3))
P
  cut -c1-120 $OUT/line.json
done
