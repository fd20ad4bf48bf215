#!/bin/bash
# per-kernel time of the headline lattice evaluation (depth walk on): rocprofv3 kernel stats, top rows
export TMPDIR=/tmp
OUT=gpurun_out/kprof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python bench.py --steps 2 --warmup 1 --features ${FEATURES:-synthetic} --no-cpu-baseline --no-genmesh --no-split-precision --no-lean-pass > $OUT/line.json 2> $OUT/err.log
python - <<'P'
import csv, glob
f = glob.glob('gpurun_out/kprof/*/*_kernel_stats.csv')[0]
for i, r in enumerate(csv.DictReader(open(f))):
    if i >= 14: break
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]
    print(f"{n:42s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']:>6s}%")
P
cut -c1-200 $OUT/line.json
