#!/bin/bash
# per-kernel time of gen_mesh 512^3 (rocprofv3 kernel stats, top rows) + the tool's own phase timings
# GENMESH_ARGS='--subject body' profiles the body-like subject instead of the headline one
export TMPDIR=/tmp
OUT=gpurun_out/gmprof
rm -rf $OUT; mkdir -p $OUT
python tools/genmesh_profile.py 512 $GENMESH_ARGS 2> $OUT/plain.err | tail -6
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python tools/genmesh_profile.py 512 $GENMESH_ARGS > $OUT/out.log 2> $OUT/err.log
python - <<'P'
import csv, glob
f = glob.glob('gpurun_out/gmprof/*/*_kernel_stats.csv')[0]
for i, r in enumerate(csv.DictReader(open(f))):
    if i >= 26: break
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]
    print(f"{n:42s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']:>6s}%")
P
