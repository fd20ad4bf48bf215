#!/bin/bash
# per-kernel time of marching cubes on a smooth and a noisy 512^3 volume (tools/mc_time.py under rocprofv3)
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/mcprof; rm -rf $OUT
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $ROOT/tools/mc_time.py ${1:-512} > $OUT.txt 2>&1
cd $ROOT; grep "\^3" $OUT.txt
python - <<'P'
import csv, glob
f = sorted(glob.glob("gpurun_out/mcprof/*/*_kernel_stats.csv"))[-1]
for i, r in enumerate(csv.DictReader(open(f))):
    if i >= 14: break
    print("%-34s calls %4s avg_us %8.1f min %8.1f max %8.1f" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
P
rm -rf $OUT
