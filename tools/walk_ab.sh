#!/bin/bash
# k_walk variants of the lab build on the headline lattice: rocprofv3 average duration of the kernel per variant
# (0 = shipped <16,2>, 1 = round-1 kernel, 3 = <8,2>, 4 = <16,1>, 5 = <32,1>, 6 = adjacent pair, 8 = <32,2>) + the bench line's throughput
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/walk_ab
cd /tmp
for v in ${@:-0 1 3 4 5 6 8}; do
  OUT=$ROOT/gpurun_out/walk_ab/v$v; rm -rf $OUT
  PIFU_WALK_VARIANT=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $ROOT/tools/lab/bench_lab.py --steps 2 --warmup 1 --no-cpu-baseline --no-genmesh --no-split-precision --no-lean-pass --no-canonical > $OUT.json 2> $OUT.err || { tail -3 $OUT.err; exit 1; }
  python - $OUT $v <<'P'
import csv, glob, sys, json
f = glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'k_walk' in r['Name']:
        print('variant', sys.argv[2], r['Name'].split('(')[0][-24:], 'calls', r['Calls'], 'avg_us %.1f' % (float(r['AverageNs']) / 1e3))
line = [l for l in open(sys.argv[1] + '.json') if l.startswith('{')][-1]
print('   Mq/s', json.loads(line)['value'])
P
  rm -rf $OUT
done
