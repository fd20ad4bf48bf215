#!/bin/bash
# GEMM variants of the lab build: bit-identity (sha of a 128^3 field, column-sharing and per-point paths) and the bench line
# usage: bash tools/gemm_ab.sh <variant> ...   (42 = shipped tri, 43 / 44 = 128 x 256 x 32 wide tile without / with s_setprio)
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  timeout -k 10 200 python $ROOT/tools/ab_gemm.py $v 128 2>&1 | grep "^variant"
  PIFU_GEMM_VARIANT=$v timeout -k 10 300 python $ROOT/tools/lab/bench_lab.py --steps 3 --warmup 1 --no-cpu-baseline --no-genmesh --no-split-precision --no-lean-pass > $ROOT/gpurun_out/gemm_ab_$v.json 2> $ROOT/gpurun_out/gemm_ab_$v.err || tail -3 $ROOT/gpurun_out/gemm_ab_$v.err
  python - $ROOT/gpurun_out/gemm_ab_$v.json $v <<'P'
import json, sys
d = json.load(open(sys.argv[1])); r = d['roofline']; c = d['canonical_path']
print('variant', sys.argv[2], 'headline %.2f Mq/s, GEMM %.1f TFLOP/s (%.3f; clock %.0f MHz -> %.3f); canonical %.2f Mq/s, GEMM %.1f' % (
    d['value'], r['achieved'], r['frac'], r['shader_clock_mhz_measured'], r['frac_at_measured_clock'], c['value'], c['gemm_roofline']['achieved']))
P
done
