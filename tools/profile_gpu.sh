#!/bin/bash
# rocprofv3 passes for profiles/: (1) kernel trace + stats of the bench command, (2)+(3) PMC passes (HBM bytes).
# Counters are collected in their own runs (no trace domains besides --kernel-trace), as the pool requires.
set -x
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
# only the headline leg runs: every k_mlp_gemm_tri launch of this trace belongs to the headline launch mix (chunk 5000), except the
# 7 small set-up launches of calibrate_occupancy_bias
BENCH="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-genmesh --no-split-precision --no-lean-pass --no-canonical --no-chunk100k --no-encoder-leg"
echo "python $BENCH" > $OUT/trace_command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err
# same lattice and batch geometry as the headline run, so per-launch averages are over the same launch mix
SMALL="bench.py --steps 1 --warmup 0 --no-canonical --no-cpu-baseline --no-genmesh --no-split-precision --no-lean-pass --no-chunk100k --no-encoder-leg"
# the canonical per-point pass launches the batch-sized k_sample: its counters come from two extra passes, so that the GEMM
# averages above stay over the headline launch mix
# (without the calc_normal leg: its per-point pass launches k_sample on RANDOM vertices, whose gathers miss the L2 - the figure
# wanted here is the lattice pass's)
CANON="bench.py --steps 1 --warmup 0 --canonical-steps 1 --no-cpu-baseline --no-genmesh --no-split-precision --no-lean-pass --no-chunk100k --no-encoder-leg --no-calc-normal-leg"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python $SMALL > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python $SMALL > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $OUT/pmc_sq -- python $SMALL > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_canon -- python $CANON > $OUT/bench_pmc_fetch_canon.json 2> $OUT/pmc_fetch_canon.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_canon -- python $CANON > $OUT/bench_pmc_write_canon.json 2> $OUT/pmc_write_canon.err
for f in $OUT/*.err; do tail -n 1 $f; done
du -sh $OUT
