#!/bin/bash
# samples power / clocks with rocm-smi while the headline bench runs (diagnostic: is the fp32 MFMA path power-capped?)
mkdir -p gpurun_out
( for i in $(seq 1 80); do rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -i "sclk\|power\|mclk" | tr '\n' '|'; echo; sleep 0.5; done ) > gpurun_out/clock_watch.log 2>&1 &
W=$!
python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-genmesh --no-split-precision > gpurun_out/bench_clock.json 2> gpurun_out/bench_clock.err
kill $W 2>/dev/null
tail -n 30 gpurun_out/clock_watch.log | cut -c1-400
cat gpurun_out/bench_clock.json | cut -c1-300
