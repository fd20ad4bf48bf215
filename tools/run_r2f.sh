set -x
mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/r2f/pytest.log 2>&1; tail -15 gpurun_out/r2f/pytest.log
timeout -k 10 600 python bench.py > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err && tail -5 gpurun_out/r2f/bench.err && cat gpurun_out/r2f/bench.json &&
timeout -k 10 300 python tools/encoder_trace.py > gpurun_out/r2f/enc_trace.txt 2>&1; tail -5 gpurun_out/r2f/enc_trace.txt
