set -x
mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/r2c/pytest.log 2>&1; tail -15 gpurun_out/r2c/pytest.log
timeout -k 10 600 python bench.py > gpurun_out/r2c/bench.json 2> gpurun_out/r2c/bench.err; tail -5 gpurun_out/r2c/bench.err; cat gpurun_out/r2c/bench.json
