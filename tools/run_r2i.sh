set -x
mkdir -p gpurun_out/r2i
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x --durations=5 > gpurun_out/r2i/pytest.log 2>&1; tail -12 gpurun_out/r2i/pytest.log
timeout -k 10 200 python tools/ab_walk.py 256 > gpurun_out/r2i/ab_walk.txt 2>&1; cat gpurun_out/r2i/ab_walk.txt
timeout -k 10 600 python bench.py > gpurun_out/r2i/bench.json 2> gpurun_out/r2i/bench.err; tail -3 gpurun_out/r2i/bench.err; cut -c1-300 gpurun_out/r2i/bench.json
