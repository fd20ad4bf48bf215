set -x
mkdir -p gpurun_out/r04g
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_query.py tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider -x --durations=5 > gpurun_out/r04g/pytest.log 2>&1; rc=$?; tail -10 gpurun_out/r04g/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --no-cpu-genmesh > gpurun_out/r04g/bench.json 2> gpurun_out/r04g/bench.err; tail -6 gpurun_out/r04g/bench.err | cut -c1-250
