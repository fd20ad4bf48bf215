set -x
mkdir -p gpurun_out/r04f
export TMPDIR=/tmp
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04f/smoke.log 2>&1; tail -2 gpurun_out/r04f/smoke.log
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider -x --durations=8 > gpurun_out/r04f/pytest.log 2>&1; rc=$?; tail -14 gpurun_out/r04f/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py > gpurun_out/r04f/bench.json 2> gpurun_out/r04f/bench.err; tail -6 gpurun_out/r04f/bench.err | cut -c1-250
