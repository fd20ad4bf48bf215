set -x
mkdir -p gpurun_out/checks
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/checks/smoke.log 2>&1; tail -2 gpurun_out/checks/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6 > gpurun_out/checks/pytest.log 2>&1; tail -12 gpurun_out/checks/pytest.log
timeout -k 10 600 python bench.py > gpurun_out/checks/bench.json 2> gpurun_out/checks/bench.err; tail -4 gpurun_out/checks/bench.err | cut -c1-250
