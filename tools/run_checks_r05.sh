# one gpurun call: smoke, the GPU suite, the bench line (N = 1) and the two-rank one-GPU rehearsal of the N > 1 line
set -x
out=gpurun_out/r05
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout -k 10 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x --durations=10 ${PYTEST_ARGS:-} > $out/pytest.log 2>&1; rc=$?; tail -16 $out/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 700 python bench.py > $out/bench.json 2> $out/bench.err; tail -8 $out/bench.err | cut -c1-250
PIFU_BENCH_ONE_GPU=1 timeout -k 10 600 python bench.py --gpus 2 --no-cpu-baseline > $out/bench_2ranks_one_gpu.json 2> $out/bench_2ranks.err; tail -6 $out/bench_2ranks.err | cut -c1-250
