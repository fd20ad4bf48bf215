set -x
mkdir -p gpurun_out/r04a
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_query.py tests/test_abi.py -m gpu -q -p no:cacheprovider -x --durations=8 > gpurun_out/r04a/pytest_query.log 2>&1; rc=$?; tail -15 gpurun_out/r04a/pytest_query.log
[ $rc -eq 0 ] && timeout -k 10 300 python tools/normals_ab.py > gpurun_out/r04a/normals_ab.txt 2>&1; tail -9 gpurun_out/r04a/normals_ab.txt
