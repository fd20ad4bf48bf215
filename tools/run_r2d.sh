set -x
mkdir -p gpurun_out/r2d
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_genmesh.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r2d/pytest.log 2>&1; tail -15 gpurun_out/r2d/pytest.log
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2d/enc -- python tools/encoder_only.py > gpurun_out/r2d/enc.log 2>&1; grep "filter pass" gpurun_out/r2d/enc.log
