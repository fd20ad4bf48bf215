set -x
mkdir -p gpurun_out/r2e
timeout -k 10 800 python -m pytest tests/test_gpu_genmesh.py tests/test_gpu_query.py -m gpu -q -p no:cacheprovider -k "imgcolor_plus or meshcleaning" > gpurun_out/r2e/pytest.log 2>&1; tail -30 gpurun_out/r2e/pytest.log
