set -x
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "column_sharing or depth_walk or nontrivial or chunked_c_abi or large_chunk or bit_identical" > gpurun_out/r2b/pytest.log 2>&1; tail -5 gpurun_out/r2b/pytest.log
for v in 42 40 41; do timeout -k 10 200 python tools/ab_gemm.py $v 128 >> gpurun_out/r2b/ab.log 2>&1; done; cat gpurun_out/r2b/ab.log
timeout -k 10 400 bash tools/layer_ab.sh 42 40 41 > gpurun_out/r2b/layers.log 2>&1; cat gpurun_out/r2b/layers.log
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2b/enc -- python tools/encoder_only.py > gpurun_out/r2b/enc.log 2>&1; tail -4 gpurun_out/r2b/enc.log
