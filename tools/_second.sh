set -x
out=gpurun_out/r05b
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_genmesh.py tests/test_gpu_input.py -m gpu -q -p no:cacheprovider -x > $out/pytest.log 2>&1; rc=$?; tail -5 $out/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/cold_start.py 512 2>&1 | grep -v amdgpu.ids > $out/cold_start.txt; cat $out/cold_start.txt
timeout -k 10 200 python tools/encoder_phases.py 2>&1 | grep -v amdgpu.ids > $out/encoder_phases.txt; cat $out/encoder_phases.txt
PIFU_BENCH_ONE_GPU=1 timeout -k 10 600 python bench.py --gpus 2 --no-cpu-baseline --config5-res 0 2> $out/bench_2ranks.err | grep '^{' > $out/bench_2ranks_one_gpu.json; tail -4 $out/bench_2ranks.err | cut -c1-200
