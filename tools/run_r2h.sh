set -x
mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_grid_mc.py tests/test_gpu_encoder.py tests/test_gpu_genmesh.py -m gpu -q -p no:cacheprovider -k "walk or column or lattice or full_size or config3 or rgbd or gen_mesh or calib" > gpurun_out/r2h/pytest.log 2>&1; tail -8 gpurun_out/r2h/pytest.log
FEATURES=filter timeout -k 10 300 bash tools/prof_kernels.sh > gpurun_out/r2h/kprof.txt 2>&1; head -16 gpurun_out/r2h/kprof.txt
timeout -k 10 200 python tools/ab_walk.py 256 > gpurun_out/r2h/ab_walk.txt 2>&1; cat gpurun_out/r2h/ab_walk.txt
timeout -k 10 600 python bench.py > gpurun_out/r2h/bench.json 2> gpurun_out/r2h/bench.err; tail -3 gpurun_out/r2h/bench.err; cut -c1-400 gpurun_out/r2h/bench.json
