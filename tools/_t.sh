export TMPDIR=/tmp
mkdir -p gpurun_out/r05h
timeout -k 10 500 python tools/soak_sizes.py 45 2>&1 | grep -v amdgpu.ids > gpurun_out/r05h/soak.txt; tail -12 gpurun_out/r05h/soak.txt
timeout -k 10 300 python tools/determinism_soak.py 5 2>&1 | grep -v amdgpu.ids > gpurun_out/r05h/determinism.txt; tail -4 gpurun_out/r05h/determinism.txt
