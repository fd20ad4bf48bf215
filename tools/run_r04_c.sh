set -x
mkdir -p gpurun_out/r04c
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_genmesh.py tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider -x --durations=6 > gpurun_out/r04c/pytest.log 2>&1; rc=$?; tail -12 gpurun_out/r04c/pytest.log
[ $rc -eq 0 ] || exit 1
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04c/prof_normals -- python3 $GRAFT_REPO_ROOT/tools/normals_ab.py 500000 > $GRAFT_REPO_ROOT/gpurun_out/r04c/normals_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_stats.py gpurun_out/r04c/prof_normals > gpurun_out/r04c/normals_kernels.txt 2>&1; head -30 gpurun_out/r04c/normals_kernels.txt
rm -rf gpurun_out/r04c/prof_normals
timeout -k 10 600 python bench.py > gpurun_out/r04c/bench.json 2> gpurun_out/r04c/bench.err; tail -5 gpurun_out/r04c/bench.err | cut -c1-250
