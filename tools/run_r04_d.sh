set -x
mkdir -p gpurun_out/r04d
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_query.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r04d/pytest.log 2>&1; rc=$?; tail -4 gpurun_out/r04d/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/normals_ab.py > gpurun_out/r04d/normals_ab.txt 2>&1; tail -8 gpurun_out/r04d/normals_ab.txt
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04d/prof_normals -- python3 $GRAFT_REPO_ROOT/tools/normals_ab.py 500000 > $GRAFT_REPO_ROOT/gpurun_out/r04d/normals_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_stats.py gpurun_out/r04d/prof_normals > gpurun_out/r04d/normals_kernels.txt 2>&1; head -16 gpurun_out/r04d/normals_kernels.txt
rm -rf gpurun_out/r04d/prof_normals
