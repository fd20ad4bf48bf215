set -x
mkdir -p gpurun_out/r04b
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04b/prof_normals -o normals -- python3 $GRAFT_REPO_ROOT/tools/normals_ab.py 500000 > $GRAFT_REPO_ROOT/gpurun_out/r04b/normals_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_stats.py gpurun_out/r04b/prof_normals > gpurun_out/r04b/normals_kernels.txt 2>&1; head -30 gpurun_out/r04b/normals_kernels.txt
rm -rf gpurun_out/r04b/prof_normals
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider -x --durations=10 > gpurun_out/r04b/pytest.log 2>&1; rc=$?; tail -16 gpurun_out/r04b/pytest.log
[ $rc -eq 0 ] && timeout -k 10 600 python bench.py > gpurun_out/r04b/bench.json 2> gpurun_out/r04b/bench.err; tail -5 gpurun_out/r04b/bench.err | cut -c1-250
