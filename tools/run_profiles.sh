set -x
export TMPDIR=/tmp
bash tools/profile_gpu.sh > gpurun_out/prof_run.log 2>&1
tail -3 gpurun_out/prof_run.log
bash tools/prof_genmesh.sh > gpurun_out/genmesh_r02.txt 2>&1; tail -26 gpurun_out/genmesh_r02.txt
mkdir -p gpurun_out/encprof; cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/encprof -- python $GRAFT_REPO_ROOT/tools/encoder_only.py > $GRAFT_REPO_ROOT/gpurun_out/encprof/out.txt 2>&1; cd $GRAFT_REPO_ROOT; grep "filter pass" gpurun_out/encprof/out.txt
