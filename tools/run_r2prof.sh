set -x
export TMPDIR=/tmp
bash tools/profile_gpu.sh > gpurun_out/prof_run.log 2>&1
tail -3 gpurun_out/prof_run.log
