#!/bin/bash
# round-5 profile set -> gpurun_out/ (copy the summaries into profiles/ afterwards): bench kernel trace + PMC passes,
# gen_mesh kernel stats, encoder steady-state pass + timeline, marching cubes with counters (before / after)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
bash tools/profile_gpu.sh > gpurun_out/prof_run.log 2>&1; tail -3 gpurun_out/prof_run.log
python tools/summarize_profile.py gpurun_out/prof r05 > gpurun_out/summarize.log 2>&1; tail -5 gpurun_out/summarize.log
cp profiles/r05_kernel_stats.csv gpurun_out/r05_kernel_stats.csv; cp profiles/traffic.json gpurun_out/r05_traffic.json
bash tools/prof_genmesh.sh > gpurun_out/genmesh_r05.txt 2>&1; tail -30 gpurun_out/genmesh_r05.txt
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/encprof -- python $R/tools/encoder_only.py 3 > $R/gpurun_out/enc_run.txt 2>&1; cd $R
T=$(ls gpurun_out/encprof/*/*kernel_trace.csv | head -1)
{ grep -v amdgpu.ids gpurun_out/enc_run.txt; python tools/encoder_pass_stats.py $T; python tools/encoder_timeline.py $T; } > gpurun_out/encoder_r05.txt 2>&1
rm -rf gpurun_out/encprof gpurun_out/prof/trace gpurun_out/prof/pmc_*
timeout -k 10 100 python tools/encoder_phases.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/encoder_r05.txt
tail -40 gpurun_out/encoder_r05.txt
# calc_normal phase: pixel tables against the per-point GEMM path, alternating in one process, then per kernel
timeout -k 10 300 python tools/normals_ab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/normals_r05.txt
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/normprof -- python3 $R/tools/normals_ab.py 500000 > /dev/null 2>&1; cd $R
python tools/summarize_stats.py gpurun_out/normprof 14 >> gpurun_out/normals_r05.txt 2>&1; rm -rf gpurun_out/normprof
tail -24 gpurun_out/normals_r05.txt
