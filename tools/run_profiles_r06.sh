#!/bin/bash
# round-6 profile set -> gpurun_out/r06prof/ (copy into profiles/ with tools/collect_profiles_r06.sh afterwards):
#   (1) the bench line itself                                   -> r06_bench_line.json
#   (2) kernel trace + PMC passes of the headline leg           -> r06_kernel_stats.csv, r06_traffic.json
#   (3) gen_mesh 512^3 per kernel: headline and body-like subject -> r06_genmesh_kernel_stats.txt, r06_genmesh_body_kernel_stats.txt
#   (4) one steady-state encoder pass by kernel + timeline + un-profiled wall clocks -> r06_encoder_kernel_stats.txt
# STEP=bench|trace|genmesh|encoder|all selects what runs (the whole set exceeds one gpurun call's budget when the box is slow).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; O=gpurun_out/r06prof; mkdir -p $O
STEP=${STEP:-all}
if [ "$STEP" = all ] || [ "$STEP" = bench ]; then
  timeout -k 10 900 python bench.py > $O/r06_bench_line.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
fi
if [ "$STEP" = all ] || [ "$STEP" = trace ]; then
  bash tools/profile_gpu.sh > $O/prof_run.log 2>&1; tail -3 $O/prof_run.log
  python tools/summarize_profile.py gpurun_out/prof r06 > $O/summarize.log 2>&1; tail -3 $O/summarize.log
  cp profiles/r06_kernel_stats.csv $O/r06_kernel_stats.csv; cp profiles/traffic.json $O/r06_traffic.json
  rm -rf gpurun_out/prof/trace gpurun_out/prof/pmc_*
fi
if [ "$STEP" = all ] || [ "$STEP" = genmesh ]; then
  { echo "# bash tools/prof_genmesh.sh: gen_mesh 512^3 (the headline bench subject) - per call the executed FLOPs behind bench.py's gen_mesh.roofline, phase timings (two runs, no profiler), then rocprofv3 --kernel-trace --stats of the same script (one 128^3 + three 512^3 calls + one more octree / marching-cubes pass)"; bash tools/prof_genmesh.sh; } > $O/r06_genmesh_kernel_stats.txt 2>&1
  tail -12 $O/r06_genmesh_kernel_stats.txt
  { echo "# GENMESH_ARGS='--subject body' bash tools/prof_genmesh.sh: the same for the body-like subject (synth.craft_body_mlps)"; GENMESH_ARGS='--subject body' bash tools/prof_genmesh.sh; } > $O/r06_genmesh_body_kernel_stats.txt 2>&1
  tail -12 $O/r06_genmesh_body_kernel_stats.txt
  rm -rf gpurun_out/gmprof
fi
if [ "$STEP" = all ] || [ "$STEP" = encoder ]; then
  cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/encprof -- python $R/tools/encoder_only.py 3 > $R/$O/enc_run.txt 2>&1; cd $R
  T=$(ls gpurun_out/encprof/*/*kernel_trace.csv | head -1)
  { echo "# tools/run_profiles_r06.sh: rocprofv3 --kernel-trace -- python tools/encoder_only.py 3 (wall clocks UNDER the profiler), tools/encoder_pass_stats.py / encoder_timeline.py on one steady-state pass, then tools/encoder_phases.py WITHOUT the profiler"; grep -v amdgpu.ids $O/enc_run.txt; python tools/encoder_pass_stats.py $T; python tools/encoder_timeline.py $T; } > $O/r06_encoder_kernel_stats.txt 2>&1
  rm -rf gpurun_out/encprof
  timeout -k 10 100 python tools/encoder_phases.py 2>&1 | grep -v amdgpu.ids >> $O/r06_encoder_kernel_stats.txt
  tail -30 $O/r06_encoder_kernel_stats.txt
fi
