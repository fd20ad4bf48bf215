set -x
out=gpurun_out/r05c
mkdir -p $out
export TMPDIR=/tmp
R=$(pwd)
timeout -k 10 200 python tools/encoder_phases.py 2>&1 | grep -v amdgpu.ids > $out/encoder_phases.txt; cat $out/encoder_phases.txt
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/encprof -- python $R/tools/encoder_only.py 3 > $R/$out/enc_run.txt 2>&1; cd $R
T=$(ls gpurun_out/encprof/*/*kernel_trace.csv | head -1)
{ python tools/encoder_pass_stats.py $T; python tools/encoder_timeline.py $T; } > $out/encoder_timeline.txt 2>&1
rm -rf gpurun_out/encprof
tail -50 $out/encoder_timeline.txt
