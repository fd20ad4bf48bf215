set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/enc
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/enc/trace -- python $GRAFT_REPO_ROOT/tools/encoder_only.py > $GRAFT_REPO_ROOT/gpurun_out/enc/out.txt 2>&1
cd $GRAFT_REPO_ROOT && f=$(ls gpurun_out/enc/trace/*/*kernel_trace.csv | head -1) && python tools/encoder_hist.py $f > gpurun_out/enc/hist.txt 2>&1; cat gpurun_out/enc/hist.txt; grep "filter pass" gpurun_out/enc/out.txt; rm -rf gpurun_out/enc/trace
