set -x
mkdir -p gpurun_out/r2g
export TMPDIR=/tmp
timeout -k 10 600 python bench.py > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err && tail -3 gpurun_out/r2g/bench.err &&
timeout -k 10 300 python tools/genmesh_host_breakdown.py > gpurun_out/r2g/host_breakdown.txt 2>&1 ; tail -30 gpurun_out/r2g/host_breakdown.txt
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2g/enctrace -- python $GRAFT_REPO_ROOT/tools/encoder_only.py > $GRAFT_REPO_ROOT/gpurun_out/r2g/enc_only.txt 2>&1
cd $GRAFT_REPO_ROOT && f=$(ls gpurun_out/r2g/enctrace/*/*kernel_trace.csv | head -1) && python tools/encoder_trace.py $f 20 > gpurun_out/r2g/enc_layers.txt 2>&1; tail -5 gpurun_out/r2g/enc_layers.txt; rm -rf gpurun_out/r2g/enctrace
