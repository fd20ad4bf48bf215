#!/bin/bash
# kernel trace of the profiling tool on one subject (default: body-like), then the per-launch views of the last gen_mesh call
# usage: SUBJECT=body|headline|pt_body bash tools/trace_body.sh      (run from the repository root on the GPU box)
export TMPDIR=/tmp
SUBJECT=${SUBJECT:-body}
OUT=gpurun_out/cw_$SUBJECT
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python tools/genmesh_profile.py 512 --subject $SUBJECT --calls 2 > $OUT/out.log 2> $OUT/err.log
f=$(ls $OUT/*/*_kernel_trace.csv | head -1)
python tools/call_window.py $f > $OUT/window.txt
python tools/walk_launches.py $f > $OUT/walk.txt
tail -45 $OUT/window.txt
