#!/bin/bash
# Marching cubes, per kernel, on a smooth and a noisy R^3 volume (tools/mc_time.py): kernel trace + three PMC passes
# (SQ_WAVE_CYCLES SQ_BUSY_CYCLES | FETCH_SIZE | WRITE_SIZE, each its own run as the pool requires), for the current library
# and for the round-2 build (tools/lab/r02/libpifuhd_hip_r02.so) when it is present.  Summary -> gpurun_out/mc_prof_R.txt
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=${1:-512}
SUM=$ROOT/gpurun_out/mc_prof_$R.txt
mkdir -p $ROOT/gpurun_out; : > $SUM
for which in now r02; do
    LIBARG=""
    if [ $which = r02 ]; then
        [ -f $ROOT/tools/lab/r02/libpifuhd_hip_r02.so ] || continue
        LIBARG="--lib $ROOT/tools/lab/r02/libpifuhd_hip_r02.so"
    fi
    OUT=$ROOT/gpurun_out/mcprof_$which; rm -rf $OUT; mkdir -p $OUT
    cd /tmp
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $ROOT/tools/mc_time.py $R $LIBARG > $OUT/trace.txt 2>&1 || exit 1
    timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python $ROOT/tools/mc_time.py $R $LIBARG > $OUT/sq.txt 2>&1 || exit 1
    timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python $ROOT/tools/mc_time.py $R $LIBARG > $OUT/fetch.txt 2>&1 || exit 1
    timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python $ROOT/tools/mc_time.py $R $LIBARG > $OUT/write.txt 2>&1 || exit 1
    cd $ROOT
    echo "== library: $which ==" >> $SUM
    grep "\^3" $OUT/trace.txt >> $SUM
    python tools/mc_prof_summary.py $OUT >> $SUM
    rm -rf $OUT
done
cat $SUM
