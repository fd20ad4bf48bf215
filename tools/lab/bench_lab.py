"""bench.py on the LAB library (tools/lab/libpifuhd_lab.so), for A/B runs that need the PIFU_* environment switches:
    PIFU_GEMM_VARIANT=28 python tools/lab/bench_lab.py --steps 1 ...
Never a headline measurement: the driver runs bench.py on the product library."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import lab  # noqa: E402

lab.use(experiments=bool(os.environ.get('PIFU_LAB_EXPERIMENTS')))
runpy.run_path(os.path.join(lab.ROOT, 'bench.py'), run_name='__main__')
