set -x
mkdir -p gpurun_out/r2k
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests/test_gpu_genmesh.py tests/test_gpu_dist.py tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r2k/pytest.log 2>&1; tail -5 gpurun_out/r2k/pytest.log
for i in 1 2; do timeout -k 10 600 python bench.py --no-cpu-baseline --no-split-precision --no-lean-pass --no-canonical --no-cpu-genmesh > gpurun_out/r2k/bench$i.json 2> gpurun_out/r2k/bench$i.err; python -c "
import json; d=json.load(open('gpurun_out/r2k/bench$i.json')); print(d['value'], {k:v for k,v in d['gen_mesh'].items() if k!='mc_check'})"; done
