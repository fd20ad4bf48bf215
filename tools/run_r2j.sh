set -x
mkdir -p gpurun_out/r2j
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_grid_mc.py tests/test_gpu_genmesh.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r2j/pytest.log 2>&1; tail -5 gpurun_out/r2j/pytest.log
timeout -k 10 300 python - > gpurun_out/r2j/tv.txt 2>&1 <<'P'
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import pifuhd_amd
from pifuhd_amd import mesh_util
from pifuhd_amd.nets import _Ctx
ctx = _Ctx('cuda')
rs = np.random.RandomState(0)
v = torch.from_numpy((rs.rand(4_000_000, 3) * 511).astype(np.float32)).cuda()
for name, tm in (('diag', np.diag([2/512, -2/512, 2/512, 1.0]) + np.array([[0,0,0,-1],[0,0,0,1],[0,0,0,-1],[0,0,0,0.]])), ('random', rs.randn(4, 4))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got = mesh_util.transform_verts(ctx, v, tm)
    t1 = time.perf_counter()
    vh = v.cpu().numpy(); t2 = time.perf_counter()
    want = (np.matmul(tm[:3, :3], vh.T) + tm[:3, 3:4]).T
    t3 = time.perf_counter()
    print(name, 'device path %.4f s, numpy %.4f s (+ D2H %.4f), all equal: %s, fallbacks %d' % (t1 - t0, t3 - t2, t2 - t1, np.array_equal(got, want), mesh_util.transform_fallbacks))
P
cat gpurun_out/r2j/tv.txt
timeout -k 10 600 python bench.py --no-cpu-baseline --no-split-precision --no-lean-pass --no-canonical --no-cpu-genmesh > gpurun_out/r2j/bench.json 2> gpurun_out/r2j/bench.err; tail -2 gpurun_out/r2j/bench.err | cut -c1-200; python -c "
import json; d=json.load(open('gpurun_out/r2j/bench.json')); print(d['value'], d['gen_mesh'])"
