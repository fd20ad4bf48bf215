#!/bin/bash
# round 6: the body-like subject - sharpness sweep, kernel profile of one gen_mesh call, then the whole bench line
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out/r06e
python tools/subject_explore.py --res 512 group:0.16:crafted:0:1:0.15 group:0.16:crafted:0:1:0.3 group:0.16:crafted:0:1:0.6 group:0.16:crafted:0:1:1 no:0.10:crafted:0:1:0.3 2>&1 | grep -v amdgpu.ids | cut -c1-520 > gpurun_out/r06e/explore.log
cat gpurun_out/r06e/explore.log
GENMESH_ARGS='--subject body' bash tools/prof_genmesh.sh > gpurun_out/r06e/genmesh_body.txt 2>&1
tail -34 gpurun_out/r06e/genmesh_body.txt
timeout -k 10 900 python bench.py > gpurun_out/r06e/bench.json 2> gpurun_out/r06e/bench.err
tail -5 gpurun_out/r06e/bench.err
python - <<'P'
import json
d = json.loads(open('gpurun_out/r06e/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], d['roofline']['frac'])
g = d['gen_mesh']
print('gen_mesh', g.get('total_s'), g.get('roofline', {}).get('frac'), g.get('error'))
print('body', json.dumps(g.get('body_subject'))[:1500])
print('pt', json.dumps(d.get('pifuhd_pt_config'))[:3000])
print('cpu', d.get('cpu_baseline'))
P
