#!/bin/bash
# gpurun_out/r06prof/* (written on the GPU box by tools/run_profiles_r06.sh) -> profiles/ (tracked), then DESIGN.md section 6
cd $(dirname $0)/..
for f in r06_bench_line.json r06_kernel_stats.csv r06_traffic.json r06_genmesh_kernel_stats.txt r06_genmesh_body_kernel_stats.txt r06_encoder_kernel_stats.txt; do
  [ -s gpurun_out/r06prof/$f ] && cp gpurun_out/r06prof/$f profiles/$f
done
# the bench line: the last JSON line of stdout, pretty-printed
python - <<'P'
import json
p = 'profiles/r06_bench_line.json'
txt = open(p).read().strip()
d = json.loads(txt.splitlines()[-1]) if not txt.startswith('{\n') else json.loads(txt)
json.dump(d, open(p, 'w'), indent=1)
P
cp profiles/r06_traffic.json profiles/traffic.json
python tools/check_design_numbers.py --write && python tools/check_design_numbers.py
