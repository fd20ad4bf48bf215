#!/usr/bin/env python3
"""Summarise rocprofv3 output (gpurun_out/prof/*) into profiles/: per-kernel stats and the HBM traffic of
k_mlp_gemm per launch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of the
bytes of wide coalesced streaming reads -> doubled; WRITE_SIZE is exact; both are in KiB."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof'
tag = sys.argv[2] if len(sys.argv) > 2 else 'r02'
os.makedirs('profiles', exist_ok=True)


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    name = name.split('(')[0]
    return name.split('<')[0][:60] if name.startswith('k_') else name[:60]


def is_col(name):
    return 'k_sample_col' in name


# (1) kernel stats
stats = sorted(glob.glob(os.path.join(src, 'trace', '*', '*_kernel_stats.csv')), key=os.path.getmtime, reverse=True)
rows = list(csv.DictReader(open(stats[0]))) if stats else []
cmd_file = os.path.join(src, 'trace_command.txt')
command = open(cmd_file).read().strip() if os.path.exists(cmd_file) else 'python bench.py ... (tools/profile_gpu.sh)'
gemm_rows = [r for r in rows if 'k_mlp_gemm' in r['Name']]
mix = ''
if gemm_rows:
    calls = sum(int(r['Calls']) for r in gemm_rows)
    mix = (f'# k_mlp_gemm_tri launch mix of this trace: {calls} launches = (1 warm-up + 2 timed) passes over the 256^3 lattice, each 2 '
           f'column-table launches + 9 batches (batch_rows 2 097 152) x 6 layer launches (= 168, the headline mix, chunk 5000) + {calls - 168} '
           f'launches of the set-up call (calibrate_occupancy_bias on a 64^3 lattice, pixel-table-free); no other leg of bench.py runs in this trace\n')
with open(f'profiles/{tag}_kernel_stats.csv', 'w') as f:
    f.write(f'# rocprofv3 --kernel-trace --stats -- {command}   (MI355X, 1 GPU)\n')
    f.write(mix)
    if len(gemm_rows) > 1:
        f.write('# k_mlp_gemm_tri appears once per template instantiation: the larger row is the one with the statistics epilogue (layer launches under '
                'GroupNorm), the smaller one stores only (2 column-table launches per pass + the depth walk\'s table launch per batch + set-up)\n')
    f.write('Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n')
    for r in rows[:16]:
        full = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        name = full[:60].replace(', ', '/').replace(',', '/') if full.startswith('k_mlp_gemm') else short(r['Name'])      # the GEMM's instantiations stay apart: <PRIO, QREL, FUSE>
        f.write(f"{name},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")


def pmc(dirname):
    out = defaultdict(lambda: defaultdict(list))
    for p in sorted(glob.glob(os.path.join(src, dirname, '*', '*_counter_collection.csv')), key=os.path.getmtime, reverse=True)[:1]:
        for r in csv.DictReader(open(p)):
            out[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return out


fetch, write, sq = pmc('pmc_fetch'), pmc('pmc_write'), pmc('pmc_sq')
fetch_c, write_c = pmc('pmc_fetch_canon'), pmc('pmc_write_canon')      # passes with the canonical per-point pass on
summary = {}
GEMM = next((k for k in ('k_mlp_gemm_tri', 'k_mlp_gemm_buf') if k in fetch), 'k_mlp_gemm')
for k in (GEMM, 'k_walk', 'k_sample', 'k_sample_col', 'k_seg_prepare', 'k_last_layer'):
    f = (fetch_c if k == 'k_sample' else fetch).get(k, {}).get('FETCH_SIZE', [])
    w = (write_c if k == 'k_sample' else write).get(k, {}).get('WRITE_SIZE', [])
    if k != GEMM and f and w:      # small set-up launches would dilute the per-launch figure of the batch-sized ones
        f = [x for x in f if x > 0.5 * max(f)]
        w = [x for x in w if x > 0.5 * max(w)]
    if f and w:
        fb = 2.0 * 1024.0 * sum(f) / len(f)      # KiB -> bytes, x2 gfx950 correction
        wb = 1024.0 * sum(w) / len(w)
        summary[k] = {'launches': len(f), 'fetch_bytes_per_launch_corrected': fb, 'write_bytes_per_launch': wb,
                      'hbm_bytes_per_launch': fb + wb}
g = sq.get(GEMM, {})
if g:
    summary[GEMM + '_sq'] = {c: sum(v) / len(v) for c, v in g.items()}
    q = summary[GEMM + '_sq']
    if q.get('GRBM_GUI_ACTIVE') and q.get('SQ_VALU_MFMA_BUSY_CYCLES'):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs: busy share of a SIMD's matrix
        # pipe while the kernel runs (counts the padded rows' MFMAs too, unlike the algorithmic-FLOP figure of bench.py)
        q['mfma_pipe_busy_fraction'] = q['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (q['GRBM_GUI_ACTIVE'] / 8.0)
json.dump({'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 1 --warmup 0 (the headline lattice and batch '
                     'geometry: per-launch averages over the same launch mix as the timed run)',
           'gemm_kernel': GEMM,
           'k_mlp_gemm_hbm_bytes_per_launch': summary.get(GEMM, {}).get('hbm_bytes_per_launch'),
           # per-point sampler of the canonical pass (the batch-sized launches only)
           'k_sample_hbm_bytes_per_launch': summary.get('k_sample', {}).get('hbm_bytes_per_launch'),
           'detail': summary}, open('profiles/traffic.json', 'w'), indent=1)
print(json.dumps(summary, indent=1))
