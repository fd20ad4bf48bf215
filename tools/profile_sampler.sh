#!/bin/bash
# HBM traffic and duration of k_sample (per-point sampler: projection + 4-tap bilinear gather of 256 + 16 channels): trace pass + two PMC passes
export TMPDIR=/tmp
OUT=gpurun_out/sprof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python tools/sampler_only.py > /dev/null 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python tools/sampler_only.py > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python tools/sampler_only.py > /dev/null 2> $OUT/write.err
python - <<'P'
import csv, glob
def col(path, pat, key):
    out = []
    for f in glob.glob(path):
        for r in csv.DictReader(open(f)):
            if pat in r.get('Kernel_Name', ''):
                out.append(r)
    return out
tr = [r for r in col('gpurun_out/sprof/trace/*/*_kernel_trace.csv', 'k_sample', None) if 'k_sample_col' not in r['Kernel_Name']]
dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in tr][1:]
f = [float(r['Counter_Value']) for r in col('gpurun_out/sprof/fetch/*/*_counter_collection.csv', 'k_sample', None) if r['Counter_Name'] == 'FETCH_SIZE' and 'k_sample_col' not in r['Kernel_Name']][1:]
w = [float(r['Counter_Value']) for r in col('gpurun_out/sprof/write/*/*_counter_collection.csv', 'k_sample', None) if r['Counter_Name'] == 'WRITE_SIZE' and 'k_sample_col' not in r['Kernel_Name']][1:]
d = sum(dur) / len(dur); fb = 2 * 1024 * sum(f) / len(f); wb = 1024 * sum(w) / len(w)
print('k_sample: %d launches, avg %.1f us per 2 M points; HBM fetch %.3f GB (gfx950-corrected x2), write %.3f GB per launch -> %.2f TB/s' % (len(dur), d, fb / 1e9, wb / 1e9, (fb + wb) / d / 1e6))
print('algorithmic: gather 4352 B/query from L2-resident maps (33.6 MB), 12 B in + 1280 B out per query -> %.3f GB per launch; gather rate %.2f TB/s' % (2e6 * 1292 / 1e9, 2e6 * 4352 / d / 1e6))
P
