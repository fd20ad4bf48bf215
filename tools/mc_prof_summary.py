"""Per-kernel summary of tools/mc_prof.sh's four rocprofv3 runs: average duration (kernel trace), SQ_WAVE_CYCLES / SQ_BUSY_CYCLES
/ GRBM_GUI_ACTIVE, FETCH_SIZE and WRITE_SIZE per launch (averaged over the launches of the run; the script times each volume
three times).  FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md, HBM section); units: the
counters' own (FETCH / WRITE in KB as rocprofv3 reports them -> shown in MB)."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]


def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:30]


def counters(sub):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in glob.glob(f'{out}/{sub}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[k][r['Counter_Name']] += 1
    return {k: {c: acc[k][c] / max(cnt[k][c], 1) for c in acc[k]} for k in acc}


stats = {}
for f in glob.glob(f'{out}/trace/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        stats[short(r['Name'])] = (int(r['Calls']), float(r['AverageNs']) / 1e3)
sq, fe, wr = counters('sq'), counters('fetch'), counters('write')
print('%-30s %5s %9s %12s %12s %10s %10s' % ('kernel', 'calls', 'avg_us', 'wave_cyc(4x)', 'busy_cyc', 'fetch_MB', 'write_MB'))
tot = 0.0
for k, (calls, us) in sorted(stats.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    if not (k.startswith('k_mc') or k.startswith('k_scan')):
        continue
    tot += calls * us
    s = sq.get(k, {})
    print('%-30s %5d %9.1f %12.3g %12.3g %10.1f %10.1f' % (k, calls, us, s.get('SQ_WAVE_CYCLES', 0), s.get('SQ_BUSY_CYCLES', 0),
                                                          2 * fe.get(k, {}).get('FETCH_SIZE', 0) / 1024, wr.get(k, {}).get('WRITE_SIZE', 0) / 1024))
print('sum over the run: %.1f us (6 marching-cubes calls: 3 smooth + 3 noisy)' % tot)
