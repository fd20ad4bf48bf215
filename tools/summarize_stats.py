"""Per-kernel totals of a rocprofv3 --kernel-trace --stats output directory (any *kernel_stats.csv below it): name (template
arguments and parameter lists cut), calls, average us, total ms, share."""
import csv
import glob
import re
import sys
rows = {}
for path in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        name = re.sub(r'\(.*', '', r['Name'].replace('(anonymous namespace)::', '').replace('void ', ''))
        c, t = rows.get(name, (0, 0.0))
        rows[name] = (c + int(r['Calls']), t + float(r['TotalDurationNs']))
tot = sum(t for _, t in rows.values()) or 1.0
for name, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 28]:
    print(f'{name[:60]:60s} calls {c:6d} avg_us {t / c / 1e3:9.1f} total_ms {t / 1e6:9.2f} {100 * t / tot:6.2f}%')
