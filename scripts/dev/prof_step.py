"""Per-step kernel table from a rocprofv3 kernel_stats.csv of a launch-by-launch bench run: only this library's kernels (k_*),
calls normalised to the most common call count (= steps).  usage: prof_step.py kernel_stats.csv"""
import csv
import re
import sys
from collections import Counter

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Name']
    m = re.search(r'(k_[a-z0-9_]+)(<[^>]*>)?|_Z\d+(k_[a-z0-9_]+)(I[A-Za-z0-9_]*E)?', name)
    if not m:
        continue
    short = (m.group(1) or m.group(3)) + (m.group(2) or (m.group(4) or ''))
    rows.append((short, int(r['Calls']), float(r['TotalDurationNs']), float(r['AverageNs'])))
if not rows:
    sys.exit('no k_* kernels')
steps = Counter(c for _, c, _, _ in rows).most_common(1)[0][0]
tot = 0.0
for short, calls, total, avg in sorted(rows, key=lambda t: -t[2]):
    per = total / steps / 1e3
    tot += per
    print(f'  {short:34s} x{calls / steps:4.1f}  {avg / 1e3:8.1f} us  {per:8.1f} us/step')
print(f'  sum {tot:.1f} us/step over {steps} steps')
