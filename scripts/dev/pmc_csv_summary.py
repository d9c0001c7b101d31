"""Average rocprofv3 --pmc counters per kernel from counter_collection.csv files under a directory.  usage: pmc_csv_summary.py dir [substr]"""
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
sub = sys.argv[2] if len(sys.argv) > 2 else ''
for k, cs in acc.items():
    if sub in k:
        print(k[:100])
        for c, v in sorted(cs.items()):
            print(f'   {c:28s} n={len(v):3d} avg={sum(v)/len(v):.4g}')
