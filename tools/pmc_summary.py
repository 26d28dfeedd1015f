"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (development aid)."""
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k][r['Counter_Name']] += 1
for k in sorted(agg):
    if 'at::' in k or 'rocclr' in k: continue
    print(k)
    for c in sorted(agg[k]):
        print(f"   {c:32s} {agg[k][c] / cnt[k][c]:18.1f}  (avg over {cnt[k][c]} dispatches)")
