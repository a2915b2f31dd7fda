"""Average rocprofv3 --pmc counters per kernel.  usage: pmc_summary.py <counter_collection.csv> [name filter]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    if flt and flt not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    for c, v in sorted(d.items()):
        print("%-60s %-32s %16.1f  (n=%d)" % (k, c, v / cnt[(k, c)], cnt[(k, c)]))
