"""Average the rocprofv3 --pmc counter_collection.csv files under a directory, per kernel."""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k[:90])
    for c, x in sorted(v.items()):
        print(f"    {c:34s} {sum(x) / len(x):16.0f}  (n={len(x)})")
