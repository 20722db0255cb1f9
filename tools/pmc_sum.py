"""pmc_sum.py DIR — sum the counters of a `rocprofv3 --pmc ... --output-format csv -d DIR` run per kernel (counter_collection.csv)."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], k) not in seen:
            seen.add((r["Dispatch_Id"], k)); cnt[k] += 1
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:8]:
    print(k, "dispatches", cnt[k], {c: v for c, v in sorted(d.items())})
