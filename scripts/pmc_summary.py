"""Sum rocprofv3 counter_collection.csv per kernel (short kernel names)."""
import collections, csv, glob, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_]+(?:<[a-z]+>)?)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); calls[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    print(k, "calls", calls[k], {n: f"{x:.4g}" for n, x in v.items()})
