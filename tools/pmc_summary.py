"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel name: sum of each counter + dispatch count."""
import csv
import collections
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0][-40:]
    rows[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], name)
    if key not in seen:
        seen.add(key)
        calls[name] += 1
for name, c in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    print(name, "calls", calls[name], " ".join("%s=%.4g" % kv for kv in sorted(c.items())))
