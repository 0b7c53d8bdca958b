"""Per DISPATCH of the kernels whose name contains argv[2]: the counters of a rocprofv3 --pmc counter_collection.csv, in dispatch order."""
import csv
import collections
import sys

rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r["Kernel_Name"]:
        continue
    key = int(r["Dispatch_Id"])
    d = rows.setdefault(key, {"name": r["Kernel_Name"].split("(")[0][-28:]})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in sorted(rows):
    d = rows[k]
    print(k, d.pop("name"), " ".join("%s=%.4g" % kv for kv in sorted(d.items())))
