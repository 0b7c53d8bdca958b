"""Summarise rocprofv3 counter CSVs of `tools/microbench/build/mb calib|sort` runs: per dispatch (in launch order) kernel name and the
summed counter value.  FETCH_SIZE / WRITE_SIZE are in KiB.

    python tools/pmc_calib.py <counter_collection.csv> [<counter name> ...]
"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    want = set(sys.argv[2:])
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if want and r["Counter_Name"] not in want:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        k = (int(r["Dispatch_Id"]), name, r["Counter_Name"])
        per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
    for (d, name, ctr), v in sorted(per.items()):
        print("%5d  %-52s %-22s %16.0f" % (d, name[:52], ctr, v))


if __name__ == "__main__":
    main()
