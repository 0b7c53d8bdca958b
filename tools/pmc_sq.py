"""profiles/sq_counters.json: the issue-side view of every kernel family from rocprofv3's derived metrics, collected over
`bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run` in passes of at most three metrics (no trace domain but --kernel-trace):

    python tools/pmc_sq.py <out.json> <batch_reads> <read_len> <db_nt> <counter_collection.csv> [<counter_collection.csv> ...]

  VALUBusy          % of the cycles a SIMD's vector ALU was issuing               (what north_star's "VALU issue fraction" asks for)
  SALUBusy          the same for the scalar ALU
  VALUUtilization   % of the 64 lanes active in the vector instructions issued
  LDSBankConflict   % of the LDS cycles lost to bank conflicts                      (north_star: "LDS-hit counters")
  MemUnitStalled    % of the cycles the vector memory unit was stalled
Each metric is averaged over the dispatches of a kernel, dispatches of the families bench.py times apart (tools/pmc_traffic.py FAMILY) weighted
equally.  Stamped like hbm_traffic.json (hash of the seed-stage sources + workload): bench.py prints roofline.counters = null otherwise."""
import collections
import csv
import json
import sys

import pmc_traffic


def main():
    out, batch, read_len, db_nt = sys.argv[1:5]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in sys.argv[5:]:
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smr::", "")
            fam = pmc_traffic.family_of(name)
            if fam is None or name.startswith("k_seed_search") or name.startswith("k_seed_cscan") or name.startswith("k_seed_colscan") or name.startswith("k_seed_emap") or name.startswith("k_seed_wbin"):
                continue                                        # (the families' minor kernels would dilute the averages)
            a = acc[fam][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    res = {"source": "rocprofv3 --pmc <derived metrics> --kernel-trace of bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run; mean over the dispatches of a kernel",
           "kernel_src_sha": pmc_traffic.kernel_src_sha(),
           "workload": {"batch_reads": int(batch), "read_len": int(read_len), "db_nt": int(db_nt)},
           "per_kernel": {fam: {c: v[0] / max(v[1], 1) for c, v in sorted(cs.items())} for fam, cs in sorted(acc.items())}}
    json.dump(res, open(out, "w"), indent=1)
    for fam, cs in res["per_kernel"].items():
        print("%-16s %s" % (fam, "  ".join("%s %.1f" % kv for kv in cs.items())))


if __name__ == "__main__":
    main()
