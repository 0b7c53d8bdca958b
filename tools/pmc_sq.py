"""profiles/sq_counters.json: the issue-side view of every kernel family from rocprofv3's derived metrics, collected over
`bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run` in passes of at most three metrics (no trace domain but --kernel-trace):

    python tools/pmc_sq.py <out.json> <batch_reads> <read_len> <db_nt> <counter_collection.csv> [<counter_collection.csv> ...]

  VALUBusy          % of the cycles a SIMD's vector ALU was issuing               (what north_star's "VALU issue fraction" asks for)
  SALUBusy          the same for the scalar ALU
  VALUUtilization   % of the 64 lanes active in the vector instructions issued
  LDSBankConflict   % of the LDS cycles lost to bank conflicts                      (north_star: "LDS-hit counters")
  MemUnitStalled    % of the cycles the vector memory unit was stalled
Each metric is averaged over the dispatches of a kernel family WEIGHTED BY EACH DISPATCH'S DURATION (round 5 weighted them equally: the round kernels'
dispatches last 3.5 us ... 12.7 ms, and the mean of k_sw16's `VALUBusy` came out at 22 % for a kernel whose arithmetic says it runs at its issue model).
The durations come from the Start_Timestamp / End_Timestamp columns of the counter file itself, or -- when it has none -- from a kernel_trace.csv of the
same run given among the files (joined on Dispatch_Id).  `--sw-cells N` (the DP cells of one profiled step, bench.py's C_SW_CELLS) with a pass that holds
SQ_INSTS_VALU adds `k_sw16.valu_per_cell_pair` = vector instructions per lane and pair of cells (the packed kernel's model: 13 with end cells, 10
without, + the systolic ramp).  Stamped like hbm_traffic.json (hash of the seed-stage sources + workload): bench.py prints roofline.counters = null otherwise."""
import collections
import csv
import json
import sys

import pmc_traffic


def main():
    args = sys.argv[1:]
    sw_cells = 0.0
    if "--sw-cells" in args:
        i = args.index("--sw-cells")
        sw_cells = float(args[i + 1])
        del args[i:i + 2]
    out, batch, read_len, db_nt = args[:4]
    files = args[4:]
    # durations per dispatch: from kernel_trace files (and from counter files that carry timestamps, below)
    dur = {}
    counters = []
    for path in files:
        rows = list(csv.DictReader(open(path)))
        if not rows:
            continue
        if "Counter_Name" in rows[0]:
            counters.append(rows)
        if "Start_Timestamp" in rows[0] and "End_Timestamp" in rows[0]:
            for r in rows:
                try:
                    dur[r["Dispatch_Id"]] = max(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), 1.0)
                except (KeyError, ValueError):
                    pass
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0.0, 0]))
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    unweighted = 0
    for rows in counters:
        for r in rows:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smr::", "")
            fam = pmc_traffic.family_of(name)
            if fam is None or name.startswith("k_seed_search") or name.startswith("k_seed_cscan") or name.startswith("k_seed_colscan") or name.startswith("k_seed_emap") or name.startswith("k_seed_wbin"):
                continue                                        # (the families' minor kernels would dilute the averages)
            w = dur.get(r["Dispatch_Id"])
            if w is None:
                w, unweighted = 1.0, unweighted + 1
            a = acc[fam][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]) * w
            a[1] += w
            a[2] += 1
            sums[name.split("<")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
    res = {"source": "rocprofv3 --pmc <derived metrics> --kernel-trace of bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run; mean over the dispatches of a kernel family, each weighted by its duration",
           "weighting": "duration" if not unweighted else "%d counter rows without a duration weighted 1" % unweighted,
           "kernel_src_sha": pmc_traffic.kernel_src_sha(),
           "workload": {"batch_reads": int(batch), "read_len": int(read_len), "db_nt": int(db_nt)},
           "per_kernel": {fam: {c: v[0] / max(v[1], 1e-9) for c, v in sorted(cs.items()) if not c.startswith("SQ_")} for fam, cs in sorted(acc.items())}}
    res["per_kernel"] = {k: v for k, v in res["per_kernel"].items() if v}
    if sw_cells and sums.get("k_sw16", {}).get("SQ_INSTS_VALU"):
        # SQ_INSTS_VALU counts wave instructions: x 64 lanes, / (cells / 2) pairs of cells
        res["k_sw16"] = {"sq_insts_valu": sums["k_sw16"]["SQ_INSTS_VALU"], "dp_cells": sw_cells,
                         "valu_per_cell_pair": sums["k_sw16"]["SQ_INSTS_VALU"] * 64.0 / (sw_cells / 2.0),
                         "note": "vector instructions per lane and packed pair of cells, every lane and step of the quad systolic arrays counted (ramp, padding rows, hand-over, reference fetch): "
                                 "the inner recurrence is 13 with end cells / 10 score-only; DP cells = the sequential walk's ssw_align calls (C_SW_CELLS) + what was scored ahead and not used"}
    json.dump(res, open(out, "w"), indent=1)
    print("weighting: %s" % res["weighting"])
    for fam, cs in res["per_kernel"].items():
        print("%-16s %s" % (fam, "  ".join("%s %.1f" % kv for kv in cs.items())))
    if "k_sw16" in res:
        print("k_sw16: %.3g vector wave-instructions for %.3g DP cells = %.1f per lane and cell pair" % (res["k_sw16"]["sq_insts_valu"], sw_cells, res["k_sw16"]["valu_per_cell_pair"]))


if __name__ == "__main__":
    main()
