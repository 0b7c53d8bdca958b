"""profiles/hbm_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected SEPARATELY, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) of `bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run`.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <batch_reads> <read_len> <db_nt> <out.json>

Per kernel: sum of the counter over its dispatches and the dispatch count.  Unit of FETCH_SIZE / WRITE_SIZE: KiB.
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE, CALIBRATED on this hardware with known-byte kernels of our own access patterns
(tools/microbench/mb.hip `calib`, profiles/r03a_pmc_calibration_*): every read request that leaves the L2 is 128 bytes (TCC_EA0_RDREQ = bytes / 128
for streaming reads of 4 / 12 / 16 bytes per lane, and exactly one request per random 4 / 8 / 16-byte gather, independent or dependent) and
FETCH_SIZE tallies it as 64 -> x 2 for every read pattern; WRITE_SIZE counts 64-byte write requests and equals the bytes written for
coalesced stores of 4 / 12 / 16 bytes per lane (x 1; per-lane scattered 12-byte stores show up as what they cost: 1.2 - 3.3 x their bytes).
Kernels are grouped into the families bench.py times apart (smr_prof_kernels); per_launch_bytes[family] = that sum / launches of k_seed_keys.
The file is stamped with a hash of the seed-stage kernel sources (SEED_SOURCES): bench.py only uses it when the hash, the batch size, the read
length and the DB size are the ones of its own run."""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


SEED_SOURCES = ["smr_seed.hpp", "smr_seed_pg.hpp", "smr_ibuild.hpp", "smr_trie_layout.hpp", "smr_host.hpp"]      # the kernels of the seed stage and the layouts they read
FAMILY = [("k_seed_keys", "k_seed_keys"), ("k_seed_emap", "k_seed_keys"), ("k_seed_wbin", "k_seed_split"), ("k_seed_cscan", "k_seed_split"), ("k_seed_colscan", "k_seed_split"), ("k_seed_split", "k_seed_split"),
          ("k_seed_bins", "k_seed_bins"), ("k_seed_hbins", "k_seed_bins"), ("k_seed_dedup", "k_seed_bins"), ("k_seed_active", "k_seed_bins"),      # (round 6: the skew paths, timed with the second sort pass)
          ("k_seed_prop<0>", "k_seed_pg<0>"), ("k_seed_prop<1>", "k_seed_pg<1>"), ("k_seed_pg<0>", "k_seed_pg<0>"), ("k_seed_search<0>", "k_seed_pg<0>"), ("k_seed_pg<1>", "k_seed_pg<1>"),
          ("k_seed_search<1>", "k_seed_pg<1>"), ("k_seed_finish", "k_seed_finish"), ("k_cand", "k_cand"), ("k_chain", "k_chain"), ("k_begins", "k_begins"),
          ("k_trace", "k_trace"), ("k_walk", "k_walk"), ("k_sw16", "k_sw16"), ("k_wnext", "k_wnext"), ("k_wlist", "k_wnext")]      # (round 5: the candidate walk in rounds, smr_walk.hpp)


def kernel_src_sha():
    h = hashlib.sha1()
    d = os.path.join(ROOT, "sortmerna_amd", "csrc")
    for f in SEED_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def per_kernel(path, counter):
    tot = collections.defaultdict(float)
    calls = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("smr::", "")
        tot[name] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], name) not in seen:
            seen.add((r["Dispatch_Id"], name))
            calls[name] += 1
    return tot, calls


def family_of(name):
    for pre, fam in FAMILY:
        if name.startswith(pre):
            return fam
    return None


def main():
    fetch_csv, write_csv, batch, read_len, db_nt, out = sys.argv[1:7]
    f, fc = per_kernel(fetch_csv, "FETCH_SIZE")
    w, wc = per_kernel(write_csv, "WRITE_SIZE")
    kern = {}
    for k in sorted(set(f) | set(w)):
        kern[k] = {"calls": int(max(fc.get(k, 0), wc.get(k, 0))), "fetch_bytes": f.get(k, 0.0) * 1024, "write_bytes": w.get(k, 0.0) * 1024}
    launches = sum(v["calls"] for k, v in kern.items() if k.startswith("k_seed_keys"))       # (the kernel is a template: k_seed_keys<...>)
    fam = collections.defaultdict(lambda: [0.0, 0.0])
    for k, v in kern.items():
        fm = family_of(k)
        if fm:
            fam[fm][0] += v["fetch_bytes"]
            fam[fm][1] += v["write_bytes"]
    per_launch = {k: (2 * v[0] + v[1]) / max(launches, 1) for k, v in fam.items()}
    seed = [k for k in per_launch if k.startswith("k_seed")]
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run; KiB x 1024",
           "note": "rocprofv3 PMC, HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE per launch of the kernel family (factors calibrated on known-byte kernels with the same access "
                   "patterns: profiles/r03a_pmc_calibration_*), kernel sources %s" % kernel_src_sha(),
           "kernel_src_sha": kernel_src_sha(),
           "workload": {"batch_reads": int(batch), "read_len": int(read_len), "db_nt": int(db_nt)},
           "seed_stage_launches": launches,
           "per_launch_bytes": per_launch,
           "per_launch_fetch_raw": {k: v[0] / max(launches, 1) for k, v in fam.items()},
           "per_launch_write_raw": {k: v[1] / max(launches, 1) for k, v in fam.items()},
           "seed_stage_bytes_per_launch": sum(per_launch[k] for k in seed),
           "kernels": kern}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
    for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["fetch_bytes"])[:16]:
        print("%-28s calls %4d  fetch(raw) %8.3f GB  write %8.3f GB" % (k, v["calls"], v["fetch_bytes"] / 1e9, v["write_bytes"] / 1e9))


if __name__ == "__main__":
    main()
