"""profiles/hbm_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected SEPARATELY, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) of `bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run`.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <batch_reads> <read_len> <db_nt> <out.json>

Per kernel: sum of the counter over its dispatches and the dispatch count.  Unit of FETCH_SIZE / WRITE_SIZE: KiB.  gfx950 correction of the
guide: FETCH_SIZE reports half of the bytes of wide coalesced reads -> the corrected figure doubles it (upper bound for the narrow accesses
of the trie walk); both are stored.  The seed stage of one launch = keys + scan + scatter + pg<0> + pg<1> + finish (+ the redo launches of
k_seed_search), summed per launch of k_seed_keys.  The file is stamped with a hash of the seed-stage kernel sources (SEED_SOURCES): bench.py only uses it when the
hash, the batch size, the read length and the DB size are the ones of its own run."""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


SEED_SOURCES = ["smr_seed.hpp", "smr_seed_pg.hpp", "smr_ibuild.hpp", "smr_trie_layout.hpp", "smr_host.hpp"]      # the kernels of the seed stage and the layouts they read


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


def main():
    fetch_csv, write_csv, batch, read_len, db_nt, out = sys.argv[1:7]
    f, fc = per_kernel(fetch_csv, "FETCH_SIZE")
    w, wc = per_kernel(write_csv, "WRITE_SIZE")
    kern = {}
    for k in sorted(set(f) | set(w)):
        kern[k] = {"calls": int(max(fc.get(k, 0), wc.get(k, 0))), "fetch_bytes": f.get(k, 0.0) * 1024, "write_bytes": w.get(k, 0.0) * 1024}
    launches = kern.get("k_seed_keys", {}).get("calls", 0)
    seed = [k for k in kern if k.startswith("k_seed") or k.startswith("k_scan")]
    fb = sum(kern[k]["fetch_bytes"] for k in seed)
    wb = sum(kern[k]["write_bytes"] for k in seed)
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --steps 1 --warmup 1 --resident-batches 2 --profile-run; KiB x 1024; "
                     "corrected = 2 x FETCH + WRITE (gfx950: FETCH_SIZE counts 64 B per 128-B request, MI355X_MICROARCH.md)",
           "kernel_src_sha": kernel_src_sha(),
           "workload": {"batch_reads": int(batch), "read_len": int(read_len), "db_nt": int(db_nt)},
           "seed_stage_launches": launches,
           "seed_stage_bytes_per_launch_raw": (fb + wb) / max(launches, 1),
           "seed_stage_bytes_per_launch": (2 * fb + wb) / max(launches, 1),
           "kernels": kern}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
    for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["fetch_bytes"])[:14]:
        print("%-28s calls %4d  fetch %8.3f GB  write %8.3f GB" % (k, v["calls"], v["fetch_bytes"] / 1e9, v["write_bytes"] / 1e9))


if __name__ == "__main__":
    main()
