"""GPU (`-m gpu`): BASELINE config 2 at FULL size -- the reference's bundled 100 000 amplicon reads against its bundled
silva-arc-16s-id95 DB (tests/golden/config2/, made by tests/golden/make_golden_config2.py from the unmodified reference binary).
Every per-read record (Read::toBinString bytes: classification, hit counts, scores, coordinates, CIGARs) must be the reference's:
compared through MD5 digests per 1000 reads.  ~50 % of the reads align, so this is also the amplicon-like traceback workload."""
import gzip
import hashlib
import json
import os
import time

import pytest

import sortmerna_amd as smr
from helpers import paths

pytestmark = pytest.mark.gpu
C2 = os.path.join(paths.GOLDEN, "config2")


@pytest.fixture(scope="module")
def setup(tmp_path_factory):
    g = json.load(open(os.path.join(C2, "config2.json")))
    tmp = tmp_path_factory.mktemp("c2")
    db = os.path.join(str(tmp), g["db"][:-3])
    with gzip.open(os.path.join(C2, g["db"]), "rb") as f, open(db, "wb") as o:
        o.write(f.read())
    e = smr.Engine(0)
    parts = smr.Index.build_gpu(e, db, 18, 3072.0, 10000)
    reads = smr.Reads.from_fastx_mt(os.path.join(C2, g["reads"]), 0)       # straight from the .gz
    assert reads.count == g["n_reads"]
    yield g, e, parts, reads
    e.close()


def digests(records, chunk):
    tot = hashlib.md5()
    chunks = []
    for c in range(0, len(records), chunk):
        h = hashlib.md5()
        for r in records[c:c + chunk]:
            b = len(r).to_bytes(4, "little") + r
            h.update(b)
            tot.update(b)
        chunks.append(h.hexdigest())
    return tot.hexdigest(), chunks


@pytest.mark.parametrize("seed_mode", [0, 1], ids=["pg", "dfs"])
@pytest.mark.parametrize("run", ["default", "num_alignments_0"])
def test_config2_fullsize_records_equal_the_reference(setup, run, seed_mode):
    g, e, parts, reads = setup
    if run == "num_alignments_0" and seed_mode == 1:
        pytest.skip("the all-alignments run is checked with the default seed kernel only")
    r = g["runs"][run]
    info = parts[0].info()
    # the reference's own Gumbel parameters (its log) -> the same minimal score through our Refstats arithmetic
    assert smr.minimal_score(r["lambda"], r["K"], info, reads.count, reads.total_len) == r["minimal_score"]
    e.set_seed_mode(seed_mode)
    p = smr.default_params(minimal_score=r["minimal_score"], **r["params"])
    e.prof_reset()
    t0 = time.time()
    smr.align(e, reads, [parts], [p], with_cigar=True, max_alignments_per_read=256 if r["params"] else None)
    dt = time.time() - t0
    recs = e.records()
    ctr = e.counters(1)
    e.set_seed_mode(0)
    assert ctr["num_aligned"] == r["num_aligned"] and sum(1 for x in recs if x) == r["n_records"]
    tot, chunks = digests(recs, g["chunk"])
    bad = [k for k, (a, b) in enumerate(zip(chunks, r["md5_chunks"])) if a != b]
    assert not bad, "%d of %d chunks of %d reads differ from the reference's records; first: reads %d.." % (len(bad), len(chunks), g["chunk"], bad[0] * g["chunk"])
    assert tot == r["md5_total"]
    pr = e.prof()
    print("config 2 (%s, seed kernel %d): %d reads, %d aligned, %.2f s incl. upload/fetch; seed %.1f ms, chain %.1f ms, traceback %.2f ms (%d launches)" % (
        run, seed_mode, reads.count, ctr["num_aligned"], dt, pr.seed_ms, pr.chain_ms, pr.trace_ms, pr.trace_launches))


def test_config2_read_set_twelve_times_over_at_the_shipped_thresholds(setup):
    """The skew paths at the thresholds they ship with (round 5's verdict, item 1c): the amplicon read set twelve times over in one batch (1.2 M reads, what
    `bench.py --workload config2` does 80 times) puts coarse key bins far above twice the average and 256 k tuples -- sorted by several blocks
    (k_seed_hbins_*) -- and makes most tuples repeats of another one's seed (k_seed_dedup / k_seed_prop).  Every copy of a read must get the record the
    reference gave the read (the per-1000-read digests of the golden run, copy by copy), and the last seed stage's sorted tuples must show the repeats."""
    g, e, parts, reads = setup
    r = g["runs"]["default"]
    seqs = []
    with gzip.open(os.path.join(C2, g["reads"]), "rt") as f:
        cur = []
        for line in f:
            if line.startswith(">"):
                if cur:
                    seqs.append("".join(cur))
                cur = []
            else:
                cur.append(line.strip())
        if cur:
            seqs.append("".join(cur))
    assert len(seqs) == g["n_reads"]
    copies = 12
    big = smr.Reads.from_seqs(seqs * copies)
    # the reference's threshold for the ORIGINAL run: the copies are aligned with it (a larger read total would raise the minimal score)
    p = smr.default_params(minimal_score=r["minimal_score"], **r["params"])
    e.set_seed_mode(0)
    smr.align(e, big, [parts], [p], with_cigar=True)
    recs = e.records()
    assert e.counters(1)["num_aligned"] == copies * r["num_aligned"]
    for k in range(copies):
        tot, chunks = digests(recs[k * g["n_reads"]:(k + 1) * g["n_reads"]], g["chunk"])
        assert tot == r["md5_total"], "copy %d of the read set: chunks %s differ" % (k, [i for i, (a, b) in enumerate(zip(chunks, r["md5_chunks"])) if a != b][:5])
    # the sorted tuples of one seed stage at this size: most are repeats (k_seed_dedup ran at its shipped threshold)
    e.upload_reads(big, 1)
    e.upload_index(parts[0], 0)
    e.reset_state()
    e.seed_scan(0, p, 0, 0)
    tuples, _, meta = e.seed_tuples()
    n_rep = int((tuples >> 63).sum())
    assert n_rep > meta["n"] // 2, (n_rep, meta["n"])
    e.unload_index(0)
    big.free()
