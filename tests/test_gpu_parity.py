"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI of libsmr_hip.so,
against the CPU oracle on the same seeded inputs.  Bar: byte-exact Read::toBinString records (classification,
hit counts, SW scores, coordinates, CIGARs) and identical Readstats counters."""
import ctypes as C
import os

import numpy as np
import pytest

import sortmerna_amd as smr
from helpers import orc, refrun
from helpers.workload import Workload, iseq_for_strand, GUMBEL_UNIFORM

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[0, 1], ids=["pg", "dfs"])
def engine(request):
    """every test runs with both seed-search kernels: the pigeonhole kernel (default) and the per-lane DFS kernel"""
    e = smr.Engine(0)      # raises without a GPU / without the HIP library: no CPU fallback
    e.set_seed_mode(request.param)
    yield e
    e.close()


@pytest.fixture(scope="module")
def wl(tmp_path_factory):
    return Workload(str(tmp_path_factory.mktemp("wl")))


def _compare(recs_gpu, recs_orc, what):
    bad = [i for i, (a, b) in enumerate(zip(recs_gpu, recs_orc)) if a != b]
    msg = ""
    if bad:
        i = bad[0]
        msg = "%s: %d/%d records differ; first read %d\n gpu=%s\n orc=%s" % (
            what, len(bad), len(recs_orc), i, refrun.parse_record(recs_gpu[i]), refrun.parse_record(recs_orc[i]))
    assert not bad, msg


def test_seed_scan_matches_oracle(engine, wl):
    """k_seed (window scan + burst-trie descent) vs traversetrie_align restated on the CPU, per strand and pass."""
    L = orc.lib()
    ix = L.orc_index_load(wl.prefix.encode(), 0, 18)
    engine.upload_reads(wl.reads, 1)
    engine.upload_index(wl.parts[0], 0)
    p = smr.default_params(minimal_score=wl.minimal_score)
    strides = [18, 9, 3]
    ids = (C.c_uint32 * 4096)()
    for strand in (0, 1):
        iseqs = [iseq_for_strand(s, strand) for s in wl.seqs]
        for pass_ in (0, 1, 2):
            engine.reset_state()
            n = engine.seed_scan(0, p, strand, pass_)
            got = engine.seed_hits()
            assert len(got) == n
            got_set = set(map(tuple, got.tolist()))
            assert len(got_set) == len(got), "duplicate (read,id,win) triples"
            exp = set()
            for r, v in enumerate(iseqs):
                if len(v) < 18:
                    continue
                numwin = (len(v) - 18 + strides[pass_]) // strides[pass_]
                for k in range(numwin):
                    w = k * strides[pass_]
                    if any(w % strides[q] == 0 for q in range(pass_)):
                        continue
                    z = C.c_int()
                    c = L.orc_window_hits(ix, v.ctypes.data, w, 18, 0, 0, ids, 4096, C.byref(z))
                    for q in range(c):
                        exp.add((r, ids[q], w))
            assert got_set == exp, "strand %d pass %d: %d gpu hits vs %d oracle hits, %d differ" % (
                strand, pass_, len(got_set), len(exp), len(got_set ^ exp))
    L.orc_index_free(ix)
    engine.unload_index(0)


@pytest.mark.parametrize("opts", [
    {},
    {"num_alignments": 0},
    {"num_alignments": 3},
    {"is_best": 0, "num_alignments": 2},
    {"is_reverse": 0},
    {"is_forward": 0},
    {"is_full_search": 1},
    {"num_seeds": 3, "edges": 10},
    # the rest of the option space of the path (alignment.cpp:134-169,251-261, options.cpp:1566-1758, read.cpp:274-288)
    {"num_seeds": 1},                                # every reference with ONE seed hit is a candidate (k_cand only marks the read)
    {"min_lis": 1},
    {"min_lis": 3, "num_alignments": 2},
    {"score_N": 0},                                  # -N
    {"gap_open": 3, "gap_ext": 2},
    {"match": 3, "mismatch": -4, "gap_open": 6, "gap_ext": 3, "score_N": -1},
    {"minoccur": 2},
    {"minimal_score_delta": 30},                     # a stricter -e
], ids=["default", "all", "best3", "nobest2", "F", "R", "full_search", "seeds3_edges10", "seeds1", "min_lis1", "min_lis3_best2", "N0", "gaps_3_2", "score_3_4_6_3", "minoccur2", "stricter_e"])
def test_align_records_match_oracle(engine, wl, opts):
    opts = dict(opts)
    if "minimal_score_delta" in opts:
        opts["minimal_score"] = wl.minimal_score + opts.pop("minimal_score_delta")
    recs_o, ctr_o = wl.oracle_records(**opts)
    recs_g, ctr_g = wl.gpu_records(engine, **opts)
    _compare(recs_g, recs_o, str(opts))
    assert ctr_g["num_aligned"] == ctr_o["num_aligned"]
    assert ctr_g["num_short"] == ctr_o["num_short"]
    assert ctr_g["reads_matched_per_db"][0] == ctr_o["per_db"]
    assert ctr_o["num_aligned"] > 100      # the workload really aligns reads


def test_multi_part_index(engine, tmp_path):
    """hit counts / records with the index split into several parts (state carried across parts like the KVDB)."""
    w = Workload(str(tmp_path), db_nt=400_000, n_reads=1500, seed=21, max_mb=1.0)
    assert w.stats.nparts >= 3
    recs_o, ctr_o = w.oracle_records()
    recs_g, ctr_g = w.gpu_records(engine)
    _compare(recs_g, recs_o, "multi-part")
    assert ctr_g["num_aligned"] == ctr_o["num_aligned"]


def test_longer_reads(engine, tmp_path):
    w = Workload(str(tmp_path), db_nt=200_000, n_reads=600, read_len=301, seed=33, frac_db=0.6)
    recs_o, ctr_o = w.oracle_records()
    recs_g, ctr_g = w.gpu_records(engine)
    _compare(recs_g, recs_o, "301 nt reads")
    assert ctr_g["num_aligned"] == ctr_o["num_aligned"] > 50


def test_reads_sharing_seeds_with_thousands_of_references(engine, tmp_path):
    """a DB of 1 500 near-identical sequences (one family, 0.5-2 % divergence): every read from it shares seeds with up to all of them, far
    beyond the 384 members the LDS candidate-set table of a wave holds -- such reads build their set in the block's global table and get
    their tuples grouped by member (round 1: SMR_ERR_CAPACITY beyond 3 072 references)"""
    w = Workload(str(tmp_path), db_nt=600_000, n_reads=160, read_len=150, frac_db=0.7, seed=123, family_size=1500, mean_len=400,
                 db_kw=dict(sub_lo=0.005, sub_hi=0.02))
    assert w.parts[0].info().numseq >= 1400
    recs_o, ctr_o = w.oracle_records()
    recs_g, ctr_g = w.gpu_records(engine)
    _compare(recs_g, recs_o, "1500-member family")
    assert ctr_g["num_aligned"] == ctr_o["num_aligned"] > 80
    recs_o, ctr_o = w.oracle_records(num_alignments=5)
    recs_g, ctr_g = w.gpu_records(engine, num_alignments=5)
    _compare(recs_g, recs_o, "1500-member family, best 5")


def test_mixed_read_lengths(engine, tmp_path):
    """60-400 nt reads in one batch: reads up to 256 nt are parked / scored four per wave, longer ones take the single-problem kernels
    (and force the parked tasks to be scored first, because their strip boundaries share LDS with the parked windows); k_begins runs
    in its one-problem-per-wave mode because the batch's longest read exceeds the four-problem kernel"""
    import numpy as np
    w = Workload(str(tmp_path), db_nt=250_000, n_reads=1800, read_len=400, seed=91, frac_db=0.5)
    rng = np.random.Generator(np.random.PCG64(17))
    w.seqs = [s[: int(rng.integers(60, 401))] if len(s) > 60 else s for s in w.seqs]
    w.reads = smr.Reads.from_seqs(w.seqs)
    w.minimal_score = smr.minimal_score(0.618874, 0.343238, w.parts[0].info(), len(w.seqs), sum(map(len, w.seqs)))
    recs_o, ctr_o = w.oracle_records()
    recs_g, ctr_g = w.gpu_records(engine)
    _compare(recs_g, recs_o, "mixed read lengths")
    assert ctr_g["num_aligned"] == ctr_o["num_aligned"] > 300
    lens = [len(s) for s, r in zip(w.seqs, recs_o) if r]
    assert min(lens) < 200 and max(lens) > 300


@pytest.mark.parametrize("lnwin", [10, 12, 14, 16])
def test_other_seed_lengths(engine, tmp_path, lnwin):
    """-L 10/12/14/16: other window lengths (partialwin 5..8), their trie depth limits, automaton tail tables and directory widths"""
    w = Workload(str(tmp_path), db_nt=120_000, n_reads=1200, seed=50 + lnwin, frac_db=0.5, lnwin=lnwin)
    recs_o, ctr_o = w.oracle_records()
    recs_g, ctr_g = w.gpu_records(engine)
    _compare(recs_g, recs_o, "L=%d" % lnwin)
    assert ctr_g["num_aligned"] == ctr_o["num_aligned"] > 100


def test_non_default_strides(engine, wl):
    """-passes 18,6,2 (the reference's own parser of that option is broken, options.cpp:704-732, so this is oracle-only)"""
    recs_o, ctr_o = wl.oracle_records(skiplengths=[18, 6, 2])
    recs_g, ctr_g = wl.gpu_records(engine, skiplengths=[18, 6, 2])
    _compare(recs_g, recs_o, "passes 18,6,2")
    assert ctr_g["num_aligned"] == ctr_o["num_aligned"]


def test_long_noisy_reads(engine, tmp_path):
    """PacBio-like reads (0.8-3 kb, ~10 % errors incl. indels): several SW strips, wide traceback bands, many seeds."""
    import numpy as np
    from sortmerna_amd import synth
    from helpers import orc
    w = Workload(str(tmp_path), db_nt=150_000, n_reads=20, seed=41, family_size=6)
    codes, offs = synth.load_db_codes(w.db)
    rng = np.random.Generator(np.random.PCG64(99))
    seqs = []
    for i in range(48):
        sq = int(rng.integers(0, len(offs) - 1))
        ln = int(min(offs[sq + 1] - offs[sq], rng.integers(800, 3000)))
        st = int(offs[sq] + rng.integers(0, offs[sq + 1] - offs[sq] - ln + 1))
        out = []
        for c in codes[st:st + ln]:
            u = rng.random()
            if u < 0.03:
                continue                                   # deletion
            if u < 0.06:
                out.append(int(rng.integers(0, 4)))        # insertion before the base
            out.append(int((c + rng.integers(1, 4)) & 3) if u > 0.96 else int(c))
        s = "".join("ACGT"[c] for c in out)
        if i % 2:
            s = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
        seqs.append(s)
    seqs += ["".join("ACGT"[c] for c in rng.integers(0, 4, size=2000)) for _ in range(6)]     # background
    w.seqs = seqs
    w.reads = smr.Reads.from_seqs(seqs)
    w.minimal_score = smr.minimal_score(0.618874, 0.343238, w.parts[0].info(), len(seqs), sum(map(len, seqs)))
    recs_o, ctr_o = w.oracle_records()
    recs_g, ctr_g = w.gpu_records(engine)
    _compare(recs_g, recs_o, "long noisy reads")
    assert ctr_g["num_aligned"] == ctr_o["num_aligned"] >= 40
    # the same reads under a scheme that goes through the striped slow path (smr_sw_striped.hpp: the LONG instantiations, scratch rows in global memory)
    recs_o, ctr_o = w.oracle_records(gap_open=3, gap_ext=3)
    recs_g, ctr_g = w.gpu_records(engine, gap_open=3, gap_ext=3)
    _compare(recs_g, recs_o, "long noisy reads, gap_open = gap_ext")
    assert ctr_g["num_aligned"] == ctr_o["num_aligned"] >= 40


def test_percent_edges_on_kilobase_reads(engine, tmp_path):
    """`-edges 8%` on 1-2 kb reads: the SW window grows by a share of the read length on both sides (alignment.cpp:283-286), so reference
    spans exceed max_len + a constant -- the LDS windows of k_chain / k_begins and the scratch of the traceback are sized from the same
    percentage (round 1 sized the traceback from the absolute value and could fail on such runs)"""
    import numpy as np
    from sortmerna_amd import synth
    w = Workload(str(tmp_path), db_nt=200_000, n_reads=20, seed=77, family_size=4, mean_len=4000)
    codes, offs = synth.load_db_codes(w.db)
    rng = np.random.Generator(np.random.PCG64(5))
    seqs = []
    for i in range(24):
        sq = int(rng.integers(0, len(offs) - 1))
        full = int(offs[sq + 1] - offs[sq])
        ln = int(min(full - 400, rng.integers(1000, 2000)))
        st = int(offs[sq] + 200 + rng.integers(0, full - ln - 400 + 1))
        out = []
        for c in codes[st:st + ln]:
            u = rng.random()
            if u < 0.02:
                continue
            if u < 0.04:
                out.append(int(rng.integers(0, 4)))
            out.append(int((c + rng.integers(1, 4)) & 3) if u > 0.97 else int(c))
        # soft-clipped ends: random letters around the homologous part, so that the window really extends into the edges
        s = "".join("ACGT"[c] for c in list(rng.integers(0, 4, size=30)) + out + list(rng.integers(0, 4, size=30)))
        if i % 2:
            s = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
        seqs.append(s)
    w.seqs = seqs
    w.reads = smr.Reads.from_seqs(seqs)
    w.minimal_score = smr.minimal_score(0.618874, 0.343238, w.parts[0].info(), len(seqs), sum(map(len, seqs)))
    opts = {"edges": 8, "is_as_percent": 1}
    recs_o, ctr_o = w.oracle_records(**opts)
    recs_g, ctr_g = w.gpu_records(engine, **opts)
    _compare(recs_g, recs_o, "-edges 8%")
    assert ctr_g["num_aligned"] == ctr_o["num_aligned"] >= 20


@pytest.mark.parametrize("scoring", [{}, {"match": 5, "mismatch": -4, "score_N": -4}], ids=["2_-3", "5_-4"])
def test_long_reads_with_large_gaps(engine, tmp_path, scoring):
    """3-5 kb reads with one 120-400 nt deletion or insertion: 16-bit SW range, >= 12 SW strips, traceback bands of
    hundreds of columns (the second/third scratch level of smr_traceback); with match 5 the scores exceed 16383, which takes
    the SW kernel's unpacked running-maximum path"""
    import numpy as np
    from sortmerna_amd import synth
    w = Workload(str(tmp_path), db_nt=200_000, n_reads=20, seed=61, family_size=3, mean_len=6000)
    codes, offs = synth.load_db_codes(w.db)
    rng = np.random.Generator(np.random.PCG64(7))
    seqs = []
    for i in range(10):
        sq = int(rng.integers(0, len(offs) - 1))
        ln = int(min(offs[sq + 1] - offs[sq], rng.integers(3000, 5000)))
        st = int(offs[sq] + rng.integers(0, offs[sq + 1] - offs[sq] - ln + 1))
        s = codes[st:st + ln].copy()
        m = rng.random(len(s)) < 0.02
        s[m] = (s[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
        cut = int(rng.integers(1000, ln - 1000))
        gap = int(rng.integers(120, 400))
        if i % 2:
            s = np.concatenate([s[:cut], s[cut + gap:]])                                   # deletion in the read
        else:
            s = np.concatenate([s[:cut], rng.integers(0, 4, size=gap, dtype=np.uint8), s[cut:]])   # insertion
        t = "".join("ACGT"[c] for c in s)
        if i % 3 == 0:
            t = t[::-1].translate(str.maketrans("ACGT", "TGCA"))
        seqs.append(t)
    w.seqs = seqs
    w.reads = smr.Reads.from_seqs(seqs)
    w.minimal_score = smr.minimal_score(0.618874, 0.343238, w.parts[0].info(), len(seqs), sum(map(len, seqs)))
    recs_o, ctr_o = w.oracle_records(**scoring)
    recs_g, ctr_g = w.gpu_records(engine, **scoring)
    _compare(recs_g, recs_o, "long reads with large gaps")
    if scoring:
        assert max(refrun.parse_record(r)["alignv"][0]["score1"] for r in recs_o if r) > 16383
    assert ctr_g["num_aligned"] == ctr_o["num_aligned"] >= 8
    spans = [refrun.parse_record(r)["alignv"][0] for r in recs_o if r]
    assert max(a["read_end1"] - a["read_begin1"] for a in spans) > 2500          # alignments really span the gap


@pytest.mark.parametrize("lnwin,db_nt,family", [(18, 300_000, 40), (18, 3_000_000, 40), (14, 3_000_000, 8), (12, 300_000, 40)],
                         ids=["L18-scan-blocks", "L18-partial-directories", "L14-full-directories", "L12-full-directories"])
def test_pigeonhole_seed_kernel_equals_the_dfs_kernel(tmp_path, lnwin, db_nt, family):
    """k_seed_pg (exact-key directories of the pigeonhole layout, candidates applied by stored DFS rank) against k_seed_search (the
    reference's DFS with the table automaton) on index shapes that exercise every kind of block: the (read, id, window) hit triples of
    every strand and pass must be the same set, and the fast kernel must really have produced them (no wave handed to the DFS kernel)."""
    import numpy as np
    w = Workload(str(tmp_path), db_nt=db_nt, n_reads=1500, seed=lnwin + db_nt % 97, frac_db=0.5, lnwin=lnwin, family_size=family)
    p = smr.default_params(minimal_score=w.minimal_score)
    p.lnwin = lnwin
    p.skiplengths[0], p.skiplengths[1], p.skiplengths[2] = lnwin, lnwin // 2, 3
    e = smr.Engine(0)
    try:
        e.upload_reads(w.reads, 1)
        e.upload_index(w.parts[0], 0)
        res = {}
        for mode in (0, 1):
            e.set_seed_mode(mode)
            e.prof_reset()
            out = []
            for strand in (0, 1):
                for pass_ in (0, 1, 2):
                    e.reset_state()
                    n = e.seed_scan(0, p, strand, pass_)
                    got = e.seed_hits()
                    assert len(got) == n
                    out.append(got[np.lexsort((got[:, 2], got[:, 1], got[:, 0]))])
            pr = e.prof()
            res[mode] = (out, pr.n_seed_redo, pr.n_entry)
        for a, b in zip(res[0][0], res[1][0]):
            assert np.array_equal(a, b)
        assert sum(len(x) for x in res[0][0]) > 10000
        assert res[0][1] == 0 and res[1][1] == 0, "waves of the fast kernel were searched again by the DFS kernel"
        if lnwin < 18 or db_nt > 1_000_000:
            assert res[0][2] < res[1][2], "the directories did not narrow the search"
    finally:
        e.close()

@pytest.mark.parametrize("env", [("SMR_HANDOVER", "0"), ("SMR_CAND_BLOOM", "64")], ids=lambda e: "%s=%s" % e)
def test_optional_paths_of_the_candidate_stage_give_the_oracle_records(wl, monkeypatch, env):
    """k_chain gathering a marked read's positions itself instead of taking k_cand's record; a 2 Kbit Bloom bitmap in k_cand (more reads
    marked by false collisions).  Neither may change a record (the emulator runs the same cases: tests/test_emu_kernels.py)."""
    monkeypatch.setenv(*env)
    e = smr.Engine(0)
    try:
        recs_o, ctr_o = wl.oracle_records()
        recs_g, ctr_g = wl.gpu_records(e)
        _compare(recs_g, recs_o, "%s=%s" % env)
        assert ctr_g["num_aligned"] == ctr_o["num_aligned"]
    finally:
        e.close()


WALK_VARIANTS = [{"SMR_WALK_SPLIT": "0"},                                   # k_chain walks every marked read (round 4's path)
                 {"SMR_WALK_ROUNDS": "1"},                                  # the last round is the first: every task scored inside k_walk<true>
                 {"SMR_WALK_ROUNDS": "2", "SMR_WALK_K": "1"},               # one task per read and round, the rest in the last round
                 {"SMR_WALK_ROUNDS": "3", "SMR_WALK_K": "8", "SMR_WALK_ASSUME": "0"},     # every look-ahead predicts "aligns": all tasks scored with end cells
                 {"SMR_WALK_ROUNDS": "12", "SMR_WALK_K": "2", "SMR_WALK_ASSUME": "100"}]  # round 0 predicts "does not align": accepted alignments stored end-pending (k_begins finds both cells)


@pytest.mark.parametrize("env", WALK_VARIANTS, ids=lambda e: ",".join("%s=%s" % (k[9:], v) for k, v in e.items()))
@pytest.mark.parametrize("opts", [{}, {"num_alignments": 0}, {"is_best": 0, "num_alignments": 2}, {"num_seeds": 1}, {"min_lis": 3, "num_alignments": 2}],
                         ids=["default", "all", "nobest2", "seeds1", "min_lis3_best2"])
def test_candidate_walk_in_rounds_gives_the_oracle_records(wl, monkeypatch, env, opts):
    """smr_walk.hpp: the candidate walk as rounds of k_walk -> k_sw16 -> k_wnext.  How many rounds there are, how many tasks a read leaves per
    round and what its look-ahead predicts decide which Smith-Waterman problems are scored when and by which kernel -- never a record, a
    counter or the number of ssw_align calls of the sequential walk (the emulator runs the same cases: tests/test_emu_kernels.py)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    e = smr.Engine(0)
    try:
        recs_o, ctr_o = wl.oracle_records(**opts)
        recs_g, ctr_g = wl.gpu_records(e, **opts)
        _compare(recs_g, recs_o, "%s %s" % (env, opts))
        assert ctr_g["num_aligned"] == ctr_o["num_aligned"] and ctr_g["reads_matched_per_db"][0] == ctr_o["per_db"]
        assert e.prof().n_sw_fwd == ctr_o["n_sw_fwd"]
        launched = {k: v["launches"] for k, v in e.prof_kernels().items()}
        assert (launched.get("k_walk", 0) > 0) == (env.get("SMR_WALK_SPLIT") != "0")
    finally:
        e.close()


def test_rounds_adapt_from_part_to_part_without_changing_a_record(wl, monkeypatch):
    """Without SMR_WALK_ROUNDS a context starts with eight rounds per (strand, pass) and lowers the number part by part towards what the previous
    part needed (its last round with more than a few reads + the closing round): every run over the same reads gives the oracle's records."""
    monkeypatch.delenv("SMR_WALK_ROUNDS", raising=False)
    e = smr.Engine(0)
    try:
        assert e.walk_rounds() == [8, 8, 8]
        recs_o, ctr_o = wl.oracle_records()
        seen = [e.walk_rounds()]
        for _ in range(5):
            recs_g, ctr_g = wl.gpu_records(e)
            _compare(recs_g, recs_o, "rounds %s" % seen[-1])
            assert ctr_g["num_aligned"] == ctr_o["num_aligned"]
            seen.append(e.walk_rounds())
        assert all(a >= b for x, y in zip(seen, seen[1:]) for a, b in zip(x, y)) and max(seen[-1]) <= 4 and min(seen[-1]) >= 2, seen
    finally:
        e.close()


@pytest.mark.parametrize("scoring", [{"gap_open": 3, "gap_ext": 3}, {"gap_open": 2, "gap_ext": 2}, {"mismatch": -5, "gap_open": 2, "gap_ext": 1}, {"score_N": 1},
                                     {"gap_open": 3, "gap_ext": 3, "num_alignments": 3}],
                         ids=["open_equals_ext", "open_equals_ext_2", "gaps_below_half_a_mismatch", "positive_N", "open_equals_ext_best3"])
def test_schemes_under_which_ssw_c_leaves_the_affine_recurrence_give_the_oracle_records(engine, wl, scoring):
    """With gap_open <= gap_ext the reference's 16-bit kernel loses gaps across stripe boundaries (tests/test_oracle_golden.py shows it on ssw.c itself),
    with gaps cheaper than half a mismatch both kernels miss adjacent gaps, and a positive score for N makes padding visible: under such schemes the
    reference's answer depends on the SIMD stripe geometry.  Rounds 1 - 5 refused them; round 6 aligns them through the slow path that reproduces that
    geometry (smr_sw_striped.hpp, selected by the scheme) -- records and counters equal the oracle's, whose ssw port is pinned to the reference binary
    over random schemes by tools/fuzz_ref.py.  (The reference's parser takes gap_ext <= gap_open only: options.cpp:1637.)"""
    recs_o, ctr_o = wl.oracle_records(**scoring)
    recs_g, ctr_g = wl.gpu_records(engine, **scoring)
    _compare(recs_g, recs_o, "stripe-sensitive scheme %s" % scoring)
    assert ctr_g["num_aligned"] == ctr_o["num_aligned"] > 100
    sc = dict(match=2, mismatch=-3, score_N=-3, gap_open=5, gap_ext=2)
    sc.update({k: v for k, v in scoring.items() if k != "num_alignments"})
    with pytest.raises(smr.SmrError, match="supported range"):       # the FAST kernels at the ssw.h seam still say that such a scheme is not theirs
        engine.ssw_batch([b"\x00\x01\x02\x03"], [b"\x00\x01\x02\x03\x00"], filters=0, mode=0, **sc)
    recs_o, _ = wl.oracle_records(gap_open=3, gap_ext=2)           # (and the context goes back to the fast kernels)
    recs_g, _ = wl.gpu_records(engine, gap_open=3, gap_ext=2)
    _compare(recs_g, recs_o, "after a stripe-sensitive scheme")


@pytest.mark.parametrize("opts,msg", [({"edges": 0}, "edges must be 1..10"), ({"edges": 11}, "edges must be 1..10"), ({"edges": 1, "is_as_percent": 1}, "rounds to 0 letters")],
                         ids=["edges0", "edges11", "one_percent_of_a_short_read"])
def test_edges_outside_what_the_reference_defines_are_refused(engine, wl, opts, msg):
    """--edges: the reference's parser takes 1..10 (options.cpp:676).  With 0 -- also what N % of a read of fewer than 100 / N letters rounds to -- its
    `tail > edges - 1` is an unsigned compare (alignment.cpp:320,345) and the read is aligned against the whole rest of the reference sequence; larger
    margins make window lengths wrap (found by tools/fuzz_emu.py).  Said before anything runs."""
    with pytest.raises(smr.SmrError, match=msg):
        wl.gpu_records(engine, **opts)
    recs_o, _ = wl.oracle_records(edges=10, is_as_percent=1)        # 10 % of the shortest searchable read (18 letters) is a letter: fine
    recs_g, _ = wl.gpu_records(engine, edges=10, is_as_percent=1)
    _compare(recs_g, recs_o, "edges 10 %")


def test_small_candidate_pool_is_redone_and_grows(wl, monkeypatch):
    """SMR_PG_CAND_CAP=8: most waves of k_seed_pg overflow their candidate pool and are searched again by the DFS kernel -- the records
    stay the oracle's -- and smr_align_part doubles the pool for the next part, so a second run over the same reads is redone less."""
    monkeypatch.setenv("SMR_PG_CAND_CAP", "8")
    e = smr.Engine(0)
    try:
        recs_o, _ = wl.oracle_records()
        e.prof_reset()
        recs_g, _ = wl.gpu_records(e)
        _compare(recs_g, recs_o, "small candidate pool")
        first = e.prof().n_seed_redo
        assert first > 0
        e.prof_reset()
        for _ in range(4):
            wl.gpu_records(e)
        e.prof_reset()
        recs_g, _ = wl.gpu_records(e)
        _compare(recs_g, recs_o, "grown candidate pool")
        assert e.prof().n_seed_redo < first
    finally:
        e.close()


def test_batch_dominated_by_one_sequence(engine, tmp_path):
    """3 000 copies of two reads among 500 others (a sample dominated by one organism's rRNA): all their windows share a few keys, so
    the tuple sort sees bins of thousands (one block of k_seed_bins handles a whole coarse bin) and whole waves of k_seed_pg search the
    same mini-trie with the same pattern"""
    w = Workload(str(tmp_path), db_nt=200_000, n_reads=3500, seed=77, frac_db=0.6)
    hot = [w.seqs[20], w.seqs[41]]
    for i in range(500, 3500):
        w.seqs[i] = hot[i & 1]
    w.reads = smr.Reads.from_seqs(w.seqs)
    lam, K = GUMBEL_UNIFORM
    w.minimal_score = smr.minimal_score(lam, K, w.parts[0].info(), len(w.seqs), sum(map(len, w.seqs)))
    recs_o, ctr_o = w.oracle_records()
    recs_g, ctr_g = w.gpu_records(engine)
    _compare(recs_g, recs_o, "hot reads")
    assert ctr_g["num_aligned"] == ctr_o["num_aligned"]
    assert recs_g[500] == recs_g[502] and recs_g[501] == recs_g[503]


SKEW_VARIANTS = [{"SMR_SEED_HOT_BIN": "64", "SMR_SEED_HOT_SUB": "200", "SMR_SEED_DEDUP": "2"},      # large coarse bins in sub-ranges of 200 tuples + every repeated seed searched once
                 {"SMR_SEED_HOT_BIN": "64", "SMR_SEED_HOT_SUB": "8192", "SMR_SEED_DEDUP": "0"},     # ... one sub-range per large bin, no search for repeats
                 {"SMR_SEED_DEDUP": "16", "SMR_PG_CAND_CAP": "8"}]                                   # repeats + waves that overflow their candidate pool (the DFS kernel meets rewritten tuples)


@pytest.mark.parametrize("mode", [0, 1], ids=["pg", "dfs"])
@pytest.mark.parametrize("env", SKEW_VARIANTS, ids=lambda e: ",".join("%s=%s" % (k.replace("SMR_SEED_", ""), v) for k, v in e.items()))
def test_skewed_batch_sorted_by_several_blocks_and_searched_once_per_seed(tmp_path, monkeypatch, env, mode):
    """The batch of test_batch_dominated_by_one_sequence with the thresholds of the two skew paths lowered to its size: coarse bins far larger
    than the average are sorted by k_seed_hbins_* (several blocks per bin), and k_seed_dedup / k_seed_prop search a seed that thousands of
    windows share once (mode 1, the exact-counter DFS kernel, searches every tuple: only the sort differs).  Records, counters and the seed
    hits of every (strand, pass) equal the oracle's."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    w = Workload(str(tmp_path), db_nt=200_000, n_reads=3500, seed=77, frac_db=0.6)
    hot = [w.seqs[20], w.seqs[41], w.seqs[7][:60] + w.seqs[20][60:]]          # the third shares most windows with the first
    for i in range(500, 3500):
        w.seqs[i] = hot[i % 3]
    w.reads = smr.Reads.from_seqs(w.seqs)
    lam, K = GUMBEL_UNIFORM
    w.minimal_score = smr.minimal_score(lam, K, w.parts[0].info(), len(w.seqs), sum(map(len, w.seqs)))
    e = smr.Engine(0)
    try:
        e.set_seed_mode(mode)
        test_seed_scan_matches_oracle(e, w)
        recs_o, ctr_o = w.oracle_records()
        recs_g, ctr_g = w.gpu_records(e)
        _compare(recs_g, recs_o, "skewed batch %s" % env)
        assert ctr_g["num_aligned"] == ctr_o["num_aligned"] > 1000
        if mode == 0 and env.get("SMR_SEED_DEDUP") != "0":
            tuples, _, meta = e.seed_tuples()
            assert sum(1 for t in tuples if int(t) >> 63) > meta["n"] // 2       # most tuples of the last launch repeat another one's seed
    finally:
        e.close()


@pytest.mark.parametrize("mode", ["1", "0"], ids=["shared", "per-part"])
def test_one_seed_sort_for_the_parts_and_references_of_a_batch(tmp_path, monkeypatch, mode):
    """Two --ref, the first cut into several index parts (processor.cpp:219-277 loops (index, part) over the same reads): from the first part on
    the searches of every part walk SIX sorted arrays built once for the batch (every read, every window; SMR_SEED_SHARED=1, the default) and
    skip the tuples of the reads that are not in the part's (strand, pass); reads with ambiguous letters keep a sort per part.  Same records and
    counters as the oracle and as the per-part sort (SMR_SEED_SHARED=0); a second alignment of the same batch after a state reset builds again."""
    monkeypatch.setenv("SMR_SEED_SHARED", mode)
    os.makedirs(str(tmp_path / "a"))
    w1 = Workload(str(tmp_path / "a"), db_nt=260_000, n_reads=1800, seed=91, frac_db=0.5, n_rate=0.004, max_mb=0.5)
    assert w1.stats.nparts >= 3
    os.makedirs(str(tmp_path / "b"))
    # the second DB: a third of the first one's sequences, every 50th letter changed -- reads align to both
    out, keep, k = [], True, 0
    for line in open(w1.db):
        if line.startswith(">"):
            k += 1
            keep = k % 3 == 0
            if keep:
                out.append(line)
        elif keep:
            a = list(line.rstrip("\n"))
            for i in range(7, len(a), 50):
                a[i] = "ACGT"[("ACGT".index(a[i]) + 1) & 3] if a[i] in "ACGT" else a[i]
            out.append("".join(a) + "\n")
    db2 = str(tmp_path / "b" / "second.fasta")
    open(db2, "w").write("".join(out))
    w2 = Workload(str(tmp_path / "b"), db_fasta=db2, seqs=w1.seqs)
    ws = [w1, w2]
    run = orc.Run(w1.seqs)
    for k, x in enumerate(ws):
        for part in range(x.stats.nparts):
            po = orc.default_params(minimal_score=x.minimal_score, num_alignments=2)
            po.index_num, po.part, po.is_last_index_part = k, part, int(k == 1 and part == x.stats.nparts - 1)
            run.align_part(x.prefix, x.db, x.stats, part, po)
    recs_o = run.records()
    n_al = run.counters.num_aligned
    run.close()
    e = smr.Engine(0)
    try:
        ps = [smr.default_params(minimal_score=x.minimal_score, num_alignments=2) for x in ws]
        for rep in range(2):
            e.prof_reset()
            smr.align(e, w1.reads, [x.parts for x in ws], ps)
            _compare(e.records(), recs_o, "two references, %d + %d parts, SMR_SEED_SHARED=%s" % (w1.stats.nparts, w2.stats.nparts, mode))
            assert e.counters(2)["num_aligned"] == n_al > 300
            pr = e.prof()
            if mode == "1":
                assert pr.n_seed_shared_builds == 1 and pr.n_seed_shared == 6 * (w1.stats.nparts + w2.stats.nparts), (pr.n_seed_shared_builds, pr.n_seed_shared)
            else:
                assert pr.n_seed_shared_builds == 0 and pr.n_seed_shared == 0
    finally:
        e.close()


def _dense_neighbourhood_workload(tmp):
    """A DB that holds, for ONE 18-mer A+B which it does not contain itself, every string within one error of it that a half-seed search can
    accept: under the key A the 255 ten-letter continuations lev1_entry accepts for B (all but the four exact ones), and in front of the key B
    the 255 accepted for A read backwards; each in a sequence of its own with its own random flanks.  Reads: a sequence's flanks around A+B."""
    pw, acgt = 9, "ACGT"
    rng = np.random.Generator(np.random.PCG64(2024))
    A = [(i + 1) & 3 for i in range(pw)]                    # CGTACGTAC: no letter equals a neighbour or a neighbour's neighbour -- the patterns with the most accepted strings
    B = [(3 * i + 2) & 3 for i in range(pw)]                # GCATGCATG

    def accepted(pat):
        P = sum(c << (2 * i) for i, c in enumerate(pat))
        u = np.uint64
        T = np.arange(4 ** (pw + 1), dtype=np.uint64)
        m2 = u((1 << (2 * pw)) - 1)
        x0, x1, x2 = (u(P) ^ T) & m2, (u(P) ^ (T >> u(2))) & m2, ((u(P) >> u(2)) ^ T) & (m2 >> u(2))
        y = x0 | u(1 << (2 * pw))
        a2 = np.log2((y & (~y + u(1))).astype(np.float64)).astype(np.uint64) & ~u(1)
        ok = ((((x0 >> a2) >> u(2)) == 0) | ((x1 >> a2) == 0) | ((x2 >> a2) == 0)) & (x0 != 0)
        return [[(int(t) >> (2 * i)) & 3 for i in range(pw + 1)] for t in T[ok]]

    def rnd(n):
        return [int(x) for x in rng.integers(0, 4, n)]

    refs = []
    for t in accepted(B):
        refs.append((rnd(36), A + t, rnd(60)))
    for t in accepted(A[::-1]):
        refs.append((rnd(35), t[::-1] + B, rnd(60)))
    assert len(refs) == 2 * 255
    db = os.path.join(tmp, "dense.fasta")
    with open(db, "w") as f:
        for i, (l, m, r) in enumerate(refs):
            f.write(">dense%d\n%s\n" % (i, "".join(acgt[c] for c in l + m + r)))
    seqs = []
    for j in range(0, len(refs), 9):                        # A+B at read position 36 (a window of every pass), flanks of sequence j
        l, m, r = refs[j]
        seqs.append("".join(acgt[c] for c in (l[-36:] if len(l) >= 36 else [0] + l) + A + B + r[:60]))
    seqs += ["".join(acgt[c] for c in rnd(150)) for _ in range(40)]
    return Workload(tmp, db_fasta=db, seqs=seqs)


def test_a_window_with_hundreds_of_hits(engine, tmp_path):
    """More accepted strings than the 128 entries the lane-local hit lists grow to by doubling (round 4: SMR_ERR_CAPACITY): the lists then get the
    size no search can exceed (SEED_HCAP_BOUND).  The (read, id, window) triples of every strand and pass and the records equal the oracle's."""
    w = _dense_neighbourhood_workload(str(tmp_path))
    L = orc.lib()
    ix = L.orc_index_load(w.prefix.encode(), 0, 18)
    ids = (C.c_uint32 * 4096)()
    v = iseq_for_strand(w.seqs[0], 0)
    z = C.c_int()
    assert L.orc_window_hits(ix, v.ctypes.data, 36, 18, 0, 0, ids, 4096, C.byref(z)) > 300      # forward + reverse neighbours of A+B
    L.orc_index_free(ix)
    test_seed_scan_matches_oracle(engine, w)
    recs_o, ctr_o = w.oracle_records()
    recs_g, ctr_g = w.gpu_records(engine)
    _compare(recs_g, recs_o, "dense neighbourhood")
    assert engine.prof().hit_list_cap in (259, 518)          # lists of 128 were too short: the bound of one search (31 * 9 - 20), or of a reverse search of the DFS kernel on top of the forward list
    assert ctr_g["num_aligned"] == ctr_o["num_aligned"] >= 50


def test_seed_work_counters_match_oracle(wl):
    """The numerator of bench.py's roofline: the device work counters of the per-lane DFS seed kernel (smr_prof_get in seed mode 1:
    windows searched, 9-mer lookups, trie nodes visited, bucket entries compared, seed hits) equal the oracle's counters of the
    reference's sequential scan (oracle/smr_oracle.c, counters next to traversetrie_align) on the same workload -- and the
    forward Smith-Waterman calls of the candidate walk equal the oracle's ssw_align calls."""
    e = smr.Engine(0)
    try:
        e.set_seed_mode(1)
        _, ctr_o = wl.oracle_records()
        e.prof_reset()
        wl.gpu_records(e)
        p = e.prof()
        got = dict(n_windows=p.n_windows, n_lookup=p.n_lookup, n_node=p.n_node, n_entry=p.n_entry, n_hit=p.n_hit, n_sw_fwd=p.n_sw_fwd)
        exp = {k: ctr_o[k] for k in got}
        assert got == exp
        # reverse passes: the reference runs one per accepted ssw_align; here only the alignments that are still stored when the part is done get one
        assert 0 < p.n_sw_rev <= ctr_o["n_sw_rev"]
        assert exp["n_entry"] > exp["n_windows"] > 0 and exp["n_node"] > 0 and p.n_read_bytes > 0
    finally:
        e.close()


def _closed_form_lev1(P, T, pw):
    """smr_seed.hpp::lev1_entry on Python ints -> (accepted, zero-error); proven equal to the reference's tables in tests/test_lev_closed_form.py"""
    eq0 = [((P >> (2 * i)) & 3) == ((T >> (2 * i)) & 3) for i in range(pw)]
    eq1 = [((P >> (2 * i)) & 3) == ((T >> (2 * (i + 1))) & 3) for i in range(pw)]
    eq2 = [((P >> (2 * (i + 1))) & 3) == ((T >> (2 * i)) & 3) for i in range(pw - 1)]

    def lead(v):
        n = 0
        while n < len(v) and v[n]:
            n += 1
        return n

    a, s0, s1, s2 = lead(eq0), lead(eq0[::-1]), lead(eq1[::-1]), lead(eq2[::-1])
    return (a + s2 >= pw - 1) or (a + s0 >= pw - 1) or (a + s1 >= pw), a >= pw


def _pg_key(T, frm, cnt):
    k = 0
    for q in range(cnt):
        k = (k << 2) | ((T >> (2 * (frm + q))) & 3)
    return k


def _host_recount_of_the_search(pg, root3, tuples, cbase, meta, direction, zero_slots, pw, full=False):
    """What k_seed_pg<direction> counts as its algorithmic HBM bytes (smr_seed_pg.hpp, C_B_PG0 / C_B_PG1), recomputed on the host from the
    pigeonhole layout of the HOST transform and the launch's sorted tuples: per tuple 8 B + its block-table entry 8 B (+ 1 B: the window's
    group bit, reverse); per search with directories 8 directory words; 4 B per string inside its four exact-key ranges; 8 B per accepted
    {rank, id}; the segment written (4 B per word) + 4 B for the window slot pointing to it; 2 B per wave chunk (its coarse bin).
    -> (bytes, slots whose forward search ended with a 0-error match)"""
    h = pw // 2
    n_fwd, n_all, fb, cb, nkh = meta["n_fwd"], meta["n"], meta["fb"], meta["cb"], meta["nkh"]
    lo, hi = (0, n_fwd) if direction == 0 else (n_fwd, n_all)
    chunks = ((hi + 63) >> 6) - (lo >> 6) if hi > lo or direction == 0 else 0
    if direction == 0:
        chunks = (n_fwd + 63) >> 6
    else:
        chunks = ((n_all + 63) >> 6) - (n_fwd >> 6)
    total = 2 * chunks + (hi - lo) * (8 + 8 + direction)
    zeros = set()
    segs, repeats = set(), []
    import bisect
    cb_list = [int(x) for x in cbase]
    for i in range(lo, hi):
        t = int(tuples[i])
        slot, chars = t & 0xFFFFFFFF, (t >> 32) & ((1 << cb) - 1)
        if t >> 63:                                                # a repeated seed (k_seed_dedup): the search only reads the tuple; k_seed_prop reads it again
            repeats.append((slot, (t >> 32) & 0x7FFFFFFF))         # + the representative's bit, and where that has a segment the slot word is read and written
            total += 9 - (8 + direction)
            continue
        c = bisect.bisect_right(cb_list, i) - 1                     # the coarse bin tuple i lies in (empty bins begin where the next one does)
        key = (c << fb) | (t >> (32 + cb))
        if direction == 1 and slot in zero_slots:
            continue                                               # no reverse search after a 0-error forward match (paralleltraversal.cpp:188)
        e = 2 * (2 * (key - (nkh if direction else 0)) + direction)
        off4, metaw = int(root3[e]), int(root3[e + 1])
        if off4 == 0xFFFFFFFF:
            continue
        n, cA, cB = metaw & 0xFFFFFF, (metaw >> 24) & 15, metaw >> 28
        blk = off4 * 4
        P = chars
        ranges = []                                                # (which key, first string in TA / TB numbering, count)
        if cA == 0:
            strings = blk
            ranges.append((0, 0, n))
        else:
            total += 32
            nA, nB = (1 << (2 * cA)) + 1, (1 << (2 * cB)) + 1
            dirA, dirB = blk, blk + nA
            strings = blk + nA + nB
            kA, kb0, kb1 = _pg_key(P, 0, cA), _pg_key(P, h, cB), _pg_key(P, h - 1, cB)
            if cB == pw - h:
                lo2 = 4 * _pg_key(P, h + 1, cB - 1); hi2 = lo2 + 4
            else:
                lo2 = _pg_key(P, h + 1, cB); hi2 = lo2 + 1
            ranges.append((0, int(pg[dirA + kA]), int(pg[dirA + kA + 1]) - int(pg[dirA + kA])))
            ranges.append((1, n + int(pg[dirB + kb0]), int(pg[dirB + kb0 + 1]) - int(pg[dirB + kb0])))
            if kb1 != kb0:
                ranges.append((2, n + int(pg[dirB + kb1]), int(pg[dirB + kb1 + 1]) - int(pg[dirB + kb1])))
            ranges.append((3, n + int(pg[dirB + lo2]), int(pg[dirB + hi2]) - int(pg[dirB + lo2])))
        mA, mB = (1 << (2 * cA)) - 1, (1 << (2 * cB)) - 1
        cands = []
        for w, u0, cnt in ranges:
            total += 4 * cnt
            for u in range(u0, u0 + cnt):
                T = int(pg[strings + u])
                tb = (T >> (2 * h)) & mB
                dup = (w != 0 and ((T ^ P) & mA) == 0) or (w == 3 and (tb == ((P >> (2 * h)) & mB) or tb == ((P >> (2 * h - 2)) & mB)))
                if dup:
                    continue
                acc, zero = _closed_form_lev1(P, T, pw)
                if acc:
                    ri = strings + (2 if cA else 1) * n + 2 * u
                    cands.append((int(pg[ri]), int(pg[ri + 1]), zero and not full))
        total += 8 * len(cands)
        hl, zero_end = [], False
        for rank, idc, cond in sorted(cands):
            present = idc in hl
            if direction == 0 and cond and not present:
                hl, zero_end = [idc], True
                break
            if not present:
                hl.append(idc)
        if hl:
            total += 4 if len(hl) == 1 else 4 * (1 + len(hl)) + 4      # one hit: it lies in the window's slot word (SEED_SEG_INLINE), no segment
            segs.add(slot)
        if zero_end:
            zeros.add(slot)
    for slot, rep in repeats:
        if rep in segs:
            total += 8
        if rep in zeros:
            zeros.add(slot)
    return total, zeros


@pytest.mark.parametrize("strand,pass_,dedup", [(0, 0, None), (1, 2, None), (0, 0, "2"), (1, 2, "2")], ids=["s0p0", "s1p2", "s0p0-dedup", "s1p2-dedup"])
def test_pigeonhole_search_bytes_equal_a_host_recount(wl, monkeypatch, strand, pass_, dedup):
    """The roofline numerator of the dominant seed kernel is counted by the kernel itself (C_B_PG0 / C_B_PG1).  Here the same quantity is
    recomputed on the HOST -- from the pigeonhole layout the host transform builds (smr_build_pigeonhole) and the launch's sorted tuples, walking
    the four exact-key ranges of every search, the closed-form automaton, the duplicate rule and the reference's list rules in Python -- and
    must equal the device's counters exactly, for the forward and for the reverse launch.  The `dedup` cases search every repeated seed once
    (SMR_SEED_DEDUP=2: every key with two tuples counts as hot) and count what the repeats cost instead."""
    if dedup:
        monkeypatch.setenv("SMR_SEED_DEDUP", dedup)
    e = smr.Engine(0)
    try:
        e.upload_reads(wl.reads, 1)
        e.upload_index(wl.parts[0], 0)
        p = smr.default_params(minimal_score=wl.minimal_score)
        pg, root3 = smr.pigeonhole_layout(wl.parts[0])
        e.reset_state()
        e.prof_reset()
        e.seed_scan(0, p, strand, pass_)
        tuples, cbase, meta = e.seed_tuples()
        assert meta["redo"] == 0 and meta["n"] > 1000 and 0 < meta["n_fwd"] < meta["n"]
        kp = e.prof_kernels()
        exp0, zeros = _host_recount_of_the_search(pg, root3, tuples, cbase, meta, 0, set(), 9)
        exp1, _ = _host_recount_of_the_search(pg, root3, tuples, cbase, meta, 1, zeros, 9)
        assert len(zeros) > 10                                    # the workload has exact seed matches: the reverse launch really skips searches
        assert (sum(1 for t in tuples if int(t) >> 63) > 20) == bool(dedup)
        assert int(kp["k_seed_pg<0>"]["bytes"]) == exp0, (int(kp["k_seed_pg<0>"]["bytes"]), exp0)
        assert int(kp["k_seed_pg<1>"]["bytes"]) == exp1, (int(kp["k_seed_pg<1>"]["bytes"]), exp1)
    finally:
        e.close()


def test_very_long_reads(engine, tmp_path):
    """8-12.5 kb reads (BASELINE config 5 is 5 kb PacBio; this is beyond it): k_chain needs more than 64 KB of dynamic LDS per
    workgroup (hipFuncSetAttribute), the packed SW kernel runs 512-row strips up to ~8 kb and the 32-bit kernel beyond, the
    traceback takes the wide kernel with bands of hundreds of diagonals"""
    import numpy as np
    from sortmerna_amd import synth
    w = Workload(str(tmp_path), db_nt=400_000, n_reads=20, seed=43, family_size=4, mean_len=14000)
    codes, offs = synth.load_db_codes(w.db)
    rng = np.random.Generator(np.random.PCG64(7))
    seqs = []
    for i in range(6):
        sq = int(rng.integers(0, len(offs) - 1))
        full = int(offs[sq + 1] - offs[sq])
        ln = min(full, [8000, 9000, 10000, 11000, 12500, 6000][i])
        st = int(offs[sq] + rng.integers(0, full - ln + 1))
        out = []
        for c in codes[st:st + ln]:
            u = rng.random()
            if u < 0.03:
                continue
            if u < 0.06:
                out.append(int(rng.integers(0, 4)))
            out.append(int((c + rng.integers(1, 4)) & 3) if u > 0.96 else int(c))
        s = "".join("ACGT"[c] for c in out)
        if i % 2:
            s = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
        seqs.append(s)
    assert max(map(len, seqs)) > 11000
    w.seqs = seqs
    w.reads = smr.Reads.from_seqs(seqs)
    w.minimal_score = smr.minimal_score(0.618874, 0.343238, w.parts[0].info(), len(seqs), sum(map(len, seqs)))
    recs_o, ctr_o = w.oracle_records()
    recs_g, ctr_g = w.gpu_records(engine)
    _compare(recs_g, recs_o, "8-12.5 kb reads")
    assert ctr_g["num_aligned"] == ctr_o["num_aligned"] >= 5
    spans = [refrun.parse_record(r)["alignv"][0] for r in recs_o if r]
    assert max(a["read_end1"] - a["read_begin1"] for a in spans) > 8000


def test_empty_batch(engine, wl):
    r = smr.Reads.from_seqs([])
    p = smr.default_params(minimal_score=wl.minimal_score)
    smr.align(engine, r, [wl.parts], [p])
    assert engine.records() == []
    assert engine.counters(1)["num_aligned"] == 0

