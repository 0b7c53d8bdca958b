"""CPU: the oracle (oracle/smr_oracle.c, the plain-C restatement of the hot path) pinned against
  (1) the reference's own golden vectors: t2's BLAST row (scripts/test.jinja:265-266) and t9's SAM rows (:447-477),
  (2) per-read KVDB records (Read::toBinString bytes) the UNMODIFIED reference produced for the committed inputs
      (tests/golden/*.records.bin, made by tests/golden/make_golden.py), for the option variants of helpers/cases.py CASES,
  (3) when oracle/_ref/sortmerna_ref and /root/reference are present: a live run on the bundled set4 reads x
      silva-arc-16s-id95 with the REFERENCE-built index files.
The index consumed in (1)/(2) is built by our own host builder (smr_index_build, plain C++ in libsmr_hip.so, no GPU
involved), so these tests pin the builder as well."""
import os

import pytest

import sortmerna_amd as smr
from helpers import golden, orc, paths, refrun
from helpers.cases import CASES, oracle_run



@pytest.mark.parametrize("case", CASES)
def test_oracle_records_equal_reference_records(case, tmp_path):
    g = golden.load()[case]
    o = oracle_run(case, tmp_path)
    assert max(o["nparts"]) == g["index_parts"]
    exp = golden.records(case)
    bad = [i for i, (a, b) in enumerate(zip(o["records"], exp)) if a != b]
    assert not bad, "%s: %d records differ, first %d\n orc=%s\n ref=%s" % (
        case, len(bad), bad[0], refrun.parse_record(o["records"][bad[0]]), refrun.parse_record(exp[bad[0]]))
    assert o["num_aligned"] == g["readstats"]["num_aligned"] == g["log"]["num_aligned"]
    assert o["per_db"] == g["readstats"]["reads_matched_per_db"]
    assert o["num_short"] == g["readstats"]["num_short"]


def test_the_reference_ignores_its_passes_option():
    """`-passes 18,6,2` never reaches the hot path of the reference (options.cpp:704-732 appends vector<uint32_t>(18) -- zeros -- and never parses
    the last number; refstats.cpp:159-165 then installs L, L/2, 3): its records are the default run's, byte for byte.  The strides themselves
    (smr_params.skiplengths) are compared with the oracle in tests/test_gpu_parity.py / tests/test_emu_kernels.py."""
    assert golden.load()["syn_passes1862"]["options"] == ["-passes", "18,6,2"]
    assert golden.records("syn_passes1862") == golden.records("syn_default")


def test_t2_blast_row_of_the_reference_test_suite(tmp_path):
    """scripts/test.jinja:265-266 (expected row of test t2) re-derived from the oracle's record."""
    expected = ["AB271211", "Unc49508", "93.5", "1430", "64", "30", "58", "1487", "1", "1446", "0", "2069", "+",
                "57S57M2I12M2D4M2I29M1D11M2I3M2D11M1I7M1D13M5D4M3D9M2D3M7D1260M"]
    assert golden.load()["t0"]["blast"][0].split("\t") == expected       # the reference binary reproduces its golden
    o = oracle_run("t0", tmp_path)
    r = refrun.parse_record(o["records"][0])
    a = r["alignv"][0]
    assert a["score1"] == 2430                                           # test.jinja:165
    assert [str(a["read_begin1"] + 1), str(a["read_end1"] + 1), str(a["ref_begin1"] + 1), str(a["ref_end1"] + 1)] == expected[6:10]
    assert golden.cigar_string(a["cigar"], a["read_begin1"], a["read_end1"], a["readlen"]) == expected[13]
    assert ("+" if a["strand"] else "-") == expected[12]
    # alignment length / gaps from the CIGAR (Read::calc_miss_gap_match, read.cpp:547-589)
    gaps = sum(c >> 4 for c in a["cigar"] if (c & 0xF) != 0)
    assert str(gaps) == expected[5]
    assert str(a["read_end1"] - a["read_begin1"] + 1) == expected[3]


def test_t9_sam_rows_of_the_reference_test_suite(tmp_path):
    """scripts/test.jinja:447-477: forward hit at POS 1 and reverse-complement hit at POS 102, 101M, AS:i:202."""
    rows = [l.split("\t") for l in golden.load()["t9"]["sam"] if not l.startswith("@")]
    assert [(r[1], r[3], r[5], r[11]) for r in rows] == [("0", "1", "101M", "AS:i:202"), ("16", "102", "101M", "AS:i:202")]
    o = oracle_run("t9", tmp_path)
    r = refrun.parse_record(o["records"][0])
    got = sorted((0 if a["strand"] else 16, a["ref_begin1"] + 1, golden.cigar_string(a["cigar"], a["read_begin1"], a["read_end1"], a["readlen"]),
                  a["score1"]) for a in r["alignv"])
    assert got == [(0, 1, "101M", 202), (16, 102, "101M", 202)]


@pytest.mark.skipif(not (paths.have_reference() and paths.have_ref_bin()), reason="needs /root/reference + oracle/_ref/sortmerna_ref")
@pytest.mark.parametrize("extra,params", [([], {}), (["-num_alignments", "0"], {"num_alignments": 0})], ids=["default", "all"])
def test_oracle_vs_live_reference_on_bundled_data(tmp_path, extra, params):
    """set4 (first 1500 reads) x silva-arc-16s-id95 with the index files the REFERENCE built (CMPH ids)."""
    db = os.path.join(paths.REF_DATA, "rRNA_databases", "silva-arc-16s-id95.fasta")
    src = os.path.join(paths.REF_DATA, "set4_mate_pairs_metatranscriptomics_1.fastq")
    reads = os.path.join(str(tmp_path), "reads.fastq")
    with open(src) as f, open(reads, "w") as g:
        for i, l in enumerate(f):
            if i >= 4 * 1500:
                break
            g.write(l)
    cache = os.path.join(paths.ORACLE_DIR, "_ref", "idx_cache")      # index build takes ~10 s: keep it between runs
    res = refrun.run_reference([db], [reads], str(tmp_path / "wd"), extra=extra + ["-v"], threads=1, idx_dir=cache)
    assert res.rc == 0, res.stdout[-1500:]
    prefix = refrun.index_prefix_for(cache, db)
    st = orc.load_stats(prefix)
    from helpers import fastx
    seqs = [r[1] for r in fastx.read_fastx(reads)]
    ms, _, _ = orc.minimal_score(res.log["lambda"][0], res.log["K"][0], st, len(seqs), sum(map(len, seqs)))
    assert ms == res.log["minimal_score"][0]
    run = orc.Run(seqs)
    p = orc.default_params(minimal_score=ms, **params)
    for part in range(st.nparts):
        p.part = part
        p.is_last_index_part = int(part == st.nparts - 1)
        run.align_part(prefix, db, st, part, p)
    recs = run.records()
    exp = [res.kvdb.get(b"0_%d" % i, b"") for i in range(len(seqs))]
    bad = [i for i in range(len(seqs)) if recs[i] != exp[i]]
    assert not bad, "%d differ, first %d" % (len(bad), bad[0])
    assert run.counters.num_aligned == res.log["num_aligned"] > 300
    run.close()


def _affine_score(read, ref, match, mismatch, go, ge):
    """best local score of the affine recurrence H = max(0, diag + s, E, F) (first gap base costs go, further ones ge; go >= ge), column by column
    in numpy: F(i) = max over k < i of Hpre(k) - go - (i - 1 - k) ge with Hpre = max(0, diag + s, E) -- a gap opened from a cell that F itself
    raised never wins while go >= ge -- i.e. a prefix maximum of Hpre(k) + k ge."""
    import numpy as np
    m = len(read)
    H = np.zeros(m + 1, dtype=np.int64)
    E = np.zeros(m + 1, dtype=np.int64)
    idx = np.arange(m + 1, dtype=np.int64)
    best = 0
    for c in ref:
        E = np.maximum(E - ge, H - go)
        s = np.where(read == c, match, mismatch)
        hp = np.zeros(m + 1, dtype=np.int64)
        hp[1:] = np.maximum(np.maximum(H[:-1] + s, E[1:]), 0)
        pm = np.maximum.accumulate(hp[1:] + idx[1:] * ge)                   # over rows 1..i
        F = np.full(m + 1, -10**9, dtype=np.int64)
        F[2:] = pm[:-1] - go - (idx[2:] - 1) * ge
        H = np.maximum(hp, F)
        H[0] = 0
        best = max(best, int(H.max()))
    return best


def test_only_schemes_with_gap_open_above_gap_ext_follow_the_affine_recurrence():
    """Why libsmr_hip refuses gap_open <= gap_ext (smr_engine.hip scheme_unsupported): the reference's 16-bit striped kernel leaves its lazy-F loop as soon
    as no lane has F - gap_ext > H - gap_open (ssw.c:496-507), which with gap_open == gap_ext is the case one cell behind a stripe boundary -- a longer gap
    across a boundary is lost.  Seeded pairs whose scores need the 16-bit kernel (> 255): the reference's result (the oracle's port of the striped kernels,
    pinned to ssw.c elsewhere; the compiled ssw.c itself when it is here) against the plain recurrence."""
    import ctypes as C
    import numpy as np
    L = orc.lib()
    real = None
    if os.path.isfile(os.path.join(paths.ORACLE_DIR, "_ref", "libssw_ref.so")):
        import sys
        sys.path.insert(0, os.path.join(paths.GOLDEN))
        import make_golden_ssw as G
        real = (G, G.ref_lib())
    differing = {}
    for match, mismatch, go, ge in [(2, -3, 3, 3), (2, -3, 5, 2), (2, -3, 4, 3)]:
        rng = np.random.Generator(np.random.PCG64(21))
        mat = np.array([(match if a == b else mismatch) if a < 4 and b < 4 else mismatch for a in range(5) for b in range(5)], dtype=np.int8)
        bad = 0
        for _ in range(400 if go == ge else 120):
            m = int(rng.integers(200, 260))
            ref = rng.integers(0, 4, m + 40, dtype=np.int8)
            a = list(ref[10:10 + m])
            for _k in range(int(rng.integers(1, 4))):
                p, ln = int(rng.integers(5, len(a) - 5)), int(rng.integers(1, 7))
                if rng.random() < 0.5:
                    del a[p:p + ln]
                else:
                    a[p:p] = list(rng.integers(0, 4, ln, dtype=np.int8))
            read = np.array(a, dtype=np.int8)
            r = orc.SswResult()
            assert L.orc_ssw(read.ctypes.data, len(read), ref.ctypes.data, len(ref), mat.ctypes.data, go, ge, 0, C.byref(r))
            if real:
                assert real[0].ssw_reference(real[1], read.tobytes(), ref.tobytes(), match, mismatch, mismatch, go, ge, 0)[0] == r.score1
            plain = _affine_score(read, ref, match, mismatch, go, ge)
            assert r.score1 <= plain and plain > 255
            bad += r.score1 != plain
        differing[(go, ge)] = bad
    assert differing[(3, 3)] >= 3 and differing[(5, 2)] == 0 and differing[(4, 3)] == 0, differing
