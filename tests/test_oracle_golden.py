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
