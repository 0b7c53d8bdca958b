"""CPU: the report writers (smr_report_*: aligned/other FASTX, BLAST tabular + cigar/qstrand, SAM) fed with the reference's own
per-read records must reproduce the reference's own report files row for row (tests/golden/golden.json holds what the unmodified
reference wrote for `-fastx -other -blast '1 qstrand cigar' -sam`): report_blast.cpp:253-354, report_sam.cpp:64-152,
report_fx_base.cpp:176-205, Read::calc_miss_gap_match read.cpp:547-589."""
import os

import pytest

import sortmerna_amd as smr
from sortmerna_amd import report
from helpers import fastx, golden

CASES = ["t0", "t9", "syn_default", "syn_all", "real_default"]


@pytest.mark.parametrize("case", CASES)
def test_reports_equal_the_reference_reports(case, tmp_path):
    g = golden.load()[case]
    db, rd, _ = golden.inputs(case)
    recs = golden.records(case)
    reads = fastx.read_fastx(rd)
    assert len(reads) == len(recs)
    parts = smr.Index.build(db, 18, 3072.0, 10000, 0)
    rep = report.Report(str(tmp_path), is_fastq=False, fastx=True, other=True, blast_cols=["qstrand", "cigar"], sam=True)
    fr, fq = report.corrected_sizes(g["log"]["K"][0], parts[0].info(), g["readstats"]["all_reads_count"], g["readstats"]["all_reads_len"])
    rep.set_db(0, g["log"]["lambda"][0], g["log"]["K"][0], fr, fq)
    for k, ix in enumerate(parts):
        rep.set_part(0, k, ix)
    for (hdr, seq, qual), rec in zip(reads, recs):
        rep.add(hdr, seq, qual, rec)
    rep.close()
    blast = [l.rstrip("\n").split("\t") for l in open(tmp_path / "aligned.blast")]
    exp = [l.split("\t") for l in g["blast"]]
    assert len(blast) == len(exp)
    for a, b in zip(blast, exp):
        # column 11 is the e-value printed with 3 significant digits: the golden log only has lambda and K rounded to 6 digits
        # (the reference computes with ALP's full doubles), so allow one unit in the last printed digit there; all else exact
        assert a[:10] == b[:10] and a[11:] == b[11:], (a, b)
        assert abs(float(a[10]) - float(b[10])) <= 1.2e-2 * float(b[10]), (a, b)
    sam = [l.rstrip("\n") for l in open(tmp_path / "aligned.sam") if not l.startswith("@")]
    assert sam == [l for l in g["sam"] if not l.startswith("@")]
    if "aligned_ids" in g:                       # t0's multi-line FASTA is mis-split by the reference's feed: ids still agree
        ids = lambda p: [l.split()[0][1:] for l in open(p) if l.startswith(">")]
        assert ids(tmp_path / "aligned.fa") == g["aligned_ids"]
        assert ids(tmp_path / "other.fa") == g.get("other_ids", [])


REPORTS2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reports2")


@pytest.mark.parametrize("case", ["t0", "t9", "syn_default", "real_default"])
def test_pairwise_blast_sq_header_and_summary_equal_the_reference(case, tmp_path):
    """`-blast 0` (report_blast.cpp:130-252), `-sam -SQ` header (report_sam.cpp:155-211) and aligned.log (summary.cpp:102-175):
    fixtures written by the unmodified reference (tests/golden/make_golden_reports2.py)."""
    g = golden.load()[case]
    db, rd, _ = golden.inputs(case)
    recs = golden.records(case)
    reads = fastx.read_fastx(rd)
    parts = smr.Index.build(db, 18, 3072.0, 10000, 0)
    hdr_exp = open(os.path.join(REPORTS2, case + ".sam_header.txt")).read().splitlines()
    cmdline = [l for l in hdr_exp if l.startswith("@PG")][0].split("CL:", 1)[1]
    rep = report.Report(str(tmp_path), is_fastq=False, fastx=False, other=False, blast_pairwise=True, sam=True, sam_sq=True, cmdline=cmdline)
    fr, fq = report.corrected_sizes(g["log"]["K"][0], parts[0].info(), g["readstats"]["all_reads_count"], g["readstats"]["all_reads_len"])
    rep.set_db(0, g["log"]["lambda"][0], g["log"]["K"][0], fr, fq)
    for k, ix in enumerate(parts):
        rep.set_part(0, k, ix)
    for (hdr, seq, qual), rec in zip(reads, recs):
        rep.add(hdr, seq, qual, rec)
    rep.close()
    got = open(tmp_path / "aligned.blast").read().split("\n")
    exp = open(os.path.join(REPORTS2, case + ".pairwise.txt")).read().split("\n")
    assert len(got) == len(exp)
    for a, b in zip(got, exp):
        if a.startswith("Score: "):               # "Score: S bits (B)\tExpect: E\tstrand: +": E has 3 digits from lambda, K known to 6 (see above)
            fa, fb = a.split("\t"), b.split("\t")
            assert fa[0] == fb[0] and fa[2] == fb[2], (a, b)
            ea, eb = float(fa[1].split(": ")[1]), float(fb[1].split(": ")[1])
            assert abs(ea - eb) <= 1.2e-2 * eb, (a, b)
        else:
            assert a == b, (a, b)
    assert [l for l in open(tmp_path / "aligned.sam").read().splitlines() if l.startswith("@")] == hdr_exp
    # aligned.log
    log_exp = open(os.path.join(REPORTS2, case + ".log.txt")).read()
    lines = log_exp.split("\n")
    ts = lines[-3].strip() + "\n"                # ctime() text incl. its newline
    rs = g["readstats"]
    opt = g["options"]
    mismatch = int(opt[opt.index("-mismatch") + 1]) if "-mismatch" in opt else -3
    report.write_summary(str(tmp_path / "aligned.log"),
                         [dict(ref_file=os.path.basename(db), skiplengths=[18, 9, 3], lam=g["log"]["lambda"][0], K=g["log"]["K"][0],
                               minimal_score=g["log"]["minimal_score"][0], reads_matched=rs["reads_matched_per_db"][0])],
                         [os.path.basename(rd)], rs["all_reads_count"], rs["num_aligned"], rs["all_reads_len"], rs["min_read_len"], rs["max_read_len"],
                         mismatch=mismatch, score_N=mismatch, sam_sq=True, threads=1, cmdline=lines[1][4:], pid="", timestamp=ts)
    assert open(tmp_path / "aligned.log").read() == log_exp


PAIRED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "paired")


def _paired_inputs():
    import json
    import struct
    g = json.load(open(os.path.join(PAIRED, "paired.json")))
    m1 = fastx.read_fastx(os.path.join(PAIRED, "paired_1.fastq"))
    m2 = fastx.read_fastx(os.path.join(PAIRED, "paired_2.fastq"))
    b = open(os.path.join(PAIRED, "paired.records.bin"), "rb").read()
    (n,) = struct.unpack_from("<I", b, 0)
    o, recs = 4, []
    for _ in range(n):
        (l,) = struct.unpack_from("<I", b, o)
        recs.append(b[o + 4:o + 4 + l])
        o += 4 + l
    assert n == 2 * len(m1) == 2 * len(m2)
    return g, m1, m2, recs


@pytest.mark.parametrize("variant", ["two_files", "paired_in", "paired_out", "out2", "paired_in_out2", "paired_out_out2", "sout", "out2_sout"])
def test_paired_fastx_reports_equal_the_reference(variant, tmp_path):
    """two mate files with -paired_in / -paired_out / -out2 / -sout: smr_report_add_pair writes the files the unmodified reference writes
    (names, which read goes where, order): ReportFastx::append report_fastx.cpp:57-133, ReportFxOther::append report_fx_other.cpp:49-113"""
    g, m1, m2, recs = _paired_inputs()
    opt = g[variant]["options"]
    rep = report.Report(str(tmp_path), is_fastq=True, fastx=True, other=True, paired_in="-paired_in" in opt, paired_out="-paired_out" in opt,
                        out2="-out2" in opt, sout="-sout" in opt)
    for i in range(len(m1)):
        rep.add_pair(m1[i] + (recs[2 * i],), m2[i] + (recs[2 * i + 1],))
    rep.close()
    got = {fn: [l.split()[0][1:] for l in open(tmp_path / fn).readlines()[0::4]] for fn in sorted(os.listdir(tmp_path)) if fn.endswith(".fq")}
    assert got == g[variant]["files"]
    # and the records themselves are copied verbatim
    by_id = {r[0].split()[0][1:]: r for r in m1 + m2}
    for fn, ids in got.items():
        lines = open(tmp_path / fn).read().split("\n")
        for k, rid in enumerate(ids[:20]):
            h, s, q = by_id[rid]
            assert lines[4 * k:4 * k + 4] == [h, s, "+", q]


def test_invalid_pairing_options_are_rejected(tmp_path):
    with pytest.raises(smr.SmrError):
        report.Report(str(tmp_path), is_fastq=True, paired_in=True, paired_out=True)
    with pytest.raises(smr.SmrError):
        report.Report(str(tmp_path), is_fastq=True, paired_in=True, sout=True)


def test_gzip_reports_hold_the_same_text(tmp_path):
    """zip_out (the reference deflates its reports for gzip reads files / -zip-out 1 and appends .gz to the names): same text inside"""
    import gzip
    case = "syn_default"
    g = golden.load()[case]
    db, rd, _ = golden.inputs(case)
    recs = golden.records(case)
    reads = fastx.read_fastx(rd)
    parts = smr.Index.build(db, 18, 3072.0, 10000, 0)
    outs = {}
    for z in (False, True):
        d = tmp_path / ("z%d" % z)
        os.makedirs(d)
        rep = report.Report(str(d), is_fastq=False, fastx=True, other=True, blast_cols=["qstrand", "cigar"], sam=True, zip_out=z)
        fr, fq = report.corrected_sizes(g["log"]["K"][0], parts[0].info(), g["readstats"]["all_reads_count"], g["readstats"]["all_reads_len"])
        rep.set_db(0, g["log"]["lambda"][0], g["log"]["K"][0], fr, fq)
        rep.set_part(0, 0, parts[0])
        for (hdr, seq, qual), rec in zip(reads, recs):
            rep.add(hdr, seq, qual, rec)
        rep.close()
        outs[z] = {fn: (gzip.open(d / fn).read() if z else open(d / fn, "rb").read()) for fn in sorted(os.listdir(d))}
    assert sorted(outs[True]) == ["aligned.blast.gz", "aligned.fa.gz", "aligned.sam.gz", "other.fa.gz"]      # the reference's names
    assert {k + ".gz": v for k, v in outs[False].items()} == outs[True]


def test_readstats_record_equals_the_reference_blob(tmp_path):
    """smr_readstats_record / smr_readstats_key = the KVDB entry of Readstats::store_to_db (readstats.cpp:133-174, 82-91, 291-295): parsed
    fields equal the golden Readstats of every case; with the reference binary at hand the key and every byte of a live run"""
    import ctypes as C
    import sortmerna_amd.capi as capi
    from helpers import golden, paths, refrun
    L = capi.load()

    def blob(rs):
        per = (C.c_uint64 * len(rs["reads_matched_per_db"]))(*rs["reads_matched_per_db"])
        n = L.smr_readstats_record(rs["all_reads_count"], rs["all_reads_len"], rs["min_read_len"], rs["max_read_len"], rs["num_aligned"], rs["num_short"], per, len(per), None, 0)
        buf = C.create_string_buffer(n)
        assert L.smr_readstats_record(rs["all_reads_count"], rs["all_reads_len"], rs["min_read_len"], rs["max_read_len"], rs["num_aligned"], rs["num_short"], per, len(per), buf, n) == n
        return buf.raw

    for case, g in golden.load().items():
        rs = g.get("readstats")
        if rs:
            assert refrun.parse_readstats(blob(rs)) == rs, case
    if not (paths.have_reference() and paths.have_ref_bin()):
        return
    db, rd, _ = golden.inputs("t9")
    res = refrun.run_reference([db], [rd], str(tmp_path / "wd"), threads=1)
    assert res.rc == 0
    entries = {k: v for k, v in res.kvdb.items() if b"_" not in k}
    assert len(entries) == 1
    key, val = next(iter(entries.items()))
    files = (C.c_char_p * 1)(rd.encode())
    kb = C.create_string_buffer(32)
    L.smr_readstats_key(files, 1, kb, 32)
    assert kb.value == key
    assert blob(refrun.parse_readstats(val)) == val
