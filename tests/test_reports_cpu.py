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
