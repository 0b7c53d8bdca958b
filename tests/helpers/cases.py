"""The golden cases (tests/golden/) set up for the oracle and for the GPU path.  TEST INFRASTRUCTURE ONLY."""
import os

import sortmerna_amd as smr

from . import golden, orc

CASES = ["t0", "t9", "syn_default", "syn_all", "syn_best3", "syn_nobest2", "syn_F", "syn_R", "syn_full_search",
         "syn_seeds3_edges10", "syn_multipart", "syn_seeds1", "syn_minlis3", "syn_minlis1", "syn_N0", "syn_gaps32", "syn_L14", "syn_score3463", "syn_passes1862", "syn_e1em8", "real_default", "real_all", "two_db_default", "two_db_all"]


def build_case(case, tmpdir):
    """-> list over --ref of dict(db, parts (our builder), prefix (reference-format files), stats, minimal_score)"""
    g = golden.load()[case]
    dbs, _, seqs = golden.inputs(case)
    if not isinstance(dbs, list):
        dbs = [dbs]
    max_mb = g["params"].get("max_mb", 3072.0)
    lnwin, evalue = g["params"].get("lnwin", 18), g["params"].get("evalue", 1.0)
    out = []
    for k, db in enumerate(dbs):
        parts = smr.Index.build(db, lnwin, max_mb, 10000, 0)
        prefix = os.path.join(str(tmpdir), "idx%d" % k)
        smr.Index.write_files(parts, db, prefix)
        st = orc.load_stats(prefix)
        # read totals as the reference's Readfeed counted them (for the multi-line FASTA of t0 it mis-counts records,
        # SURVEY.md 0.3; everywhere else these equal len(seqs) / sum of lengths)
        ms, _, _ = orc.minimal_score(g["log"]["lambda"][k], g["log"]["K"][k], st, g["readstats"]["all_reads_count"], g["readstats"]["all_reads_len"], evalue)
        assert ms == g["log"]["minimal_score"][k]                       # refstats.cpp:238-265 restated
        out.append(dict(db=db, parts=parts, prefix=prefix, stats=st, minimal_score=ms))
    if case != "t0":
        assert (g["readstats"]["all_reads_count"], g["readstats"]["all_reads_len"]) == (len(seqs), sum(map(len, seqs)))
    return out, seqs


def oracle_run(case, tmpdir):
    g = golden.load()[case]
    idx, seqs = build_case(case, tmpdir)
    params = {k: v for k, v in g["params"].items() if k not in ("max_mb", "evalue")}
    run = orc.Run(seqs)
    for k, d in enumerate(idx):
        p = orc.default_params(minimal_score=d["minimal_score"], index_num=k, **params)
        for part in range(d["stats"].nparts):
            p.part = part
            p.is_last_index_part = int(k == len(idx) - 1 and part == d["stats"].nparts - 1)
            run.align_part(d["prefix"], d["db"], d["stats"], part, p)
    recs = run.records()
    ctr = run.counters
    out = dict(records=recs, nparts=[d["stats"].nparts for d in idx], num_aligned=ctr.num_aligned, num_short=ctr.num_short,
               per_db=[ctr.reads_matched_per_db[k] for k in range(len(idx))], seqs=seqs)
    run.close()
    for d in idx:
        for ix in d["parts"]:
            ix.free()
    return out



def gpu_run(engine, case, tmpdir, with_cigar=True):
    """The same case through libsmr_hip (C ABI): -> dict(records, num_aligned, num_short, per_db)"""
    g = golden.load()[case]
    idx, seqs = build_case(case, tmpdir)
    params = {k: v for k, v in g["params"].items() if k not in ("max_mb", "evalue", "lnwin")}        # (the seed length is the index's)
    reads = smr.Reads.from_seqs(seqs)
    plist = [smr.default_params(minimal_score=d["minimal_score"], **params) for d in idx]
    slots = 256 if plist[0].num_alignments == 0 else None          # -num_alignments 0 = all: a read may align to every reference
    smr.align(engine, reads, [d["parts"] for d in idx], plist, with_cigar=with_cigar, max_alignments_per_read=slots)
    ctr = engine.counters(len(idx))
    out = dict(records=engine.records(), num_aligned=ctr["num_aligned"], num_short=ctr["num_short"], per_db=ctr["reads_matched_per_db"])
    reads.free()
    for d in idx:
        for ix in d["parts"]:
            ix.free()
    return out
