"""Differential fuzzing of the ORACLE against the REFERENCE, without a GPU: the same seeded random workloads and the same random points of the
option space that tools/fuzz_emu.py runs through the kernel sources (workload shape, seed length, scoring scheme, -N, -num_seeds, -min_lis,
-edges in letters and percent, -no-best / -num_alignments, -F / -R, -full_search, -e, index parts by -m, one or two --ref) go through
oracle/_ref/sortmerna_ref -- the unmodified reference, built from /root/reference by oracle/Makefile -- and through oracle/smr_oracle.c; the
per-read records (Read::toBinString bytes from the reference's KVDB) and the Readstats counters must be equal.  Round 5's fuzzer refereed the
kernels with the oracle, and the oracle was pinned to the reference on 24 fixed cases: this closes the loop over the random option space, so
the emulator campaigns inherit a reference pedigree.  The oracle reads the index files the REFERENCE built (its CMPH ids) and takes the minimal
score from the reference's log, like tests/test_oracle_golden.py's live case.  TEST INFRASTRUCTURE (uses oracle/ and tests/helpers).

Differences to fuzz_emu.py's draw, all forced by the reference's command line: strides are the defaults (its -passes parser never stores them,
options.cpp:704-732), `minoccur` has no option, -min_lis excludes -num_alignments and -no-best (options.cpp:1653-1665), gap_ext <= gap_open
(:1637), -num_alignments needs an output format (-fastx).  Scoring schemes the LIBRARY refuses (gap_open <= gap_ext, 2 gap < |mismatch|,
positive N) are drawn here: the oracle restates ssw.c's stripe geometry and must follow the reference there too.

    python tools/fuzz_ref.py [first_seed [n_cases]]        # one line per case; a differing one with everything needed to repeat it
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from helpers import orc, paths, refrun  # noqa: E402
from sortmerna_amd import synth  # noqa: E402

SCHEMES = [(2, -3, 5, 2), (2, -3, 5, 2), (2, -3, 3, 2), (3, -4, 6, 3), (5, -4, 5, 2), (1, -2, 3, 1), (2, -3, 4, 3), (4, -5, 7, 3), (2, -3, 10, 2), (1, -1, 2, 1),
           (2, -3, 3, 3), (2, -3, 2, 2), (2, -5, 2, 1), (1, -3, 1, 1)]       # the last four: outside what libsmr_hip accepts (ssw.c's stripe effects are part of the answer)


def draw(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]  # noqa: E731
    lnwin = pick([18, 18, 18, 18, 16, 14, 12])
    wk = dict(db_nt=int(pick([40_000, 80_000, 150_000, 300_000])), n_reads=int(pick([200, 400, 700, 33, 12])), read_len=int(pick([40, 75, 100, 150, 150, 220, 301, 301, 600, 1100])),
              frac_db=float(pick([0.2, 0.5, 0.8])), n_rate=float(pick([0.0, 0.002, 0.02, 0.06])), family_size=int(pick([1, 4, 40, 40, 200])), mean_len=int(pick([300, 1500])),
              db_kw=dict(sub_lo=float(pick([0.0, 0.01, 0.03])), sub_hi=float(pick([0.02, 0.06, 0.10])), indel=float(pick([0.0, 0.005, 0.02]))),
              read_kw=dict(sub=float(pick([0.0, 0.005, 0.005, 0.03, 0.08])), indel=float(pick([0.0, 0.0001, 0.002, 0.01]))))
    if wk["read_len"] >= 600:
        wk["n_reads"], wk["mean_len"] = min(wk["n_reads"], 60), 1500
    if wk["db_kw"]["sub_hi"] < wk["db_kw"]["sub_lo"]:
        wk["db_kw"]["sub_hi"] = wk["db_kw"]["sub_lo"] + 0.01
    wk["db_kw"]["min_len"] = min(400, wk["mean_len"])
    match, mismatch, go, ge = pick(SCHEMES)
    cli, params = ["-match", str(match), "-mismatch", str(mismatch), "-gap_open", str(go), "-gap_ext", str(ge)], dict(match=match, mismatch=mismatch, gap_open=go, gap_ext=ge, score_N=mismatch)
    if lnwin != 18:
        cli += ["-L", str(lnwin)]
    params["lnwin"], params["skiplengths"] = lnwin, [lnwin, lnwin // 2, 3]
    n_score = pick([None, None, 0, -1, 1, -min(2 * go, 2 * ge, 127)])
    if n_score is not None:
        cli += ["-N", str(n_score)]
        params["score_N"] = int(n_score)
    ns = int(pick([1, 2, 2, 2, 3, 4]))
    if ns != 2:
        cli += ["-num_seeds", str(ns)]
    params["num_seeds"] = ns
    mode = pick(["best", "best", "min_lis", "num", "num", "nobest", "nobest_num"])
    if mode == "min_lis":
        v = int(pick([1, 3, 4]))
        cli += ["-min_lis", str(v)]
        params["min_lis"] = v
    elif mode in ("num", "nobest_num"):
        v = int(pick([0, 1, 2, 3, 5, 8]))
        cli += ["-num_alignments", str(v)]
        params["num_alignments"] = v
    if mode.startswith("nobest"):
        cli += ["-no-best"]
        params["is_best"] = 0
    if pick([0, 0, 1]):
        cli += ["-full_search"]
        params["is_full_search"] = 1
    if pick([0, 0, 1]):
        v = int(pick([6, 8, 10, 10]))
        cli += ["-edges", "%d%%" % v]
        params["edges"], params["is_as_percent"] = v, 1
    else:
        v = int(pick([1, 2, 4, 4, 10]))
        if v != 4:
            cli += ["-edges", str(v)]
        params["edges"] = v
    fr = pick(["", "", "F", "R"])
    if fr == "F":
        cli += ["-F"]
        params["is_reverse"] = 0
    elif fr == "R":
        cli += ["-R"]
        params["is_forward"] = 0
    ev = pick([None, None, None, "1e-5", "10", "1e-12"])
    if ev:
        cli += ["-e", ev]
    max_mb = pick([None, None, None, 0.4, 0.8])
    if max_mb:
        cli += ["-m", str(max_mb)]
    amb = float(pick([0.0, 0.0, 0.0005, 0.003]))
    return wk, cli, params, lnwin, amb


def make_inputs(seed, wk, amb, tmp):
    db = os.path.join(tmp, "db.fasta")
    synth.make_db(db, wk["db_nt"], seed=seed, family_size=wk["family_size"], mean_len=wk["mean_len"], **wk["db_kw"])
    codes, offs = synth.load_db_codes(db)
    if amb > 0:
        arng = np.random.Generator(np.random.PCG64(seed + 3))
        out = []
        for line in open(db, "rb"):
            if not line.startswith(b">"):
                a = np.frombuffer(line.rstrip(b"\r\n"), dtype=np.uint8).copy()
                m = arng.random(len(a)) < amb
                a[m] = np.frombuffer(b"NNNRYKMSWn", dtype=np.uint8)[arng.integers(0, 10, int(m.sum()))]
                line = a.tobytes() + b"\n"
            out.append(line)
        open(db, "wb").write(b"".join(out))
    letters = synth.make_reads(codes, offs, wk["n_reads"], read_len=wk["read_len"], frac_db=wk["frac_db"], seed=seed + 1, n_rate=wk["n_rate"], **wk["read_kw"])
    seqs = [bytes(x).decode() for x in letters]
    rng = np.random.Generator(np.random.PCG64(seed + 2))
    for i in range(0, len(seqs), 17):
        seqs[i] = seqs[i][: int(rng.integers(12, wk["read_len"]))]
    if len(seqs) > 10:
        seqs[3], seqs[9] = seqs[3][:12], seqs[9][:18]
    reads = os.path.join(tmp, "reads.fasta")
    with open(reads, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">r%d\n%s\n" % (i, s))
    return db, reads, seqs


def second_db(db, tmp, seed):
    """the first DB's sequences with 3 % of their letters changed and a fifth of them dropped (fuzz_emu.py's second DB)"""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0xDB2))
    out, keep = [], True
    for line in open(db, "rb"):
        if line.startswith(b">"):
            keep = rng.random() < 0.8
            if keep:
                out.append(line)
        elif keep:
            a = np.frombuffer(line.rstrip(b"\r\n"), dtype=np.uint8).copy()
            m = rng.random(len(a)) < 0.03
            a[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(m.sum()))]
            out.append(a.tobytes() + b"\n")
    db2 = os.path.join(tmp, "second.fasta")
    open(db2, "wb").write(b"".join(out))
    return db2


def one_case(seed, tmp):
    """-> (ok, line)"""
    t = time.time()
    wk, cli, params, lnwin, amb = draw(seed)
    db, reads, seqs = make_inputs(seed, wk, amb, tmp)
    dbs = [db]
    if seed % 4 == 3:
        dbs.append(second_db(db, tmp, seed))
        if seed % 8 == 7:
            dbs.reverse()
    idx = os.path.join(tmp, "idx")
    res = refrun.run_reference(dbs, [reads], os.path.join(tmp, "wd"), extra=cli + ["-fastx", "-v"], threads=1, idx_dir=idx, timeout=3000)
    if res.rc != 0 or not res.log.get("minimal_score"):
        return None, "seed %d: the reference refused the case (rc %d): %s   options %s" % (seed, res.rc, res.stdout.strip().splitlines()[-1][:140] if res.stdout.strip() else "", " ".join(cli))
    run = orc.Run(seqs)
    nparts = []
    for k, d in enumerate(dbs):
        prefix = refrun.index_prefix_for(idx, d)
        st = orc.load_stats(prefix)
        nparts.append(st.nparts)
        p = orc.default_params(minimal_score=res.log["minimal_score"][k], index_num=k, **params)
        for part in range(st.nparts):
            p.part, p.is_last_index_part = part, int(k == len(dbs) - 1 and part == st.nparts - 1)
            run.align_part(prefix, d, st, part, p)
    recs = run.records()
    exp = [res.kvdb.get(b"0_%d" % i, b"") for i in range(len(seqs))]
    rs = [v for k, v in res.kvdb.items() if b"_" not in k]
    stats = refrun.parse_readstats(rs[0]) if rs else {}
    per_db = [int(run.counters.reads_matched_per_db[k]) for k in range(len(dbs))]
    bad = [i for i in range(len(seqs)) if recs[i] != exp[i]]
    same_ctr = run.counters.num_aligned == res.log["num_aligned"] and (not stats or (stats["reads_matched_per_db"][:len(dbs)] == per_db and stats["num_short"] == run.counters.num_short))
    n_al = run.counters.num_aligned
    run.close()
    ok = not bad and same_ctr
    line = "seed %d %s: %d reads, %d aligned%s, parts %s, min score %s, %.1f s   %s" % (
        seed, "ok" if ok else "DIFFERS", len(seqs), n_al, " (two DBs: %s)" % per_db if len(dbs) > 1 else "", nparts, res.log["minimal_score"], time.time() - t, " ".join(cli))
    if not ok:
        line += "\n    records differing %d (first read %s)  counters oracle %d / %s reference %s / %s\n    workload %s ambiguous %g\n    oracle  %s\n    reference %s" % (
            len(bad), bad[:1], n_al, per_db, res.log["num_aligned"], stats.get("reads_matched_per_db"), wk, amb,
            refrun.parse_record(recs[bad[0]]) if bad else "", refrun.parse_record(exp[bad[0]]) if bad else "")
    return ok, line


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    assert paths.have_reference() and paths.have_ref_bin(), "needs /root/reference and oracle/_ref/sortmerna_ref (make -C oracle ref)"
    bad = refused = 0
    for seed in range(first, first + n):
        with tempfile.TemporaryDirectory(prefix="smr_fuzzref_") as tmp:
            try:
                ok, line = one_case(seed, tmp)
            except Exception as x:  # noqa: BLE001
                ok, line = False, "seed %d ERROR %s: %s" % (seed, type(x).__name__, x)
        print(line, flush=True)
        refused += ok is None
        bad += ok is False
    print("%d case(s), %d refused by the reference, %d differing or failing" % (n, refused, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
