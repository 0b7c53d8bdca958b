"""Generate the golden fixtures of tests/golden/ by running the UNMODIFIED reference (oracle/_ref/sortmerna_ref,
built from /root/reference by oracle/Makefile) in the build container.  The GPU box and CI have no /root/reference:
they consume the committed outputs of this script.

    python tests/golden/make_golden.py          # rewrites tests/golden/*

What is produced
  t0_read.fasta / t0_ref.fasta     the reference's bundled data/test_read.fasta (+ trailing newline) and
                                   data/test_ref.fasta: input of its tests t0/t2 (scripts/test.jinja:132-168,247-266)
  t9_reads.fasta / t9_ref.fasta    data/illumina_GQ099317.fasta, data/ref_GQ099317_forward_and_rc.fasta (test t9, :425-478)
  syn_db.fasta / syn_reads.fasta   a small seeded synthetic rRNA-like DB (families of mutated copies) and 150-nt reads
                                   (half sampled from the DB, ragged / too short / empty / N-containing records included)
  golden.json                      per case: command-line options, Gumbel lambda/K, minimal_score, Readstats counters,
                                   BLAST/SAM rows, and the name of the records file
  <case>.records.bin               the reference's own per-read KVDB values (Read::toBinString bytes, read.cpp:429-462):
                                   u32 n, then n x (u32 len, bytes), in read order (len 0 = read has no record)
"""
import json
import os
import shutil
import struct
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)

from helpers import paths, refrun  # noqa: E402
from sortmerna_amd import synth  # noqa: E402

import numpy as np  # noqa: E402

# (case name, extra reference options, the same options in smr_params / orc_params vocabulary)
SYN_CASES = [
    ("syn_default", [], {}),
    ("syn_all", ["-num_alignments", "0"], {"num_alignments": 0}),
    ("syn_best3", ["-num_alignments", "3"], {"num_alignments": 3}),
    ("syn_nobest2", ["-no-best", "-num_alignments", "2"], {"is_best": 0, "num_alignments": 2}),
    ("syn_F", ["-F"], {"is_reverse": 0}),
    ("syn_R", ["-R"], {"is_forward": 0}),
    ("syn_full_search", ["-full_search"], {"is_full_search": 1}),
    ("syn_seeds3_edges10", ["-num_seeds", "3", "-edges", "10"], {"num_seeds": 3, "edges": 10}),
    # no "-passes a,b,c" case: the reference's parser of that option (options.cpp:704-732) never stores the three
    # strides per index (it emplaces vectors of that SIZE), so the run silently uses the defaults {L, L/2, 3}.
    ("syn_multipart", ["-m", "0.15"], {"max_mb": 0.15}),
    # round 4: the rest of the option space of the hot path (added with `make_golden.py --only ...`: the cases above were not re-run)
    ("syn_seeds1", ["-num_seeds", "1"], {"num_seeds": 1}),
    ("syn_minlis3", ["-min_lis", "3"], {"min_lis": 3}),        # (the reference refuses -min_lis together with -num_alignments, options.cpp:1656)
    ("syn_minlis1", ["-min_lis", "1"], {"min_lis": 1}),
    ("syn_N0", ["-N", "0"], {"score_N": 0}),
    ("syn_gaps32", ["-gap_open", "3", "-gap_ext", "2"], {"gap_open": 3, "gap_ext": 2}),
    # ... a seed length of 14 (its own index: passes L, L/2, 3 -- options.cpp), another scoring scheme (its own Gumbel lambda / K and minimal
    # score), the (inert) -passes option, a stricter E-value ("evalue" is not a Params field: the tests take the minimal score computed from it)
    ("syn_L14", ["-L", "14"], {"lnwin": 14, "skiplengths": [14, 7, 3]}),
    ("syn_score3463", ["-match", "3", "-mismatch", "-4", "-gap_open", "6", "-gap_ext", "3"], {"match": 3, "mismatch": -4, "score_N": -4, "gap_open": 6, "gap_ext": 3}),      # (without -N the N penalty IS the mismatch: options.cpp:1707-1708)
    # `-passes 18,6,2` does NOT reach the hot path in the reference: opt_passes (options.cpp:704-732) does skiplengths.emplace_back(18) on a
    # vector<vector<uint32_t>> -- a vector of 18 zeros --, never parses the last number, and refstats.cpp:159-165 then sees zeros and installs the
    # defaults L, L/2, 3.  The golden pins exactly that: the records equal the default run's.
    ("syn_passes1862", ["-passes", "18,6,2"], {}),
    ("syn_e1em8", ["-e", "1e-8"], {"evalue": 1e-8}),
]


def write_records(path, recs):
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(recs)))
        for r in recs:
            f.write(struct.pack("<I", len(r)))
            f.write(r)


def records_in_read_order(kvdb, n_reads):
    out = []
    for i in range(n_reads):
        out.append(kvdb.get(b"0_%d" % i, b""))
    return out


def run_case(name, refs, reads, n_reads, extra, tmp, with_reports=True):
    wd = os.path.join(tmp, name)
    ex = list(extra) + ["-v"]
    if with_reports:
        ex += ["-sam", "-blast", "1 qstrand cigar", "-fastx", "-other"]
    res = refrun.run_reference(refs, reads, wd, extra=ex, threads=1)
    assert res.rc == 0, res.stdout[-2000:]
    recs = records_in_read_order(res.kvdb, n_reads)
    write_records(os.path.join(HERE, name + ".records.bin"), recs)
    rs = [v for k, v in res.kvdb.items() if b"_" not in k]
    stats = refrun.parse_readstats(rs[0]) if rs else {}
    out = dict(options=extra, log=res.log, readstats=stats, records=name + ".records.bin",
               n_records=sum(1 for r in recs if r))
    outd = os.path.join(wd, "out")
    for fn, key in (("aligned.blast", "blast"), ("aligned.sam", "sam")):
        p = os.path.join(outd, fn)
        if os.path.isfile(p):
            out[key] = [l.rstrip("\r\n") for l in open(p)]
    for fn, key in (("aligned.fa", "aligned_ids"), ("other.fa", "other_ids"), ("aligned.fq", "aligned_ids"), ("other.fq", "other_ids")):
        p = os.path.join(outd, fn)
        if os.path.isfile(p):
            out[key] = [l.split()[0][1:] for l in open(p) if l[:1] in ">@" and not l.startswith("@I")]
    nparts = 0
    for pfx in res.idx_prefixes:
        nparts = max(nparts, len([1 for f in os.listdir(os.path.dirname(pfx)) if f.startswith(os.path.basename(pfx) + ".kmer_")]))
    out["index_parts"] = nparts
    return out


def only(names):
    """`make_golden.py --only case,case`: run the reference for these synthetic cases alone (committed syn_db.fasta / syn_reads.fasta) and add them
    to golden.json; nothing else is rewritten"""
    tmp = tempfile.mkdtemp(prefix="golden_")
    G = json.load(open(os.path.join(HERE, "golden.json")))
    db, reads = os.path.join(HERE, "syn_db.fasta"), os.path.join(HERE, "syn_reads.fasta")
    n = sum(1 for l in open(reads) if l.startswith(">"))
    for name, extra, params in SYN_CASES:
        if name not in names:
            continue
        G[name] = run_case(name, [db], [reads], n, extra, tmp, with_reports=False)
        G[name]["params"] = params
        print(name, "aligned", G[name]["log"].get("num_aligned"), "records", G[name]["n_records"], "parts", G[name]["index_parts"])
    json.dump(G, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
    shutil.rmtree(tmp, ignore_errors=True)


def main():
    assert paths.have_reference() and paths.have_ref_bin(), "needs /root/reference and oracle/_ref/sortmerna_ref (make -C oracle)"
    if len(sys.argv) > 2 and sys.argv[1] == "--only":
        return only(sys.argv[2].split(","))
    tmp = tempfile.mkdtemp(prefix="golden_")
    G = {}
    # ---- the reference's own test inputs ----
    t0_read = os.path.join(HERE, "t0_read.fasta")
    with open(t0_read, "wb") as f:
        f.write(open(os.path.join(paths.REF_DATA, "test_read.fasta"), "rb").read().rstrip(b"\r\n") + b"\n")
    shutil.copyfile(os.path.join(paths.REF_DATA, "test_ref.fasta"), os.path.join(HERE, "t0_ref.fasta"))
    shutil.copyfile(os.path.join(paths.REF_DATA, "illumina_GQ099317.fasta"), os.path.join(HERE, "t9_reads.fasta"))
    shutil.copyfile(os.path.join(paths.REF_DATA, "ref_GQ099317_forward_and_rc.fasta"), os.path.join(HERE, "t9_ref.fasta"))
    for p in ("t0_ref.fasta", "t9_reads.fasta", "t9_ref.fasta"):
        os.chmod(os.path.join(HERE, p), 0o644)
    G["t0"] = run_case("t0", [os.path.join(HERE, "t0_ref.fasta")], [t0_read], 1, [], tmp)
    G["t0"]["params"] = {}
    G["t9"] = run_case("t9", [os.path.join(HERE, "t9_ref.fasta")], [os.path.join(HERE, "t9_reads.fasta")], 1,
                       ["-num_alignments", "0", "-mismatch", "-3"], tmp)
    G["t9"]["params"] = {"num_alignments": 0}
    # ---- synthetic workload ----
    db = os.path.join(HERE, "syn_db.fasta")
    synth.make_db(db, 60_000, seed=11, family_size=8, mean_len=1200)
    codes, offs = synth.load_db_codes(db)
    letters = synth.make_reads(codes, offs, 180, read_len=150, frac_db=0.5, seed=12, n_rate=0.003)
    noisy = synth.make_reads(codes, offs, 180, read_len=150, frac_db=0.8, seed=14, sub=0.08, indel=0.006, n_rate=0.003)
    seqs = [bytes(x).decode() for x in letters] + [bytes(x).decode() for x in noisy]
    rng = np.random.Generator(np.random.PCG64(13))
    for i in range(0, len(seqs), 13):
        seqs[i] = seqs[i][: int(rng.integers(18, 150))]
    seqs[3] = seqs[3][:12]          # shorter than the seed
    seqs[9] = seqs[9][:18]          # exactly one window
    seqs[21] = seqs[21].lower()     # lower case is accepted (common.hpp:68-77)
    seqs[22] = seqs[22].replace("T", "U")
    reads = os.path.join(HERE, "syn_reads.fasta")
    with open(reads, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">r%d\n%s\n" % (i, s))
    for name, extra, params in SYN_CASES:
        G[name] = run_case(name, [db], [reads], len(seqs), extra, tmp, with_reports=(name in ("syn_default", "syn_all")))
        G[name]["params"] = params
        print(name, "aligned", G[name]["log"].get("num_aligned"), "records", G[name]["n_records"], "parts", G[name]["index_parts"])
    # ---- real data: a slice of the bundled silva-arc-16s-id95 DB (IUPAC letters, long headers) and set4 reads that hit it ----
    real_db = os.path.join(HERE, "real_db.fasta")
    src_db = os.path.join(paths.REF_DATA, "rRNA_databases", "silva-arc-16s-id95.fasta")
    with open(src_db) as f, open(real_db, "w") as g:
        nseq = 0
        for line in f:
            if line.startswith(">"):
                nseq += 1
                if nseq > 48:
                    break
            g.write(line)
    from helpers import fastx
    allr = fastx.read_fastx(os.path.join(paths.REF_DATA, "set4_mate_pairs_metatranscriptomics_1.fastq"))
    tmp_reads = os.path.join(tmp, "set4_all.fasta")
    with open(tmp_reads, "w") as f:
        for i, (h, sq, _) in enumerate(allr):
            f.write(">q%d\n%s\n" % (i, sq))
    res = refrun.run_reference([real_db], [tmp_reads], os.path.join(tmp, "real_pick"), extra=["-v"], threads=1)
    assert res.rc == 0, res.stdout[-1500:]
    hit = [i for i in range(len(allr)) if res.kvdb.get(b"0_%d" % i)]
    miss = [i for i in range(len(allr)) if not res.kvdb.get(b"0_%d" % i)]
    pick = sorted(hit[:260] + miss[:90])
    real_reads = os.path.join(HERE, "real_reads.fasta")
    with open(real_reads, "w") as f:
        for i in pick:
            f.write(">%s\n%s\n" % (allr[i][0][1:].split()[0], allr[i][1]))
    for name, extra, params in (("real_default", [], {}), ("real_all", ["-num_alignments", "0"], {"num_alignments": 0})):
        G[name] = run_case(name, [real_db], [real_reads], len(pick), extra, tmp, with_reports=(name == "real_default"))
        G[name]["params"] = params
        print(name, "aligned", G[name]["log"].get("num_aligned"), "records", G[name]["n_records"])
    # ---- two reference DBs in one run (index_num 0 and 1; per-read state carried across indexes) ----
    both = os.path.join(tmp, "both_reads.fasta")
    with open(both, "w") as f:
        f.write(open(reads).read())
        f.write(open(real_reads).read())
    shutil.copyfile(both, os.path.join(HERE, "two_db_reads.fasta"))
    n_both = len(seqs) + len(pick)
    for name, extra, params in (("two_db_default", [], {}), ("two_db_all", ["-num_alignments", "0"], {"num_alignments": 0})):
        G[name] = run_case(name, [db, real_db], [os.path.join(HERE, "two_db_reads.fasta")], n_both, extra, tmp, with_reports=False)
        G[name]["params"] = params
        print(name, "aligned", G[name]["log"].get("num_aligned"), "per db", G[name]["readstats"].get("reads_matched_per_db"))
    json.dump(G, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
    shutil.rmtree(tmp, ignore_errors=True)
    print("t0 blast:", G["t0"]["blast"])
    print("t9 sam:", [l for l in G["t9"]["sam"] if not l.startswith("@")])


if __name__ == "__main__":
    main()
