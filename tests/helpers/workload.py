"""Seeded synthetic workloads shared by the GPU parity tests, smoke() and tools.  Everything is generated on the
machine the test runs on (the GPU box has no /root/reference)."""
import os

import numpy as np

import sortmerna_amd as smr
from sortmerna_amd import synth

from . import orc

# Gumbel (lambda, K) for match 2 / mismatch -3 / gap 5,2 as computed by the reference's ALP
# (refstats.cpp:194-233) for near-uniform background frequencies (reference run in the build container,
# see tests/golden/README.md).  They only feed minimal_score; parity tests pass the same value to both sides.
GUMBEL_UNIFORM = (0.618874, 0.343238)


class Workload:
    def __init__(self, tmpdir, db_nt=300_000, n_reads=4000, read_len=150, frac_db=0.4, seed=5, max_mb=3072.0,
                 n_rate=0.002, family_size=40, lnwin=18, mean_len=1500, db_kw=None, db_fasta=None, seqs=None, read_kw=None, db_ambiguous=0.0):
        """db_fasta + seqs: a DB file and reads the test wrote itself (a crafted case) instead of the seeded synthetic ones"""
        self.lnwin = lnwin
        self.dir = tmpdir
        if db_fasta is not None:
            self.db = db_fasta
            self.seqs = list(seqs)
        else:
            self.db = os.path.join(tmpdir, "db_%d_%d.fasta" % (db_nt, seed))
            synth.make_db(self.db, db_nt, seed=seed, family_size=family_size, mean_len=mean_len, **(db_kw or {}))
            codes, offs = synth.load_db_codes(self.db)
            if db_ambiguous > 0:                 # a fraction of the reference letters become IUPAC ambiguity codes / lower case (the reads are sampled from the clean text)
                arng = np.random.Generator(np.random.PCG64(seed + 3))
                out = []
                for line in open(self.db, "rb"):
                    if not line.startswith(b">"):
                        a = np.frombuffer(line.rstrip(b"\r\n"), dtype=np.uint8).copy()
                        m = arng.random(len(a)) < db_ambiguous
                        a[m] = np.frombuffer(b"NNNRYKMSWn", dtype=np.uint8)[arng.integers(0, 10, int(m.sum()))]
                        lo = arng.random(len(a)) < db_ambiguous
                        a[lo] = np.frombuffer(bytes(a[lo]).lower(), dtype=np.uint8)
                        line = a.tobytes() + b"\n"
                    out.append(line)
                open(self.db, "wb").write(b"".join(out))
            self.letters = synth.make_reads(codes, offs, n_reads, read_len=read_len, frac_db=frac_db, seed=seed + 1, n_rate=n_rate, **(read_kw or {}))
            # ragged lengths, a too-short read and an empty read to cover the edge cases
            self.seqs = [bytes(x).decode() for x in self.letters]
            rng = np.random.Generator(np.random.PCG64(seed + 2))
            for i in range(0, n_reads, 17):
                self.seqs[i] = self.seqs[i][: int(rng.integers(lnwin, read_len))]
            if n_reads > 10:
                self.seqs[3] = self.seqs[3][:12]
                self.seqs[7] = ""
                self.seqs[9] = self.seqs[9][:lnwin]
        self.parts = smr.Index.build(self.db, lnwin, max_mb, 10000, 0)
        self.prefix = os.path.join(tmpdir, "idx_%d_%d" % (db_nt, seed))
        smr.Index.write_files(self.parts, self.db, self.prefix)
        self.stats = orc.load_stats(self.prefix)
        self.reads = smr.Reads.from_seqs(self.seqs)
        lam, K = GUMBEL_UNIFORM
        self.minimal_score = smr.minimal_score(lam, K, self.parts[0].info(), len(self.seqs), sum(map(len, self.seqs)))

    def oracle_records(self, **kw):
        """Run the CPU oracle over all parts; returns (records, counters)."""
        kw = dict(kw)
        p = orc.default_params(minimal_score=kw.pop("minimal_score", self.minimal_score), **kw)
        if "skiplengths" not in kw:
            p.lnwin = self.lnwin
            p.skiplengths[0], p.skiplengths[1], p.skiplengths[2] = self.lnwin, self.lnwin // 2, 3      # refstats.cpp:159-166
        run = orc.Run(self.seqs)
        for part in range(self.stats.nparts):
            p.part = part
            p.is_last_index_part = int(part == self.stats.nparts - 1)
            run.align_part(self.prefix, self.db, self.stats, part, p)
        recs = run.records()
        ctr = run.counters
        out = dict(num_aligned=ctr.num_aligned, num_short=ctr.num_short, per_db=ctr.reads_matched_per_db[0],
                   n_windows=ctr.n_windows, n_lookup=ctr.n_lookup, n_node=ctr.n_node, n_entry=ctr.n_entry, n_hit=ctr.n_hit,
                   n_sw_fwd=ctr.n_sw_fwd, n_sw_rev=ctr.n_sw_rev)
        run.close()
        return recs, out

    def gpu_records(self, engine, with_cigar=True, **kw):
        kw = dict(kw)
        p = smr.default_params(minimal_score=kw.pop("minimal_score", self.minimal_score), **kw)
        smr.align(engine, self.reads, [self.parts], [p], with_cigar=with_cigar)
        return engine.records(), engine.counters(1)


def iseq_for_strand(seq, strand):
    """Read in the 0..3 alphabet as the reference sees it at the start of a strand with no prior SW:
    forward N->0; reverse-complement of that (N ends up as 3)  (read.cpp:334-357)."""
    m = {"A": 0, "C": 1, "G": 2, "T": 3, "U": 3, "a": 0, "c": 1, "g": 2, "t": 3, "u": 3}
    v = np.array([m.get(c, 0) for c in seq], dtype=np.uint8)
    if strand:
        v = (3 - v[::-1]).astype(np.uint8)
    return v
