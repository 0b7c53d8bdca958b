"""Seeded synthetic workloads (no network, /root/reference is absent on the GPU box): an rRNA-like clustered
reference DB and Illumina-like reads, following SURVEY.md 8(d) config 3.  Deterministic for a given numpy."""
import os

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


def make_db(path, total_nt, seed=42, mean_len=1500, family_size=40, sub_lo=0.03, sub_hi=0.10, indel=0.005, min_len=400, tag="fam"):
    """Write a FASTA of ~total_nt nucleotides: families of mutated copies of random ancestors.
    Returns (n_seqs, total_nt_written)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n_written = 0
    n_seqs = 0
    with open(path, "wb") as f:
        fam = 0
        while n_written < total_nt:
            L = int(max(min_len, rng.normal(mean_len, mean_len * 0.07)))
            anc = rng.integers(0, 4, size=L, dtype=np.uint8)
            for m in range(family_size):
                if n_written >= total_nt:
                    break
                s = anc.copy()
                rate = rng.uniform(sub_lo, sub_hi)
                mask = rng.random(L) < rate
                s[mask] = (s[mask] + rng.integers(1, 4, size=int(mask.sum()), dtype=np.uint8)) & 3
                # a few indels
                k = rng.poisson(indel * L)
                for _ in range(k):
                    p = int(rng.integers(0, len(s)))
                    if rng.random() < 0.5:
                        s = np.delete(s, p)
                    else:
                        s = np.insert(s, p, rng.integers(0, 4, dtype=np.uint8))
                f.write(b">%s%d_m%d synthetic rRNA-like\n" % (tag.encode(), fam, m))
                f.write(_ACGT[s].tobytes())
                f.write(b"\n")
                n_written += len(s)
                n_seqs += 1
            fam += 1
    return n_seqs, n_written


def load_db_codes(path):
    """-> (codes uint8 array of all sequences concatenated, offsets int64[n+1])"""
    seqs = []
    with open(path, "rb") as f:
        for line in f:
            if not line.startswith(b">"):
                seqs.append(np.frombuffer(line.rstrip(b"\r\n"), dtype=np.uint8))
    lut = np.zeros(256, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    offs = np.zeros(len(seqs) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    return lut[np.concatenate(seqs)], offs


def make_reads(db_codes, db_offs, n_reads, read_len=150, frac_db=0.10, seed=1234, sub=0.005, indel=0.0001, n_rate=0.001):
    """-> uint8 array (n_reads, read_len) of ASCII letters.  frac_db of the reads are sampled from the DB (either
    strand, with sequencing errors), the rest are uniform random background."""
    rng = np.random.Generator(np.random.PCG64(seed))
    codes = rng.integers(0, 4, size=(n_reads, read_len), dtype=np.uint8)
    from_db = rng.random(n_reads) < frac_db
    idx = np.nonzero(from_db)[0]
    if len(idx):
        nseq = len(db_offs) - 1
        sq = rng.integers(0, nseq, size=len(idx))
        lens = db_offs[sq + 1] - db_offs[sq]
        ok = lens >= read_len
        sq, idx, lens = sq[ok], idx[ok], lens[ok]
        start = db_offs[sq] + (rng.random(len(idx)) * (lens - read_len + 1)).astype(np.int64)
        g = db_codes[start[:, None] + np.arange(read_len)[None, :]]
        rc = rng.random(len(idx)) < 0.5
        g[rc] = _COMP[g[rc][:, ::-1]]
        m = rng.random(g.shape) < sub
        g[m] = (g[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
        # rare indels: shift the tail of the read by one
        has = np.nonzero(rng.random(len(idx)) < indel * read_len)[0]
        for r in has:
            p = int(rng.integers(1, read_len - 1))
            if rng.random() < 0.5:
                g[r, p:-1] = g[r, p + 1:]
            else:
                g[r, p + 1:] = g[r, p:-1].copy()
        codes[idx] = g
    letters = _ACGT[codes]
    nm = rng.random(letters.shape) < n_rate
    letters[nm] = ord("N")
    return letters


def make_reads_fast(db_codes, db_offs, n_reads, read_len=150, frac_db=0.10, seed=1234, sub=0.005, indel=0.0001, n_rate=0.001):
    """The distribution of make_reads at a third of its time (bench.py makes 16 M reads per rank before anything is timed): the N letters
    and the substitutions are placed by drawing their POSITIONS instead of one uniform number per letter.  Not the same reads as
    make_reads for a given seed (the golden fixtures keep using that one)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    codes = rng.integers(0, 4, size=(n_reads, read_len), dtype=np.uint8)
    from_db = rng.random(n_reads) < frac_db
    idx = np.nonzero(from_db)[0]
    if len(idx):
        nseq = len(db_offs) - 1
        sq = rng.integers(0, nseq, size=len(idx))
        lens = db_offs[sq + 1] - db_offs[sq]
        ok = lens >= read_len
        sq, idx, lens = sq[ok], idx[ok], lens[ok]
        start = db_offs[sq] + (rng.random(len(idx)) * (lens - read_len + 1)).astype(np.int64)
        g = db_codes[start[:, None] + np.arange(read_len)[None, :]]
        rc = rng.random(len(idx)) < 0.5
        g[rc] = _COMP[g[rc][:, ::-1]]
        ns = int(rng.binomial(g.size, sub))
        if ns:
            pos = rng.integers(0, g.size, size=ns)
            gf = g.reshape(-1)
            gf[pos] = (gf[pos] + rng.integers(1, 4, size=ns, dtype=np.uint8)) & 3
        has = np.nonzero(rng.random(len(idx)) < indel * read_len)[0]
        for r in has:
            p = int(rng.integers(1, read_len - 1))
            if rng.random() < 0.5:
                g[r, p:-1] = g[r, p + 1:]
            else:
                g[r, p + 1:] = g[r, p:-1].copy()
        codes[idx] = g
    letters = _ACGT[codes]
    nn = int(rng.binomial(letters.size, n_rate))
    if nn:
        letters.reshape(-1)[rng.integers(0, letters.size, size=nn)] = ord("N")
    return letters


def make_long_reads(db_codes, db_offs, n_reads, mean_len=5000, sd_len=500, min_len=1000, max_len=30000, frac_db=1.0, seed=77,
                    ins=0.06, dele=0.04, sub=0.02):
    """PacBio-like reads: target length ~N(mean_len, sd_len) clipped to [min_len, max_len]; a read sampled from the
    DB is a stretch of one DB sequence (either strand; as long as the target when the sequence allows, random letters around it
    otherwise), then 12 % errors (6 % insertions, 4 % deletions, 2 % substitutions).  Returns (blob bytes of ASCII letters, offsets uint64[n+1])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nseq = len(db_offs) - 1
    out = []
    offs = np.zeros(n_reads + 1, dtype=np.uint64)
    for i in range(n_reads):
        T = int(min(max(rng.normal(mean_len, sd_len), min_len), max_len))
        g = rng.integers(0, 4, size=T, dtype=np.uint8)
        if rng.random() < frac_db:
            sq = int(rng.integers(0, nseq))
            L = int(db_offs[sq + 1] - db_offs[sq])
            take = min(L, T)
            a = int(rng.integers(0, L - take + 1))
            piece = db_codes[db_offs[sq] + a:db_offs[sq] + a + take]
            if rng.random() < 0.5:
                piece = _COMP[piece[::-1]]
            b = int(rng.integers(0, T - take + 1))
            g[b:b + take] = piece
        # errors: one pass, per position: deletion / substitution / (letter kept), then possibly an inserted letter after it
        u = rng.random(T)
        keep = u >= dele
        subm = (u >= dele) & (u < dele + sub)
        g = g.copy()
        g[subm] = (g[subm] + rng.integers(1, 4, size=int(subm.sum()), dtype=np.uint8)) & 3
        insm = rng.random(T) < ins
        rep = keep.astype(np.int64) + insm.astype(np.int64)
        res = np.repeat(g, rep)
        # the inserted letters (the second copy of a kept letter, or the only copy of a deleted one) become random letters
        idx_end = np.cumsum(rep)
        ins_pos = idx_end[insm] - 1
        res[ins_pos] = rng.integers(0, 4, size=len(ins_pos), dtype=np.uint8)
        if len(res) < min_len:
            res = np.concatenate([res, rng.integers(0, 4, size=min_len - len(res), dtype=np.uint8)])
        out.append(_ACGT[res])
        offs[i + 1] = offs[i] + np.uint64(len(res))
    return np.concatenate(out).tobytes(), offs


def write_fastq_ragged(path, blob, offs, n=None, first_id=0):
    n = (len(offs) - 1) if n is None else n
    with open(path, "wb") as f:
        for i in range(n):
            a, b = int(offs[i]), int(offs[i + 1])
            f.write(b"@r%d\n" % (first_id + i))
            f.write(blob[a:b])
            f.write(b"\n+\n")
            f.write(b"I" * (b - a))
            f.write(b"\n")


def write_fastq(path, letters, first_id=0):
    n, L = letters.shape
    qual = b"I" * L
    with open(path, "wb") as f:
        for i in range(n):
            f.write(b"@r%d\n" % (first_id + i))
            f.write(letters[i].tobytes())
            f.write(b"\n+\n")
            f.write(qual)
            f.write(b"\n")


def write_fasta(path, letters, first_id=0):
    n, _ = letters.shape
    with open(path, "wb") as f:
        for i in range(n):
            f.write(b">r%d\n" % (first_id + i))
            f.write(letters[i].tobytes())
            f.write(b"\n")


def ensure_db(cache_dir, total_nt, seed=42, **kw):
    os.makedirs(cache_dir, exist_ok=True)
    p = os.path.join(cache_dir, "synth_db_%d_s%d.fasta" % (total_nt, seed))
    if not os.path.isfile(p):
        make_db(p + ".tmp", total_nt, seed=seed, **kw)
        os.replace(p + ".tmp", p)
    return p
