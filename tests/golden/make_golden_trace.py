"""Known-answer vectors for the traceback kernels: seeded random (read, reference window) pairs run through the reference's OWN ssw.c
(oracle/_ref/libssw_ref.so) with ssw_align(..., flag 2, ...), which, when score1 >= filters, also returns the CIGAR of banded_sw
(ssw.c:919-932).  Stored per pair: both sequences, score1 / begin / end positions and the CIGAR operations.

    python tests/golden/make_golden_trace.py        # rewrites tests/golden/trace_pairs.json.gz

The pairs are made to reach every part of the kernels: gapless and single-indel short reads (bands 1..3: 8 lanes per alignment),
several indels (bands 4..7: 16 lanes), long indels and long noisy reads (wide kernel, several strips of 64 diagonals, band
doubling), tiny windows whose band is wider than the window (the reference's `edge` slot lands on the last reference column), and a
scoring scheme with gap_open < gap_ext (the F scan has to iterate)."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ssw as G  # noqa: E402


def ssw_with_cigar(L, read, ref, sc):
    rd = np.frombuffer(read, dtype=np.int8).copy()
    rf = np.frombuffer(ref, dtype=np.int8).copy()
    mat = G.scoring_matrix(sc["match"], sc["mismatch"], sc["score_N"])
    prof = C.c_void_p(L.ssw_init(rd.ctypes.data, len(rd), mat.ctypes.data, 5, 2))
    a = L.ssw_align(prof, rf.ctypes.data, len(rf), sc["gap_open"], sc["gap_ext"], 2, sc["filters"], 0, 0)
    r = a.contents
    out = [int(r.score1), int(r.ref_begin1), int(r.ref_end1), int(r.read_begin1), int(r.read_end1)]
    cig = [int(r.cigar[k]) for k in range(r.cigarLen)] if r.cigar else None
    L.align_destroy(C.byref(a))
    L.init_destroy(C.byref(prof))
    return out, cig


def mutate(rng, read, sub, ins, dele, long_gap=0, long_len=(2, 12)):
    out = []
    q = 0
    m = len(read)
    while q < m:
        u = rng.random()
        if u < sub:
            out.append(int(rng.integers(0, 4))); q += 1
        elif u < sub + ins:
            out += [int(x) for x in rng.integers(0, 4, size=int(rng.integers(long_len[0], long_len[1])) if rng.random() < long_gap else 1)]
        elif u < sub + ins + dele:
            q += int(rng.integers(long_len[0], long_len[1])) if rng.random() < long_gap else 1
        else:
            out.append(int(read[q]) if read[q] < 4 else 0); q += 1
    return out


def make_pairs(seed, kind, n_pairs):
    rng = np.random.Generator(np.random.PCG64(seed))
    pairs = []
    for i in range(n_pairs):
        if kind == "short":          # Illumina-like: mostly gapless, a few single indels
            m = int(rng.integers(30, 260))
            read = rng.integers(0, 4, size=m).astype(np.uint8)
            read[rng.random(m) < 0.01] = 4
            core = mutate(rng, read, 0.02, 0.004 * (i % 3), 0.004 * (i % 4))
        elif kind == "indels":       # several indels, some a few bases long: bands 4..30
            m = int(rng.integers(60, 400))
            read = rng.integers(0, 4, size=m).astype(np.uint8)
            core = mutate(rng, read, 0.03, 0.02, 0.02, long_gap=0.3, long_len=(2, 9))
        elif kind == "long":         # long noisy reads with long gaps: wide kernel, band doubling, several strips
            m = int(rng.integers(700, 3000))
            read = rng.integers(0, 4, size=m).astype(np.uint8)
            read[rng.random(m) < 0.005] = 4
            core = mutate(rng, read, 0.06, 0.03, 0.03, long_gap=0.15, long_len=(5, 90))
        else:                        # tiny windows with large length differences: band wider than the window
            m = int(rng.integers(12, 60))
            read = rng.integers(0, 4, size=m).astype(np.uint8)
            a, b = sorted(int(x) for x in rng.integers(1, m, size=2))
            core = [int(x) for x in read[:a]] + [int(x) for x in rng.integers(0, 4, size=int(rng.integers(0, 40)))] + [int(x) for x in read[b:]] if i % 2 else \
                [int(x) for x in read[:a]] + [int(x) for x in read[a:]] + [int(x) for x in rng.integers(0, 4, size=int(rng.integers(0, 3)))]
            if i % 2 == 0:
                read = np.concatenate([read[:a], rng.integers(0, 4, size=int(rng.integers(5, 45))).astype(np.uint8), read[a:]])
        fl, fr = int(rng.integers(0, 10)), int(rng.integers(0, 10))
        ref = np.array(list(rng.integers(0, 4, size=fl)) + core + list(rng.integers(0, 4, size=fr)), dtype=np.uint8)
        if ref.size == 0:
            ref = np.array([0], dtype=np.uint8)
        pairs.append((read.tobytes(), ref.tobytes()))
    return pairs


SCHEMES = [
    dict(match=2, mismatch=-3, score_N=-3, gap_open=5, gap_ext=2, filters=10),
    dict(match=5, mismatch=-4, score_N=-4, gap_open=5, gap_ext=2, filters=10),
    dict(match=2, mismatch=-3, score_N=-3, gap_open=2, gap_ext=3, filters=10),      # gap_open < gap_ext
    dict(match=3, mismatch=-2, score_N=-1, gap_open=3, gap_ext=3, filters=10),
]
SETS = [("short", 160), ("indels", 120), ("tiny", 160), ("long", 24)]


def main():
    assert os.path.isfile(G.LIB), "make -C oracle ref  (needs /root/reference)"
    L = G.ref_lib()
    tr = bytes.maketrans(bytes(range(5)), b"ACGTN")
    out = {"alphabet": "ACGTN", "cases": []}
    for k, sc in enumerate(SCHEMES):
        for kind, n in SETS:
            if kind == "long" and k >= 2:
                n = 8
            reads, refs, exp, cigs = [], [], [], []
            for r, f in make_pairs(977 * (k + 1) + len(kind), kind, n):
                e, cg = ssw_with_cigar(L, r, f, sc)
                if cg is None:
                    continue                                  # score below the filter: no CIGAR
                reads.append(r.translate(tr).decode()); refs.append(f.translate(tr).decode()); exp.append(e); cigs.append(cg)
            bands = [abs((e[2] - e[1]) - (e[4] - e[3])) + 1 for e in exp]
            out["cases"].append(dict(scoring=sc, kind=kind, reads=reads, refs=refs, expected=exp, cigars=cigs))
            print("scheme %d %-6s pairs %3d  initial band max %3d  ops max %3d  leading-0M %d" % (
                k, kind, len(reads), max(bands), max(len(c) for c in cigs), sum(1 for c in cigs if (c[-1] >> 4) == 0 or (c[0] >> 4) == 0)))
    import gzip
    with gzip.GzipFile(os.path.join(HERE, "trace_pairs.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(out, separators=(",", ":")).encode())
    print(os.path.getsize(os.path.join(HERE, "trace_pairs.json.gz")), "bytes")


if __name__ == "__main__":
    sys.exit(main())
