"""Known-answer vectors for the Smith-Waterman kernels at the ssw.h seam: seeded random (read, reference window) pairs run through
the reference's OWN ssw.c (oracle/_ref/libssw_ref.so = /root/reference/src/sortmerna/ssw.c compiled as it lies, oracle/Makefile),
called exactly like alignment.cpp:362-383 does: ssw_init(read, len, Read::initScoringMatrix's 5x5 matrix, 5, score_size 2) and
ssw_align(profile, ref, refLen, gap_open, gap_ext, flag 2, filters, 0, 0).

    python tests/golden/make_golden_ssw.py        # rewrites tests/golden/ssw_pairs.json

Pairs: lengths 1..900 (a few up to 2500), ~1.5 % N in the read, reference window = a mutated copy of the read (substitutions,
insertions, deletions incl. long ones, N) inside random flanks, or an unrelated sequence; two scoring schemes."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(REPO, "oracle", "_ref", "libssw_ref.so")


class SAlign(C.Structure):      # include/ssw.h:58-71
    _fields_ = [("cigar", C.POINTER(C.c_uint32)), ("ref_num", C.c_uint32), ("ref_begin1", C.c_int32), ("ref_end1", C.c_int32),
                ("read_begin1", C.c_int32), ("read_end1", C.c_int32), ("readlen", C.c_uint32), ("score1", C.c_uint16), ("part", C.c_uint16),
                ("index_num", C.c_uint16), ("cigarLen", C.c_uint16), ("strand", C.c_bool)]


def ref_lib():
    L = C.CDLL(LIB)
    L.ssw_init.restype = C.c_void_p
    L.ssw_init.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int8]
    L.ssw_align.restype = C.POINTER(SAlign)
    L.ssw_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint16, C.c_int32, C.c_int32]
    L.init_destroy.argtypes = [C.POINTER(C.c_void_p)]
    L.align_destroy.argtypes = [C.POINTER(C.POINTER(SAlign))]
    return L


def scoring_matrix(match, mismatch, score_n):          # Read::initScoringMatrix, read.cpp:274-288
    m = []
    for a in range(4):
        m += [match if a == b else mismatch for b in range(4)] + [score_n]
    m += [score_n] * 5
    return np.array(m, dtype=np.int8)


def ssw_reference(L, read, ref, match, mismatch, score_n, go, ge, filters):
    """-> [score1, ref_begin1, ref_end1, read_begin1, read_end1] of the reference's ssw_align (flag 2)"""
    rd = np.frombuffer(read, dtype=np.int8).copy()
    rf = np.frombuffer(ref, dtype=np.int8).copy()
    mat = scoring_matrix(match, mismatch, score_n)
    prof = C.c_void_p(L.ssw_init(rd.ctypes.data, len(rd), mat.ctypes.data, 5, 2))
    a = L.ssw_align(prof, rf.ctypes.data, len(rf), go, ge, 2, filters, 0, 0)
    r = a.contents
    out = [int(r.score1), int(r.ref_begin1), int(r.ref_end1), int(r.read_begin1), int(r.read_end1)]
    L.align_destroy(C.byref(a))
    L.init_destroy(C.byref(prof))
    return out


def make_pairs(seed, n_pairs):
    rng = np.random.Generator(np.random.PCG64(seed))
    pairs = []
    for i in range(n_pairs):
        big = i % 40 == 39
        m = int(rng.integers(900, 2500)) if big else int(rng.integers(1, 900)) if i % 3 else int(rng.integers(18, 160))
        read = rng.integers(0, 4, size=m).astype(np.uint8)
        read[rng.random(m) < 0.015] = 4
        kind = i % 5
        if kind == 4:                                        # unrelated reference
            ref = rng.integers(0, 4, size=max(1, m + int(rng.integers(-10, 30)))).astype(np.uint8)
        else:
            sub = [0.01, 0.05, 0.12, 0.2][kind]
            out = []
            q = 0
            while q < m:
                u = rng.random()
                if u < sub:
                    out.append(int(rng.integers(0, 4))); q += 1
                elif u < sub + 0.01:
                    out.append(int(rng.integers(0, 4)))      # insertion in the reference
                elif u < sub + 0.02:
                    q += 1 if rng.random() < 0.7 else int(rng.integers(2, 12))      # deletion (sometimes long)
                elif u < sub + 0.025:
                    out.append(4); q += 1                    # N in the reference
                else:
                    out.append(int(read[q]) if read[q] < 4 else 0); q += 1
            fl, fr = int(rng.integers(0, 12)), int(rng.integers(0, 12))
            ref = np.array(list(rng.integers(0, 4, size=fl)) + out + list(rng.integers(0, 4, size=fr)), dtype=np.uint8)
            if ref.size == 0:
                ref = np.array([0], dtype=np.uint8)
        pairs.append((read.tobytes(), ref.tobytes()))
    return pairs


SCHEMES = [dict(match=2, mismatch=-3, score_N=-3, gap_open=5, gap_ext=2, filters=30), dict(match=5, mismatch=-4, score_N=-4, gap_open=5, gap_ext=2, filters=60)]


# Schemes under which ssw.c's striped kernels leave the affine recurrence (round 6: the library's slow path reproduces their stripe geometry,
# smr_sw_striped.hpp): gap_open <= gap_ext (the 16-bit kernel's early exit from its lazy-F loop, ssw.c:496-507), 2 * gap < |mismatch| (E stored
# before the lazy-F loop has raised H, ssw.c:267), a positive score for N.  -> ssw_pairs_striped.json (make_golden_ssw.py --striped)
STRIPED_SCHEMES = [dict(match=2, mismatch=-3, score_N=-3, gap_open=3, gap_ext=3, filters=30), dict(match=2, mismatch=-3, score_N=-3, gap_open=2, gap_ext=2, filters=30),
                   dict(match=2, mismatch=-5, score_N=-5, gap_open=2, gap_ext=1, filters=30), dict(match=1, mismatch=-3, score_N=-3, gap_open=1, gap_ext=1, filters=20),
                   dict(match=2, mismatch=-3, score_N=1, gap_open=5, gap_ext=2, filters=30), dict(match=2, mismatch=-3, score_N=-3, gap_open=5, gap_ext=2, filters=30)]


def main():
    assert os.path.isfile(LIB), "make -C oracle ref  (needs /root/reference)"
    L = ref_lib()
    out = {"alphabet": "ACGTN", "cases": []}
    if "--striped" in sys.argv:
        for k, sc in enumerate(STRIPED_SCHEMES):
            pairs = make_pairs(20261001 + k, 100)
            exp = [ssw_reference(L, r, f, sc["match"], sc["mismatch"], sc["score_N"], sc["gap_open"], sc["gap_ext"], sc["filters"]) for r, f in pairs]
            tr = bytes.maketrans(bytes(range(5)), b"ACGTN")
            out["cases"].append(dict(scoring=sc, reads=[r.translate(tr).decode() for r, _ in pairs], refs=[f.translate(tr).decode() for _, f in pairs], expected=exp))
            print("scheme", sc, "pairs", len(pairs), "with begin", sum(1 for e in exp if e[1] >= 0), "max score", max(e[0] for e in exp))
        json.dump(out, open(os.path.join(HERE, "ssw_pairs_striped.json"), "w"))
        print(os.path.getsize(os.path.join(HERE, "ssw_pairs_striped.json")), "bytes")
        return 0
    for k, sc in enumerate(SCHEMES):
        pairs = make_pairs(20260926 + k, 160)
        exp = [ssw_reference(L, r, f, sc["match"], sc["mismatch"], sc["score_N"], sc["gap_open"], sc["gap_ext"], sc["filters"]) for r, f in pairs]
        tr = bytes.maketrans(bytes(range(5)), b"ACGTN")
        out["cases"].append(dict(scoring=sc, reads=[r.translate(tr).decode() for r, _ in pairs], refs=[f.translate(tr).decode() for _, f in pairs], expected=exp))
        print("scheme", k, "pairs", len(pairs), "with begin", sum(1 for e in exp if e[1] >= 0), "max score", max(e[0] for e in exp))
    json.dump(out, open(os.path.join(HERE, "ssw_pairs.json"), "w"))
    print(os.path.getsize(os.path.join(HERE, "ssw_pairs.json")), "bytes")


if __name__ == "__main__":
    sys.exit(main())
