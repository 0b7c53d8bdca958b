"""tests/golden/trace_pairs.json.gz: (read, reference window) pairs with the alignment (score, begin/end) and the CIGAR that the
reference's own ssw_align / banded_sw returned for them (written by tests/golden/make_golden_trace.py).  TEST INFRASTRUCTURE."""
import gzip
import json
import os

import numpy as np

from . import paths

_TR = bytes.maketrans(b"ACGTN", bytes(range(5)))


def load():
    with gzip.open(os.path.join(paths.REPO, "tests", "golden", "trace_pairs.json.gz"), "rb") as f:
        g = json.loads(f.read().decode())
    return g["cases"]


def cigar_text(c):
    return "".join("%d%s" % (int(x) >> 4, "MID"[int(x) & 15]) for x in c)


def check(engine, kinds=None, max_pairs=None, schemes=None):
    """the traceback kernels through smr_cigar_batch against banded_sw's CIGARs; returns the number of pairs checked"""
    n = 0
    for ci, c in enumerate(load()):
        if kinds is not None and c["kind"] not in kinds:
            continue
        sc = c["scoring"]
        if schemes is not None and ci // 4 not in schemes:
            continue
        k = len(c["reads"]) if max_pairs is None else min(max_pairs, len(c["reads"]))
        reads, refs, scores = [], [], []
        for rd, rf, e in zip(c["reads"][:k], c["refs"][:k], c["expected"][:k]):
            reads.append(rd.encode().translate(_TR)[e[3]:e[4] + 1])
            refs.append(rf.encode().translate(_TR)[e[1]:e[2] + 1])
            scores.append(e[0])
        got = engine.cigar_batch(reads, refs, scores, match=sc["match"], mismatch=sc["mismatch"], score_N=sc["score_N"], gap_open=sc["gap_open"], gap_ext=sc["gap_ext"])
        for i in range(k):
            exp = np.array(c["cigars"][i], dtype=np.uint32)
            assert got[i].tolist() == exp.tolist(), "scoring %s, %s pair %d (read span %d, reference span %d, score %d): got %s, banded_sw %s" % (
                sc, c["kind"], i, len(reads[i]), len(refs[i]), scores[i], cigar_text(got[i]), cigar_text(exp))
            # a CIGAR consumes exactly both spans
            assert sum(int(x) >> 4 for x in exp if int(x) & 15 in (0, 1)) == len(reads[i]) and sum(int(x) >> 4 for x in exp if int(x) & 15 in (0, 2)) == len(refs[i])
        n += k
    return n


def check_variants(engine):
    """the same vectors through the code paths that ordinary inputs do not reach: DP rows of the wide kernel in global memory (bands too
    wide for LDS) and a CIGAR pool that has to be regrown several times"""
    import os
    n = 0
    try:
        os.environ["SMR_TRACE_GLOBAL_ROWS"] = "1"
        n += check(engine, kinds=["indels"], max_pairs=40)
        n += check(engine, kinds=["long"], max_pairs=2, schemes=[0, 2])
        del os.environ["SMR_TRACE_GLOBAL_ROWS"]
        os.environ["SMR_CIGAR_POOL_WORDS"] = "64"
        n += check(engine, kinds=["short", "indels"], max_pairs=60, schemes=[0])
    finally:
        os.environ.pop("SMR_TRACE_GLOBAL_ROWS", None)
        os.environ.pop("SMR_CIGAR_POOL_WORDS", None)
    return n
