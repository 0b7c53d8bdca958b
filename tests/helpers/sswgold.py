"""tests/golden/ssw_pairs.json: (read, reference window) pairs and what the reference's own ssw_align returned for them
(written by tests/golden/make_golden_ssw.py).  TEST INFRASTRUCTURE."""
import json
import os

import numpy as np

from . import paths

_TR = bytes.maketrans(b"ACGTN", bytes(range(5)))


def load(name="ssw_pairs.json"):
    g = json.load(open(os.path.join(paths.REPO, "tests", "golden", name)))
    for c in g["cases"]:
        c["reads_b"] = [s.encode().translate(_TR) for s in c["reads"]]
        c["refs_b"] = [s.encode().translate(_TR) for s in c["refs"]]
        c["expected_a"] = np.array(c["expected"], dtype=np.int32)
    return g["cases"]


def check(engine, modes=(0, 1, 2), max_pairs=None):
    """both SW kernels through smr_ssw_batch against the reference's answers; returns the number of pairs checked"""
    n = 0
    for c in load():
        sc = c["scoring"]
        k = len(c["reads_b"]) if max_pairs is None else min(max_pairs, len(c["reads_b"]))
        for mode in modes:
            got = engine.ssw_batch(c["reads_b"][:k], c["refs_b"][:k], match=sc["match"], mismatch=sc["mismatch"], score_N=sc["score_N"],
                                   gap_open=sc["gap_open"], gap_ext=sc["gap_ext"], filters=sc["filters"], mode=mode)
            exp = c["expected_a"][:k]
            bad = np.nonzero((got != exp).any(axis=1))[0]
            assert bad.size == 0, "mode %d scoring %s: %d of %d pairs differ from ssw.c; first pair %d (m=%d, n=%d): got %s, ssw.c %s" % (
                mode, sc, bad.size, k, bad[0], len(c["reads_b"][bad[0]]), len(c["refs_b"][bad[0]]), got[bad[0]].tolist(), exp[bad[0]].tolist())
        n += k
    return n


def check_x4(engine):
    """the four-problems-per-wave kernel (smr_ssw_batch mode 3) on the pairs it takes (read spans <= 256, any four of them per wave)"""
    n = 0
    for c in load():
        sc = c["scoring"]
        idx = [i for i in range(len(c["reads_b"])) if len(c["reads_b"][i]) <= 256]
        got = engine.ssw_batch([c["reads_b"][i] for i in idx], [c["refs_b"][i] for i in idx], match=sc["match"], mismatch=sc["mismatch"], score_N=sc["score_N"],
                               gap_open=sc["gap_open"], gap_ext=sc["gap_ext"], filters=sc["filters"], mode=3)
        exp = c["expected_a"][idx]
        bad = np.nonzero((got != exp).any(axis=1))[0]
        assert bad.size == 0, "x4 kernel, scoring %s: %d of %d pairs differ from ssw.c; first pair %d (m=%d, n=%d): got %s, ssw.c %s" % (
            sc, bad.size, len(idx), idx[bad[0]], len(c["reads_b"][idx[bad[0]]]), len(c["refs_b"][idx[bad[0]]]), got[bad[0]].tolist(), exp[bad[0]].tolist())
        n += len(idx)
    return n


def check_striped(engine, max_pairs=None):
    """the slow path that reproduces ssw.c's stripe geometry (smr_ssw_batch mode 4) on the schemes under which the fast kernels' affine recurrence is NOT
    what ssw.c computes (tests/golden/ssw_pairs_striped.json, answers of the reference's own ssw.c) -- and on the default scheme, where both must agree"""
    n = 0
    for c in load("ssw_pairs_striped.json"):
        sc = c["scoring"]
        k = len(c["reads_b"]) if max_pairs is None else min(max_pairs, len(c["reads_b"]))
        got = engine.ssw_batch(c["reads_b"][:k], c["refs_b"][:k], match=sc["match"], mismatch=sc["mismatch"], score_N=sc["score_N"],
                               gap_open=sc["gap_open"], gap_ext=sc["gap_ext"], filters=sc["filters"], mode=4)
        exp = c["expected_a"][:k]
        bad = np.nonzero((got != exp).any(axis=1))[0]
        assert bad.size == 0, "striped path, scoring %s: %d of %d pairs differ from ssw.c; first pair %d (m=%d, n=%d): got %s, ssw.c %s" % (
            sc, bad.size, k, bad[0], len(c["reads_b"][bad[0]]), len(c["refs_b"][bad[0]]), got[bad[0]].tolist(), exp[bad[0]].tolist())
        n += k
    return n
