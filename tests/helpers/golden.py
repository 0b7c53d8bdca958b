"""Access to the committed golden fixtures (tests/golden/, produced by tests/golden/make_golden.py from the
unmodified reference).  TEST INFRASTRUCTURE ONLY."""
import json
import os
import struct

from . import fastx, paths

_G = None


def load():
    global _G
    if _G is None:
        _G = json.load(open(os.path.join(paths.GOLDEN, "golden.json")))
    return _G


def records(case):
    d = open(os.path.join(paths.GOLDEN, load()[case]["records"]), "rb").read()
    (n,) = struct.unpack_from("<I", d, 0)
    o = 4
    out = []
    for _ in range(n):
        (l,) = struct.unpack_from("<I", d, o)
        o += 4
        out.append(d[o:o + l])
        o += l
    return out


def inputs(case):
    """-> (db fasta path or list of paths, reads fasta path, [read sequences])"""
    stem = case.split("_")[0] if case.startswith(("syn", "real")) else ("two" if case.startswith("two_db") else case)
    dbs = {"t0": "t0_ref.fasta", "t9": "t9_ref.fasta", "syn": "syn_db.fasta", "real": "real_db.fasta", "two": ["syn_db.fasta", "real_db.fasta"]}[stem]
    rd = os.path.join(paths.GOLDEN, {"t0": "t0_read.fasta", "t9": "t9_reads.fasta", "syn": "syn_reads.fasta", "real": "real_reads.fasta",
                                     "two": "two_db_reads.fasta"}[stem])
    db = [os.path.join(paths.GOLDEN, d) for d in dbs] if isinstance(dbs, list) else os.path.join(paths.GOLDEN, dbs)
    return db, rd, [r[1] for r in fastx.read_fastx(rd)]


def cigar_string(cig, read_begin1, read_end1, readlen):
    """BAM-style u32 cigar -> text with soft clips, as report_blast.cpp:293-311 / report_sam.cpp:118-135 print it."""
    s = ""
    if read_begin1 > 0:
        s += "%dS" % read_begin1
    for c in cig:
        s += "%d%s" % (c >> 4, "MID"[c & 0xF])
    tail = readlen - read_end1 - 1
    if tail > 0:
        s += "%dS" % tail
    return s
