"""Minimal FASTA/FASTQ reader for the tests (sequence may span several lines in FASTA)."""
import gzip


def read_fastx(path):
    """-> list of (header_line, sequence, quality_or_None)"""
    op = gzip.open if path.endswith(".gz") else open
    recs = []
    with op(path, "rt") as f:
        lines = [l.rstrip("\r\n") for l in f]
    i = 0
    n = len(lines)
    while i < n:
        l = lines[i]
        if not l:
            i += 1
            continue
        if l[0] == ">":
            hdr = l
            i += 1
            seq = []
            while i < n and (not lines[i] or lines[i][0] != ">"):
                seq.append(lines[i].strip())
                i += 1
            recs.append((hdr, "".join(seq), None))
        elif l[0] == "@":
            hdr = l
            seq = lines[i + 1]
            qual = lines[i + 3]
            recs.append((hdr, seq, qual))
            i += 4
        else:
            raise ValueError("bad record at line %d of %s" % (i, path))
    return recs
