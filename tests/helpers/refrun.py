"""Drive the unmodified reference binary (oracle/_ref/sortmerna_ref) and collect what it produced.
TEST INFRASTRUCTURE ONLY."""
import glob
import os
import re
import shutil
import struct
import subprocess

from . import paths


def parse_kvdb_dump(path):
    """-> dict key(bytes) -> value(bytes).  Format written by oracle/shim/rocksdb/db.h."""
    d = open(path, "rb").read()
    (n,) = struct.unpack_from("<Q", d, 0)
    o = 8
    out = {}
    for _ in range(n):
        (kl,) = struct.unpack_from("<Q", d, o)
        o += 8
        k = d[o:o + kl]
        o += kl
        (vl,) = struct.unpack_from("<Q", d, o)
        o += 8
        out[k] = d[o:o + vl]
        o += vl
    return out


def parse_record(b):
    """Read::toBinString bytes -> dict (read.cpp:429-462, ssw.hpp:106-140)."""
    if not b:
        return None
    o = 0
    f = {}
    (f["lastIndex"], f["lastPart"], _, _, _, _) = struct.unpack_from("<6I", b, o)
    o += 24
    f["is_done"], f["is_hit"], f["null_align_output"] = struct.unpack_from("<3B", b, o)
    o += 3
    (f["max_SW_count"],) = struct.unpack_from("<H", b, o)
    o += 2
    (f["num_alignments"],) = struct.unpack_from("<i", b, o)
    o += 4
    (f["hit_seeds"],) = struct.unpack_from("<I", b, o)
    o += 4
    (asz,) = struct.unpack_from("<Q", b, o)
    o += 8
    f["min_index"], f["max_index"] = struct.unpack_from("<2I", b, o)
    o += 8
    (na,) = struct.unpack_from("<Q", b, o)
    o += 8
    al = []
    for _ in range(na):
        (rl,) = struct.unpack_from("<Q", b, o)
        o += 8
        (cl,) = struct.unpack_from("<Q", b, o)
        o += 8
        cig = struct.unpack_from("<%dI" % cl, b, o)
        o += 4 * cl
        ref_num, rb, re_, qb, qe, readlen = struct.unpack_from("<IiiiiI", b, o)
        o += 24
        score1, part, index_num, strand = struct.unpack_from("<HHHB", b, o)
        o += 7
        al.append(dict(cigar=cig, ref_num=ref_num, ref_begin1=rb, ref_end1=re_, read_begin1=qb, read_end1=qe,
                       readlen=readlen, score1=score1, part=part, index_num=index_num, strand=strand))
    f["alignv"] = al
    assert o == len(b), (o, len(b))
    return f


def parse_readstats(b):
    """Readstats::toBstring (readstats.cpp:133-174)."""
    v = struct.unpack_from("<QQIIQQQQQQQ", b, 0)
    n = v[10]
    per_db = struct.unpack_from("<%dQ" % n, b, 80)
    return dict(all_reads_count=v[0], all_reads_len=v[1], min_read_len=v[2], max_read_len=v[3], num_aligned=v[4],
                num_short=v[9], reads_matched_per_db=list(per_db))


def parse_log(path):
    t = open(path).read()
    out = {"lambda": [float(x) for x in re.findall(r"Gumbel lambda = ([0-9.eE+-]+)", t)],
           "K": [float(x) for x in re.findall(r"Gumbel K = ([0-9.eE+-]+)", t)],
           "minimal_score": [int(x) for x in re.findall(r"Minimal SW score based on E-value = (\d+)", t)]}
    m = re.search(r"Total reads = (\d+)", t)
    out["total_reads"] = int(m.group(1)) if m else None
    m = re.search(r"Total reads passing E-value threshold = (\d+)", t)
    out["num_aligned"] = int(m.group(1)) if m else None
    return out


class RefResult:
    pass


def run_reference(refs, reads, workdir, extra=(), threads=1, timeout=1800, idx_dir=None, index_only=False):
    """Run sortmerna_ref; returns RefResult with .kvdb (dict), .log (dict), .workdir, .idx_prefixes"""
    if os.path.isdir(os.path.join(workdir, "kvdb")):
        shutil.rmtree(os.path.join(workdir, "kvdb"))
    if os.path.isdir(os.path.join(workdir, "out")):
        shutil.rmtree(os.path.join(workdir, "out"))
    os.makedirs(workdir, exist_ok=True)
    cmd = [paths.REF_BIN]
    for r in refs:
        cmd += ["-ref", r]
    for r in reads:
        cmd += ["-reads", r]
    cmd += ["-workdir", workdir, "-threads", str(threads)]
    if idx_dir:
        cmd += ["-idx-dir", idx_dir]
    if index_only:
        cmd += ["-index", "1"]
    cmd += list(extra)
    env = dict(os.environ)
    dump = os.path.join(workdir, "kvdb_dump.bin")
    env["SMR_KVDB_DUMP"] = dump
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    res = RefResult()
    res.rc = p.returncode
    res.stdout = p.stdout.decode("latin-1")
    res.workdir = workdir
    res.cmd = cmd
    d = idx_dir or os.path.join(workdir, "idx")
    res.idx_prefixes = sorted(x[:-len(".stats")] for x in glob.glob(os.path.join(d, "*.stats")))
    if index_only:
        return res
    res.kvdb = parse_kvdb_dump(dump) if os.path.isfile(dump) else {}
    logp = os.path.join(workdir, "out", "aligned.log")
    res.log = parse_log(logp) if os.path.isfile(logp) else {}
    m = re.search(r"Done alignment in ([0-9.eE+-]+) sec", res.stdout)
    res.align_sec = float(m.group(1)) if m else None
    return res


def index_prefix_for(idx_dir, ref_fasta):
    """The reference names index files by std::hash of the FASTA basename (index.cpp:75-77); we find it by the
    name stored inside the .stats file instead of re-deriving the hash."""
    base = os.path.basename(ref_fasta).encode()
    for st in glob.glob(os.path.join(idx_dir, "*.stats")):
        b = open(st, "rb").read(4096)
        (nl,) = struct.unpack_from("<I", b, 8)
        name = b[12:12 + nl].rstrip(b"\0")
        if os.path.basename(name) == base:
            return st[:-len(".stats")]
    raise FileNotFoundError("no index for %s in %s" % (ref_fasta, idx_dir))
