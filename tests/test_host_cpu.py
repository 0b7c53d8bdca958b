"""CPU: the host side of libsmr_hip.so (no GPU call): C-ABI surface, read packer, index loader/builder/writer,
minimal_score, and the fail-loudly contract when no GPU is present."""
import ctypes as C
import filecmp
import os

import numpy as np
import pytest

import __graft_entry__ as entry
import sortmerna_amd as smr
from sortmerna_amd import capi
from helpers import golden, orc, paths, refrun


def test_library_exports_every_declared_symbol():
    L = capi.load()
    declared = entry.declared_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing
    assert sorted(capi.EXPORTS) == declared          # the Python binding covers the whole header


def test_no_gpu_means_loud_failure_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(smr.SmrError, match="no HIP device|no CPU fallback"):
        smr.Engine(0)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under sortmerna_amd/ or include/ may reference it."""
    bad = []
    for root in ("sortmerna_amd", "include"):
        for d, _, files in os.walk(os.path.join(paths.REPO, root)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                    t = open(os.path.join(d, f), errors="replace").read()
                    if "smr_oracle" in t or "oracle/" in t or "orc_" in t:
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def test_product_has_no_route_to_the_kernel_emulator():
    """tests/emu is a development aid: the product's Python never loads it, the product sources do not know it (the emulator puts ITS
    smr_device_ops.hpp in front of csrc/'s when it compiles the kernel sources for the host: no `#ifdef` for it anywhere in csrc/), and
    libsmr_hip.so carries none of its symbols"""
    import subprocess
    csrc = os.path.join(paths.REPO, "sortmerna_amd", "csrc")
    for f in os.listdir(csrc):
        if os.path.isfile(os.path.join(csrc, f)):
            assert "SMR_EMU" not in open(os.path.join(csrc, f)).read(), f
    for d, _, files in os.walk(os.path.join(paths.REPO, "sortmerna_amd")):
        for f in files:
            if f.endswith(".py"):
                t = open(os.path.join(d, f)).read()
                assert "emu" not in t.replace("enumerate", ""), os.path.join(d, f)
    from sortmerna_amd import build
    assert "SMR_EMU" not in open(build.__file__).read()
    syms = subprocess.check_output(["nm", "-D", "--defined-only", build.LIB]).decode()
    assert "emu" not in syms and "wave_exchange" not in syms


def test_reads_pack_layout_and_ambiguity_mask():
    seqs = ["ACGTUacgtuNnRX", "", "A" * 33 + "N", "ACGT" * 40]
    r = smr.Reads.from_seqs(seqs)
    assert (r.count, r.total_len, r.min_len, r.max_len) == (4, sum(map(len, seqs)), 0, 160)
    r.free()


def test_reads_load_fastx_multiline_and_range(tmp_path):
    p = tmp_path / "x.fasta"
    p.write_text(">a\nACGT\nACGT\n>b\nTTTT\n>c desc\nGG\nGG\nG")     # multi-line + unterminated last line
    r = smr.Reads.from_fastx(str(p))
    assert (r.count, r.total_len, r.max_len, r.min_len) == (3, 17, 8, 4)
    r2 = smr.Reads.from_fastx(str(p), 1, 1)
    assert (r2.count, r2.total_len) == (1, 4)
    q = tmp_path / "x.fastq"
    q.write_text("@r0\nACGTN\n+\nIIIII\n@r1\nAC\n+\nII\n")
    r3 = smr.Reads.from_fastx(str(q))
    assert (r3.count, r3.total_len) == (2, 7)
    with pytest.raises(smr.SmrError):
        smr.Reads.from_fastx(str(tmp_path / "missing.fa"))
    e = tmp_path / "empty.fa"
    e.write_text("")
    assert smr.Reads.from_fastx(str(e)).count == 0


def test_multithreaded_reader_equals_the_serial_one(tmp_path):
    """smr_reads_load_fastx_mt (byte ranges cut at record boundaries, one thread each) == smr_reads_load_fastx on FASTQ with '@' / '+'
    at the start of quality lines, multi-line FASTA, CRLF, a missing final newline, N letters, ragged lengths; any thread count"""
    rng = np.random.default_rng(5)
    L = np.frombuffer(b"ACGTN", dtype=np.uint8)
    fq = tmp_path / "r.fastq"
    with open(fq, "wb") as f:
        for i in range(30000):
            n = int(rng.integers(30, 200))
            seq = L[rng.choice(5, size=n, p=[.245, .245, .245, .245, .02])].tobytes()
            q = bytes(rng.integers(33, 74, size=n, dtype=np.uint8))
            if i % 7 == 0:
                q = b"@" + q[1:]                  # quality lines may start with '@' ...
            if i % 11 == 0:
                q = b"+" + q[1:]                  # ... or '+'
            f.write(b"@r%d some text\n" % i + seq + (b"\r\n" if i % 5 == 0 else b"\n") + b"+\n" + q + (b"" if i == 29999 else b"\n"))
    fa = tmp_path / "r.fasta"
    with open(fa, "wb") as f:
        for i in range(20000):
            n = int(rng.integers(1, 400))
            seq = L[rng.choice(5, size=n, p=[.245, .245, .245, .245, .02])].tobytes()
            f.write(b">s%d\n" % i)
            for k in range(0, n, 60):
                f.write(seq[k:k + 60] + b"\n")
    import gzip
    gz = tmp_path / "r.fastq.gz"                     # two gzip members back to back = the same text
    raw = fq.read_bytes()
    cutb = raw.index(b"\n@r15000 ") + 1
    gz.write_bytes(gzip.compress(raw[:cutb]) + gzip.compress(raw[cutb:]))
    for p in (fq, fa, gz):
        serial = smr.Reads.from_fastx(str(fq if p is gz else p))
        for t in (1, 2, 3, 8, 31):
            mt = smr.Reads.from_fastx_mt(str(p), t)
            assert (mt.count, mt.total_len, mt.min_len, mt.max_len) == (serial.count, serial.total_len, serial.min_len, serial.max_len), (p, t)
            assert mt.digest == serial.digest, (p, t)
            mt.free()
        serial.free()
    # the text-keeping variant: same batch, and every record's header / letters / quality as the (python) reference reader sees them
    from helpers import fastx
    for p in (fq, fa, gz):
        rt = smr.Reads.from_fastx_text(str(p), 5)
        ser = smr.Reads.from_fastx(str(fq if p is gz else p))
        assert rt.digest == ser.digest and rt.is_fastq == (p is not fa)
        exp = fastx.read_fastx(str(fq if p is gz else p))
        assert rt.count == len(exp)
        for i in list(range(0, rt.count, 997)) + [rt.count - 1]:
            h, sq, q = rt.record_text(i)
            assert (h, sq, q) == (exp[i][0], exp[i][1].rstrip("\r"), exp[i][2] or ""), (p, i)
        rt.free(); ser.free()
    e = tmp_path / "empty.fa"
    e.write_text("")
    assert smr.Reads.from_fastx_mt(str(e), 4).count == 0
    bad = tmp_path / "bad.gz"
    bad.write_bytes(gz.read_bytes()[:5000])
    with pytest.raises(RuntimeError, match="gzip"):
        smr.Reads.from_fastx_mt(str(bad), 2)


def test_minimal_score_equals_reference_log():
    g = golden.load()["syn_default"]
    db, _, seqs = golden.inputs("syn_default")
    parts = smr.Index.build(db, 18, 3072.0, 10000, 0)
    ms = smr.minimal_score(g["log"]["lambda"][0], g["log"]["K"][0], parts[0].info(), len(seqs), sum(map(len, seqs)))
    assert ms == g["log"]["minimal_score"][0]


def test_minimal_score_under_score_split_equals_reference_log():
    """-score_split (refstats.cpp:247-265): the read totals are divided by the reference's number of processing threads.  tests/golden/score_split.json
    holds what the unmodified binary logged for 1, 2, 4 and 7 threads (make_golden_score_split.py): 33, 32, 31, 30 on the synthetic inputs."""
    import json
    g = json.load(open(os.path.join(paths.GOLDEN, "score_split.json")))
    db, _, seqs = golden.inputs("syn_default")
    info = smr.Index.build(db, 18, 3072.0, 10000, 0)[0].info()
    seen = set()
    for r in g["runs"]:
        scale = r["threads"] if r["score_split"] else 1
        assert smr.minimal_score(r["lambda"], r["K"], info, len(seqs), sum(map(len, seqs)), full_read_scale=scale) == r["minimal_score"], r
        seen.add(r["minimal_score"])
    assert len(seen) >= 3                                   # the option does change the threshold on these inputs


def test_builder_part_split_matches_reference():
    db, _, _ = golden.inputs("syn_multipart")
    parts = smr.Index.build(db, 18, 0.15, 10000, 0)
    assert len(parts) == golden.load()["syn_multipart"]["index_parts"] == 4
    i0 = parts[0].info()
    assert i0.n_parts == 4 and i0.lnwin == 18 and i0.n_kmers == 4 ** 9


@pytest.mark.parametrize("L", [10, 12, 18])
def test_pigeonhole_layout_holds_the_same_entries_with_their_dfs_ranks(L):
    """the second device layout (per mini-trie: entries sorted by string and by their second half, exact-key directories, DFS ranks)
    vs the reference-shaped arena, every mini-trie; L = 10 and 12 give blocks with full directories, L = 18 mostly scan blocks"""
    db, _, _ = golden.inputs("syn_default")
    for ix in smr.Index.build(db, L, 3072.0, 10000, 0):
        ix.selfcheck()
    db, _, _ = golden.inputs("real_default")
    smr.Index.build(db, L, 3072.0, 10000, 0)[0].selfcheck()


def test_index_write_then_load_round_trip(tmp_path):
    db, _, _ = golden.inputs("syn_default")
    parts = smr.Index.build(db, 18, 3072.0, 10000, 0)
    pfx = str(tmp_path / "a")
    smr.Index.write_files(parts, db, pfx)
    again = smr.Index.load_files(pfx, 0, db)
    a, b = parts[0].info(), again.info()
    for f in ("lnwin", "n_kmers", "trie_words", "n_ids", "n_pos", "n_refs", "ref_bytes", "n_nodes", "n_buckets", "n_entries", "full_len", "numseq"):
        assert getattr(a, f) == getattr(b, f), f
    pfx2 = str(tmp_path / "b")
    smr.Index.write_files([again], db, pfx2)
    for ext in (".kmer_0.dat", ".bursttrie_0.dat", ".pos_0.dat", ".stats"):
        assert filecmp.cmp(pfx + ext, pfx2 + ext, shallow=False), ext


def test_damaged_index_files_are_refused(tmp_path):
    """the loader parses the part's files in place with many threads: a trie or position file cut anywhere must end in an error
    (SMR_ERR_IO), never in a crash or in an index with missing tries"""
    import shutil
    db, _, _ = golden.inputs("syn_default")
    parts = smr.Index.build(db, 18, 3072.0, 10000, 0)
    pfx = str(tmp_path / "a")
    smr.Index.write_files(parts, db, pfx)
    for ext in (".bursttrie_0.dat", ".pos_0.dat", ".kmer_0.dat"):
        whole = open(pfx + ext, "rb").read()
        for frac in (0.999, 0.5, 0.01):
            bad = str(tmp_path / ("bad%s_%g" % (ext.split("_")[0], frac)))
            for e in (".kmer_0.dat", ".bursttrie_0.dat", ".pos_0.dat", ".stats"):
                shutil.copy(pfx + e, bad + e)
            with open(bad + ext, "wb") as f:
                f.write(whole[: int(len(whole) * frac)])
            with pytest.raises(smr.SmrError):
                smr.Index.load_files(bad, 0, db)
    # a position file that announces 2^32 - 1 lists (12 * 4 G bytes of tables if the count were believed before the file size is checked)
    bad = str(tmp_path / "bad_nid")
    for e in (".kmer_0.dat", ".bursttrie_0.dat", ".pos_0.dat", ".stats"):
        shutil.copy(pfx + e, bad + e)
    whole = open(pfx + ".pos_0.dat", "rb").read()
    with open(bad + ".pos_0.dat", "wb") as f:
        f.write(b"\xff\xff\xff\xff" + whole[4:])
    with pytest.raises(smr.SmrError):
        smr.Index.load_files(bad, 0, db)
    with pytest.raises(smr.SmrError):
        smr.Index.load_files(str(tmp_path / "nothing_here"), 0, db)


@pytest.mark.skipif(not (paths.have_reference() and paths.have_ref_bin()), reason="needs /root/reference + oracle/_ref/sortmerna_ref")
def test_reference_index_files_round_trip_byte_exact(tmp_path):
    """Load the files the REFERENCE's indexer wrote (CMPH ids, its BFS trie stream) and write them back: identical bytes.
    Also: our own builder produces an index with the same statistics for the same FASTA."""
    db = os.path.join(paths.REF_DATA, "rRNA_databases", "silva-arc-23s-id98.fasta")
    idx = str(tmp_path / "idx")
    res = refrun.run_reference([db], [os.path.join(paths.REF_DATA, "illumina_GQ099317.fasta")], str(tmp_path / "wd"), idx_dir=idx, index_only=True)
    assert res.rc == 0, res.stdout[-800:]
    pfx = refrun.index_prefix_for(idx, db)
    ix = smr.Index.load_files(pfx, 0, db)
    ix.selfcheck()                                  # the pigeonhole device layout built from the reference's own tries
    out = str(tmp_path / "mine")
    smr.Index.write_files([ix], db, out)
    for ext in (".kmer_0.dat", ".bursttrie_0.dat", ".pos_0.dat", ".stats"):
        assert filecmp.cmp(pfx + ext, out + ext, shallow=False), ext
    mine = smr.Index.build(db, 18, 3072.0, 10000, 0)
    a, b = ix.info(), mine[0].info()
    # shape-independent statistics are identical; the burst SHAPE may differ in a few nodes: the reference bursts a
    # bucket one level at the insertion that makes it exceed 128 B (indexdb.cpp:225-228), so a child that inherits all 17
    # entries stays un-burst until its next insertion, whereas our sort-based builder bursts every bucket > 16 entries.
    # Search results do not depend on the shape (tests/test_oracle_golden.py pins that against the reference's records).
    for f in ("n_ids", "n_pos", "n_refs", "ref_bytes", "n_entries", "full_len", "numseq"):
        assert getattr(a, f) == getattr(b, f), f
    assert abs(a.n_nodes - b.n_nodes) < 0.01 * a.n_nodes
    assert list(a.bg) == list(b.bg)


def test_flat_index_cache_round_trip(tmp_path):
    """smr_index_save / smr_index_load_flat: the part that comes back writes byte-identical reference-format files (every array and every
    statistic survived), a wrong stamp and a damaged file are refused"""
    import hashlib
    import sortmerna_amd as smr
    from sortmerna_amd import synth
    db = str(tmp_path / "db.fasta")
    synth.make_db(db, 300_000, seed=9, family_size=6, mean_len=1300)
    parts = smr.Index.build(db, 18, 0.8, 10000, 0)
    assert len(parts) >= 2
    flats = []
    for k, ix in enumerate(parts):
        f = str(tmp_path / ("part%d.flat" % k))
        ix.save(f, stamp=77 + k)
        flats.append(smr.Index.load_flat(f, stamp=77 + k))
    smr.Index.write_files(parts, db, str(tmp_path / "a"))
    smr.Index.write_files(flats, db, str(tmp_path / "b"))
    names = sorted(n[2:] for n in os.listdir(tmp_path) if n.startswith("a."))
    assert len(names) >= 1 + 3 * len(parts)
    for n in names:
        da, dbb = open(tmp_path / ("a." + n), "rb").read(), open(tmp_path / ("b." + n), "rb").read()
        assert hashlib.sha1(da).digest() == hashlib.sha1(dbb).digest(), n
    i0, i1 = parts[0].info(), flats[0].info()
    assert (i0.n_ids, i0.n_pos, i0.trie_words, i0.numseq, i0.full_len, i0.n_parts) == (i1.n_ids, i1.n_pos, i1.trie_words, i1.numseq, i1.full_len, i1.n_parts)
    with pytest.raises(smr.SmrError, match="stamp"):
        smr.Index.load_flat(str(tmp_path / "part0.flat"), stamp=1)
    blob = open(tmp_path / "part0.flat", "rb").read()
    open(tmp_path / "cut.flat", "wb").write(blob[:len(blob) // 2])
    with pytest.raises(smr.SmrError, match="damaged"):
        smr.Index.load_flat(str(tmp_path / "cut.flat"), stamp=77)
    with pytest.raises(smr.SmrError):
        smr.Index.load_flat(str(tmp_path / "absent.flat"), stamp=77)
    for ix in parts + flats:
        ix.free()
