"""BASELINE config 2 at full size without a GPU: the bundled set2 amplicon reads (100 000 reads, .gz) against the bundled
silva-arc-16s-id95 DB (the DB the config names, silva-bac-16s-id85, is not in the repository), three ways:
  (1) the unmodified reference binary (oracle/_ref/sortmerna_ref, 1 thread so that KVDB keys are read numbers),
  (2) the kernel sources on the wave64 host emulator (tests/emu), reads loaded by smr_reads_load_fastx_mt straight from the .gz,
  (3) optionally the C oracle (--oracle).
Per-read records (Read::toBinString bytes) must be identical.  Test infrastructure; needs /root/reference (build container only).

    python tools/emu_fullsize_check.py [--reads N] [--all] [--oracle]
"""
import argparse
import gzip
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import sortmerna_amd as smr  # noqa: E402
from helpers import emu, fastx, orc, paths, refrun  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=0, help="first N reads only (0 = all 100 000)")
    ap.add_argument("--all", action="store_true", help="-num_alignments 0")
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--db", action="append", help="reference FASTA (repeatable: several --ref in one run); default silva-arc-16s-id95")
    ap.add_argument("--reads-file", default=None, help="FASTA/FASTQ instead of the bundled set2 (.gz accepted)")
    a = ap.parse_args()
    if not a.db:
        a.db = [os.path.join(paths.REF_DATA, "rRNA_databases", "silva-arc-16s-id95.fasta")]
    tmp = tempfile.mkdtemp(prefix="smr_full_")
    gz = a.reads_file or os.path.join(paths.REF_DATA, "set2_environmental_study_550_amplicon.fasta.gz")
    is_fq = ".fastq" in gz or ".fq" in gz
    flat = os.path.join(tmp, "reads.fastq" if is_fq else "reads.fasta")
    with (gzip.open(gz, "rb") if gz.endswith(".gz") else open(gz, "rb")) as f, open(flat, "wb") as g:
        data = f.read()
        if a.reads:
            if is_fq:
                data = b"\n".join(data.split(b"\n")[: 4 * a.reads]) + b"\n"
            else:
                data = b">".join(data.split(b">")[: a.reads + 1])
        g.write(data)
    recs_in = fastx.read_fastx(flat)
    seqs = [r[1] for r in recs_in]
    print("reads:", len(seqs), "letters:", sum(map(len, seqs)))
    extra = ["-num_alignments", "0"] if a.all else []
    params = {"num_alignments": 0} if a.all else {}
    t = time.time()
    cache = os.path.join(paths.ORACLE_DIR, "_ref", "idx_cache")
    res = refrun.run_reference(a.db, [flat], os.path.join(tmp, "wd"), extra=extra + ["-v"], threads=1, idx_dir=cache, timeout=7200)
    assert res.rc == 0, res.stdout[-2000:]
    exp = [res.kvdb.get(b"0_%d" % i, b"") for i in range(len(seqs))]
    print("reference: %.0f s, aligned %d, minimal_score %s" % (time.time() - t, res.log["num_aligned"], res.log["minimal_score"]))
    ms = res.log["minimal_score"][0]
    with emu.active():
        t = time.time()
        parts = [smr.Index.build(db, 18, 3072.0, 10000, 0) for db in a.db]
        print("index build: %.0f s, parts %s" % (time.time() - t, [len(x) for x in parts]))
        reads = smr.Reads.from_fastx_mt(gz if not a.reads else flat, 0)
        assert reads.count == len(seqs)
        eng = smr.Engine(0)
        plist = [smr.default_params(minimal_score=res.log["minimal_score"][k], **params) for k in range(len(a.db))]
        t = time.time()
        smr.align(eng, reads, parts, plist, with_cigar=True, max_alignments_per_read=256 if a.all else None)
        got = eng.records()
        ctr = eng.counters(len(a.db))
        print("emulated kernels (SW kernel mode %d): %.0f s, aligned %d, per db %s" % (eng.sw_mode(), time.time() - t, ctr["num_aligned"], ctr["reads_matched_per_db"]))
        eng.close()
    bad = [i for i in range(len(seqs)) if got[i] != exp[i]]
    print("kernels vs reference: %d of %d records differ%s" % (len(bad), len(seqs), (" first " + str(bad[:5])) if bad else ""))
    if bad:
        i = bad[0]
        print(" got", refrun.parse_record(got[i]))
        print(" exp", refrun.parse_record(exp[i]))
    assert ctr["num_aligned"] == res.log["num_aligned"]
    if a.oracle:
        assert len(a.db) == 1, "--oracle: one --db"
        a.db = a.db[0]
        prefix = refrun.index_prefix_for(cache, a.db)
        st = orc.load_stats(prefix)
        run = orc.Run(seqs)
        po = orc.default_params(minimal_score=ms, **params)
        t = time.time()
        for part in range(st.nparts):
            po.part = part
            po.is_last_index_part = int(part == st.nparts - 1)
            run.align_part(prefix, a.db, st, part, po)
        orecs = run.records()
        bo = [i for i in range(len(seqs)) if orecs[i] != exp[i]]
        print("oracle: %.0f s; oracle vs reference: %d records differ" % (time.time() - t, len(bo)))
        assert not bo
    assert not bad
    print("OK")


if __name__ == "__main__":
    main()
