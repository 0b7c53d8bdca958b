"""BASELINE config 2 at FULL size as a committed fixture: the reference's bundled amplicon reads
(data/set2_environmental_study_550_amplicon.fasta.gz, 100 000 reads) against its bundled silva-arc-16s-id95 DB (the DB the config
names, silva-bac-16s-id85, is a release download that is not in the repository), run through the UNMODIFIED reference binary
(oracle/_ref/sortmerna_ref, 1 thread so that KVDB keys are read numbers).  Committed: the two input files (copies of the reference's
test data, gzip), and config2.json with the log values and MD5 digests of the per-read records (Read::toBinString bytes), one per
1000 reads plus the total -- small, and a differing chunk still localises a failure.

    python tests/golden/make_golden_config2.py          # needs /root/reference and `make -C oracle ref`
"""
import gzip
import hashlib
import json
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import fastx, paths, refrun  # noqa: E402

OUT = os.path.join(HERE, "config2")
CHUNK = 1000


def digests(records):
    tot = hashlib.md5()
    chunks = []
    for c in range(0, len(records), CHUNK):
        h = hashlib.md5()
        for r in records[c:c + CHUNK]:
            b = len(r).to_bytes(4, "little") + r
            h.update(b)
            tot.update(b)
        chunks.append(h.hexdigest())
    return tot.hexdigest(), chunks


def main():
    os.makedirs(OUT, exist_ok=True)
    db_src = os.path.join(paths.REF_DATA, "rRNA_databases", "silva-arc-16s-id95.fasta")
    gz_src = os.path.join(paths.REF_DATA, "set2_environmental_study_550_amplicon.fasta.gz")
    tmp = tempfile.mkdtemp(prefix="smr_c2_")
    flat = os.path.join(tmp, "reads.fasta")
    with gzip.open(gz_src, "rb") as f, open(flat, "wb") as g:
        g.write(f.read())
    n = len(fastx.read_fastx(flat))
    out = {"reads": os.path.basename(gz_src), "db": os.path.basename(db_src) + ".gz", "n_reads": n, "chunk": CHUNK, "runs": {}}
    for name, extra in (("default", []), ("num_alignments_0", ["-num_alignments", "0"])):
        res = refrun.run_reference([db_src], [flat], os.path.join(tmp, "wd_" + name), extra=extra + ["-v"], threads=1,
                                   idx_dir=os.path.join(paths.ORACLE_DIR, "_ref", "idx_cache"), timeout=7200)
        assert res.rc == 0, res.stdout[-2000:]
        recs = [res.kvdb.get(b"0_%d" % i, b"") for i in range(n)]
        tot, chunks = digests(recs)
        out["runs"][name] = {"params": {"num_alignments": 0} if extra else {}, "lambda": res.log["lambda"][0], "K": res.log["K"][0],
                             "minimal_score": res.log["minimal_score"][0], "num_aligned": res.log["num_aligned"],
                             "n_records": sum(1 for r in recs if r), "n_alignments": sum(len(refrun.parse_record(r)["alignv"]) for r in recs if r),
                             "md5_total": tot, "md5_chunks": chunks}
        print(name, {k: v for k, v in out["runs"][name].items() if k != "md5_chunks"})
    shutil.copyfile(gz_src, os.path.join(OUT, os.path.basename(gz_src)))
    with open(db_src, "rb") as f, gzip.GzipFile(os.path.join(OUT, os.path.basename(db_src) + ".gz"), "wb", mtime=0) as g:
        g.write(f.read())
    json.dump(out, open(os.path.join(OUT, "config2.json"), "w"), indent=0)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
