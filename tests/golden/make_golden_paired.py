"""Paired-read report fixtures: the UNMODIFIED reference run on two mate files (the first 200 pairs of the bundled set4 mate-pair FASTQ
files) against tests/golden/real_db.fasta with -fastx -other and the pairing options; per variant the read ids of every output file.

    python tests/golden/make_golden_paired.py     # rewrites tests/golden/paired/*

paired_1.fastq / paired_2.fastq   the inputs;  paired.records.bin  the reference's per-read records (mate 1 of pair i, mate 2 of pair i, ...);
paired.json   {variant: {"options": [...], "files": {file name: [read id, ...]}}}"""
import json
import os
import shutil
import struct
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)

from helpers import paths, refrun  # noqa: E402

VARIANTS = {"two_files": [], "paired_in": ["-paired_in"], "paired_out": ["-paired_out"], "out2": ["-out2"], "paired_in_out2": ["-paired_in", "-out2"],
            "paired_out_out2": ["-paired_out", "-out2"], "sout": ["-sout"], "out2_sout": ["-out2", "-sout"]}
N_PAIRS = 200


def main():
    assert paths.have_reference() and paths.have_ref_bin()
    out = os.path.join(HERE, "paired")
    os.makedirs(out, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="golden_paired_")
    rd = []
    for k in (1, 2):
        src = os.path.join(paths.REF_DATA, "set4_mate_pairs_metatranscriptomics_%d.fastq" % k)
        dst = os.path.join(out, "paired_%d.fastq" % k)
        with open(src) as f, open(dst, "w") as g:
            g.writelines(f.readlines()[: 4 * N_PAIRS])
        rd.append(dst)
    db = os.path.join(HERE, "real_db.fasta")
    G = {}
    for name, extra in VARIANTS.items():
        res = refrun.run_reference([db], rd, os.path.join(tmp, name), extra=extra + ["-fastx", "-other", "-v"], threads=1)
        assert res.rc == 0, res.stdout[-2000:]
        o = os.path.join(tmp, name, "out")
        files = {}
        for fn in sorted(os.listdir(o)):
            if fn.endswith(".fq"):
                files[fn] = [l.split()[0][1:] for l in open(os.path.join(o, fn)).readlines()[0::4]]
        G[name] = dict(options=extra, files=files)
        print(name, {k: len(v) for k, v in files.items()})
        if name == "two_files":
            recs = []
            for i in range(N_PAIRS):
                recs.append(res.kvdb.get(b"0_%d" % i, b""))
                recs.append(res.kvdb.get(b"1_%d" % i, b""))
            with open(os.path.join(out, "paired.records.bin"), "wb") as f:
                f.write(struct.pack("<I", len(recs)))
                for r in recs:
                    f.write(struct.pack("<I", len(r)))
                    f.write(r)
            G[name]["log"] = res.log
    # one interleaved file (mate 1, mate 2, mate 1, ...) with -paired_in / -paired_out
    inter = os.path.join(tmp, "interleaved.fastq")
    a, b = open(rd[0]).readlines(), open(rd[1]).readlines()
    with open(inter, "w") as f:
        for i in range(N_PAIRS):
            f.writelines(a[4 * i:4 * i + 4]); f.writelines(b[4 * i:4 * i + 4])
    for name, extra in (("interleaved_paired_in", ["-paired_in"]), ("interleaved_paired_out_out2", ["-paired_out", "-out2"])):
        res = refrun.run_reference([db], [inter], os.path.join(tmp, name), extra=extra + ["-fastx", "-other", "-v"], threads=1)
        assert res.rc == 0, res.stdout[-2000:]
        o = os.path.join(tmp, name, "out")
        files = {fn: [l.split()[0][1:] for l in open(os.path.join(o, fn)).readlines()[0::4]] for fn in sorted(os.listdir(o)) if fn.endswith(".fq")}
        G[name] = dict(options=extra, files=files)
        print(name, {k: len(v) for k, v in files.items()})
    json.dump(G, open(os.path.join(out, "paired.json"), "w"), indent=0, sort_keys=True)
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
