"""Second set of report fixtures: what the UNMODIFIED reference writes for `-blast 0` (BLAST-like pairwise text), `-sam -SQ`
(@SQ header lines) and aligned.log, on inputs already committed under tests/golden/ (written by make_golden.py).

    python tests/golden/make_golden_reports2.py      # rewrites tests/golden/reports2/*

Per case: <case>.pairwise.txt = aligned.blast, <case>.sam_header.txt = the @-lines of aligned.sam, <case>.log.txt = aligned.log.
Paths inside the files are rewritten relative to tests/golden/ so the fixtures do not depend on the checkout location."""
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)

from helpers import golden, paths, refrun  # noqa: E402

CASES = {"t0": [], "t9": ["-num_alignments", "0", "-mismatch", "-3"], "syn_default": [], "real_default": []}


def main():
    assert paths.have_reference() and paths.have_ref_bin()
    out = os.path.join(HERE, "reports2")
    os.makedirs(out, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="golden2_")
    for case, extra in CASES.items():
        db, rd, _ = golden.inputs(case)
        wd = os.path.join(tmp, case)
        res = refrun.run_reference([db], [rd], wd, extra=list(extra) + ["-blast", "0", "-sam", "-SQ", "-v"], threads=1)
        assert res.rc == 0, res.stdout[-2000:]
        o = os.path.join(wd, "out")
        fix = lambda t: t.replace(HERE + "/", "").replace(wd, "WORKDIR").replace(paths.REF_BIN, "sortmerna")
        open(os.path.join(out, case + ".pairwise.txt"), "w").write(open(os.path.join(o, "aligned.blast")).read())
        open(os.path.join(out, case + ".sam_header.txt"), "w").write(fix("".join(l for l in open(os.path.join(o, "aligned.sam")) if l.startswith("@"))))
        open(os.path.join(out, case + ".log.txt"), "w").write(fix(open(os.path.join(o, "aligned.log")).read()))
        print(case, os.path.getsize(os.path.join(out, case + ".pairwise.txt")), "bytes of pairwise text")
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
