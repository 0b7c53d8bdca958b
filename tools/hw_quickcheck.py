"""A 20-second hardware check without torch (for a GPU session with almost no budget): does smr_create's packed-SW self-check pass on
the device, do both SW kernels reproduce the reference's ssw.c answers, do a golden case and the device index build come out right?
    python tools/hw_quickcheck.py      (prints one line per check, also into gpurun_out/hw_quickcheck.log)"""
import hashlib
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "hw_quickcheck.log"), "a")
T0 = time.time()


def say(*a):
    line = "[%6.1fs] " % (time.time() - T0) + " ".join(str(x) for x in a)
    print(line, flush=True)
    LOG.write(line + "\n")
    LOG.flush()


from sortmerna_amd import capi  # noqa: E402
capi.load(rebuild_if_stale=False)                     # the library built in the container; never rebuild here
import sortmerna_amd as smr  # noqa: E402
say("library loaded")
e = smr.Engine(0)
say("engine created; SW kernel in use after smr_create's self-check:", "packed" if e.sw_mode() == 1 else "32-bit (SELF-CHECK FAILED)")
for n, seed, mx in ((2000, 3, 300), (500, 4, 1200), (100, 5, 3500)):
    say("sw_selfcheck cases", 2 * n, "max_len", mx, "-> differing:", e.sw_selfcheck(n, seed, mx))
from helpers import sswgold  # noqa: E402
try:
    say("ssw.c known answers, both kernels: pairs checked", sswgold.check(e))
except AssertionError as x:
    say("ssw.c known answers: FAILED", str(x)[:300])
from helpers import golden  # noqa: E402
from helpers.cases import gpu_run  # noqa: E402
for case in ("syn_default", "real_all", "two_db_default"):
    o = gpu_run(e, case, tempfile.mkdtemp(prefix="hwq_"))
    exp = golden.records(case)
    bad = [i for i, (a, b) in enumerate(zip(o["records"], exp)) if a != b]
    say("golden case", case, "records differing:", len(bad), "of", len(exp))


def digest(parts, db):
    d = tempfile.mkdtemp(prefix="hwq_ix_")
    smr.Index.write_files(parts, db, os.path.join(d, "i"))
    h = hashlib.md5()
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


for db, mb, mp in ((os.path.join(ROOT, "tests", "golden", "syn_db.fasta"), 3072.0, 10000), (os.path.join(ROOT, "tests", "golden", "syn_db.fasta"), 0.15, 3)):
    t = time.time()
    dev = smr.Index.build_gpu(e, db, 18, mb, mp)
    td = time.time() - t
    host = smr.Index.build(db, 18, mb, mp, 0)
    say("device index build", os.path.basename(db), mb, mp, "%.2f s" % td, "files identical to the host build:", digest(dev, db) == digest(host, db))
e.close()
say("done")
