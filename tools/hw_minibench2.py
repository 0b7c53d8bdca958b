"""GPU session aid: do two batches in flight on ONE GPU (two contexts = two HIP streams, one host thread each) finish sooner than the same
batches one after the other?  Same workload as hw_minibench.py.  MB_CTX=1 runs the sequential reference with the same code."""
import ctypes as C
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T0 = time.time()


def say(*a):
    print("[%6.1fs] " % (time.time() - T0) + " ".join(str(x) for x in a), flush=True)


import numpy as np  # noqa: E402
from sortmerna_amd import capi  # noqa: E402
capi.load(rebuild_if_stale=False)
import sortmerna_amd as smr  # noqa: E402
from sortmerna_amd import synth  # noqa: E402

DB_NT = int(os.environ.get("MB_DB_NT", 140_000_000))
BATCH = int(os.environ.get("MB_BATCH", 2_000_000))
NCTX = int(os.environ.get("MB_CTX", 2))
NB = 2                                                   # resident batches per context
STEPS = int(os.environ.get("MB_STEPS", 4))               # timed steps per context
d = tempfile.mkdtemp(prefix="smr_mb2_")
db = os.path.join(d, "db.fasta")
synth.make_db(db, DB_NT, seed=42)
engs = [smr.Engine(0) for _ in range(NCTX)]
parts = smr.Index.build_gpu(engs[0], db, 18, 3072.0, 10000)
info = parts[0].info()
for e in engs:
    for s, ix in enumerate(parts):
        e.upload_index(ix, s)
say("%d context(s), index resident in each" % NCTX)
codes, offs = synth.load_db_codes(db)
tot = 0
for k, e in enumerate(engs):
    for b in range(NB):
        letters = synth.make_reads(codes, offs, BATCH, read_len=150, frac_db=0.10, seed=1234 + NB * k + b, sub=0.005, indel=0.0001, n_rate=0.001)
        o = (np.arange(BATCH + 1, dtype=np.uint64) * np.uint64(150))
        h = C.c_void_p()
        assert e.L.smr_reads_pack(letters.tobytes(), o.ctypes.data, BATCH, C.byref(h)) == 0
        r = smr.Reads(h)
        e.select_batch(b); e.upload_reads(r, 1)
        tot += r.total_len
        r.free()
ms = smr.minimal_score(0.618874, 0.343238, info, NCTX * NB * BATCH, tot)
params = smr.default_params(minimal_score=ms)


def run(e, n):
    for i in range(n):
        e.select_batch(i % NB); e.reset_state()
        smr.align_resident(e, list(range(len(parts))), [params], with_cigar=True)


for e in engs:
    run(e, 1)                                            # warm-up, one context at a time
for rep in range(2):
    th = [threading.Thread(target=run, args=(e, STEPS)) for e in engs]
    t = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t
    al = [e.counters(1)["num_aligned"] for e in engs]
    say("%d context(s) x %d steps of %d reads: %.2f M reads/s (%.1f ms per step of one context); aligned in the last batch of each: %s" % (
        NCTX, STEPS, BATCH, NCTX * STEPS * BATCH / dt / 1e6, dt / STEPS * 1e3, al))
for e in engs:
    e.close()
say("done")
