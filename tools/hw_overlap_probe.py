"""Probe (GPU): does running two batches CONCURRENTLY -- two smr_ctx on the same device, one host thread each -- raise the throughput?
The sort of the seed stage waits on memory, the searches and k_chain are VALU-bound; kernels of different contexts can share the chip."""
import ctypes as C
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import sortmerna_amd as smr  # noqa: E402
from sortmerna_amd import synth  # noqa: E402

BATCH = 2_000_000
d = tempfile.mkdtemp(prefix="smr_ov_")
db = os.path.join(d, "db.fasta")
synth.make_db(db, 140_000_000, seed=42)
engs = [smr.Engine(0), smr.Engine(0)]
parts = smr.Index.build_gpu(engs[0], db, 18, 3072.0, 10000)
info = parts[0].info()
for e in engs:
    for s, ix in enumerate(parts):
        e.upload_index(ix, s)
codes, offs = synth.load_db_codes(db)
tot = 0
NB = 3
for k, e in enumerate(engs):
    for b in range(NB):
        letters = synth.make_reads(codes, offs, BATCH, read_len=150, frac_db=0.10, seed=1234 + 10 * k + b, sub=0.005, indel=0.0001, n_rate=0.001)
        o = (np.arange(BATCH + 1, dtype=np.uint64) * np.uint64(150))
        h = C.c_void_p()
        assert e.L.smr_reads_pack(letters.tobytes(), o.ctypes.data, BATCH, C.byref(h)) == 0
        r = smr.Reads(h)
        e.select_batch(b); e.upload_reads(r, 1)
        tot += r.total_len
        r.free()
ms = smr.minimal_score(0.618874, 0.343238, info, 2 * NB * BATCH, tot)


def steps(e, n):
    p = smr.default_params(minimal_score=ms)
    for i in range(n):
        e.select_batch(i % NB); e.reset_state()
        smr.align_resident(e, list(range(len(parts))), [p], with_cigar=True)


for e in engs:
    steps(e, 1)
t = time.perf_counter(); steps(engs[0], 4); t1 = time.perf_counter() - t
print("serial, one context: %.1f ms per step = %.2f M reads/s" % (t1 / 4 * 1e3, 4 * BATCH / t1 / 1e6), flush=True)
th = [threading.Thread(target=steps, args=(e, 4)) for e in engs]
t = time.perf_counter()
for x in th:
    x.start()
for x in th:
    x.join()
t2 = time.perf_counter() - t
print("two contexts concurrently: %.1f ms per step = %.2f M reads/s" % (t2 / 8 * 1e3, 8 * BATCH / t2 / 1e6), flush=True)
