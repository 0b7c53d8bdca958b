"""Torch-free mini bench for a GPU session with ~1 minute of budget: bench.py's default workload (140 Mnt synthetic DB, 2 M-read batches
of 150-nt reads), index built ON THE DEVICE, 1 warm-up + 2 timed steps with the packed SW kernel and again with the 32-bit kernel.
Prints as it goes (also into gpurun_out/hw_minibench.log).  Not the contract bench (no roofline object, no CPU baseline): bench.py is."""
import ctypes as C
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "hw_minibench.log"), "a")
T0 = time.time()


def say(*a):
    line = "[%6.1fs] " % (time.time() - T0) + " ".join(str(x) for x in a)
    print(line, flush=True)
    LOG.write(line + "\n")
    LOG.flush()


import numpy as np  # noqa: E402
from sortmerna_amd import capi  # noqa: E402
capi.load(rebuild_if_stale=False)
import sortmerna_amd as smr  # noqa: E402
from sortmerna_amd import synth  # noqa: E402

DB_NT = int(os.environ.get("MB_DB_NT", 140_000_000))
BATCH = int(os.environ.get("MB_BATCH", 2_000_000))
NB = 3
d = tempfile.mkdtemp(prefix="smr_mb_")
db = os.path.join(d, "db.fasta")
t = time.time(); synth.make_db(db, DB_NT, seed=42); say("make_db %d nt: %.1f s" % (DB_NT, time.time() - t))
eng = smr.Engine(0)
say("engine; SW kernel after self-check:", eng.sw_mode())
t = time.time()
try:
    parts = smr.Index.build_gpu(eng, db, 18, 3072.0, 10000)
    say("DEVICE index build: %.2f s, %d part(s)" % (time.time() - t, len(parts)))
except Exception as x:  # noqa: BLE001
    say("device index build failed:", x)
    t = time.time(); parts = smr.Index.build(db, 18, 3072.0, 10000, 0); say("host index build: %.1f s" % (time.time() - t))
info = parts[0].info()
say("index: ids %d, positions %d, trie words %d" % (info.n_ids, info.n_pos, info.trie_words))
t = time.time()
for s, ix in enumerate(parts):
    eng.upload_index(ix, s)
say("index upload: %.1f s" % (time.time() - t))
t = time.time(); codes, offs = synth.load_db_codes(db); say("load_db_codes %.1f s" % (time.time() - t))
tot = 0
t = time.time()
for b in range(NB):
    letters = synth.make_reads(codes, offs, BATCH, read_len=150, frac_db=0.10, seed=1234 + b, sub=0.005, indel=0.0001, n_rate=0.001)
    o = (np.arange(BATCH + 1, dtype=np.uint64) * np.uint64(150))
    h = C.c_void_p()
    assert eng.L.smr_reads_pack(letters.tobytes(), o.ctypes.data, BATCH, C.byref(h)) == 0
    r = smr.Reads(h)
    eng.select_batch(b); eng.upload_reads(r, 1)
    tot += r.total_len
    r.free()
say("%d batches of %d reads resident: %.1f s" % (NB, BATCH, time.time() - t))
ms = smr.minimal_score(0.618874, 0.343238, info, NB * BATCH, tot)
params = smr.default_params(minimal_score=ms)


def step(b):
    eng.select_batch(b); eng.reset_state()
    smr.align_resident(eng, list(range(len(parts))), [params], with_cigar=True)


START = eng.sw_mode()                                  # 1, or 2 with SMR_SW_PACKED=2 (the wave_ror variant)
for mode in ((START, 0, START) if os.environ.get('MB_SW32') else (START, START)):
    eng.sw_mode(mode)
    step(0)
    eng.prof_reset()
    t = time.perf_counter()
    for b in (1, 2):
        step(b)
    dt = time.perf_counter() - t
    p = eng.prof()
    eng.select_batch(1)
    al = eng.counters(1)["num_aligned"]
    say("spec SW issued %d used %d; " % (p.n_sw_spec, p.n_sw_spec_used) + "SW kernel %s: %.2f M reads/s (%.1f ms per 2 M-read step); seed stage %.2f ms/launch x %d, k_chain %.2f ms/launch x %d, k_trace %.2f ms x %d; aligned(batch 1) %d" % (
        {0: "32-bit", 1: "packed", 2: "packed (wave_ror)"}[mode], 2 * BATCH / dt / 1e6, dt / 2 * 1e3, p.seed_ms / max(p.seed_launches, 1), p.seed_launches,
        p.chain_ms / max(p.chain_launches, 1), p.chain_launches, p.trace_ms / max(p.trace_launches, 1), p.trace_launches, al))
    say("kernels: " + "  ".join("%s %.3f ms x %d (%.0f GB/s)" % (k, v["ms"] / max(v["launches"], 1), v["launches"], v["bytes"] / max(v["ms"], 1e-9) / 1e6) for k, v in eng.prof_kernels().items()))
if os.environ.get("MB_HOST_BUILD"):
    t = time.time(); h2 = smr.Index.build(db, 18, 3072.0, 10000, 0); say("host index build: %.1f s" % (time.time() - t))
eng.close()
say("done")
