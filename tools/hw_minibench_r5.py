"""Round-5 mini bench: ONE process, several library configurations on the same resident index and read batches (a gpurun call costs minutes
before its command starts, so A/B runs share one).  bench.py's default workload (140 Mnt synthetic DB, 150-nt reads, 10 % from the DB),
MB_BATCH reads per batch (default 8 M), index built on the device once; every configuration = a fresh engine created under its environment.

    python tools/hw_minibench_r5.py NAME[:ENV=val[,ENV=val...]] ...

Prints per configuration: reads/s, ms per step, per kernel family HIP-event ms per launch.  Not the contract bench (bench.py is)."""
import ctypes as C
import os
import sys
import tempfile
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
BATCH = int(os.environ.get("MB_BATCH", 8_000_000))
NB = int(os.environ.get("MB_NB", 2))
STEPS = int(os.environ.get("MB_STEPS", 3))
d = tempfile.mkdtemp(prefix="smr_mb_")
db = os.path.join(d, "db.fasta")
t = time.time(); synth.make_db(db, DB_NT, seed=42); say("make_db %d nt: %.1f s" % (DB_NT, time.time() - t))
eng = smr.Engine(0)
t = time.time()
parts = smr.Index.build_gpu(eng, db, 18, 3072.0, 10000)
say("DEVICE index build: %.2f s, %d part(s)" % (time.time() - t, len(parts)))
eng.close()
info = parts[0].info()
codes, offs = synth.load_db_codes(db)
reads, tot = [], 0
t = time.time()
for b in range(NB):
    letters = synth.make_reads_fast(codes, offs, BATCH, read_len=150, frac_db=0.10, seed=1234 + b, sub=0.005, indel=0.0001, n_rate=0.001)
    o = (np.arange(BATCH + 1, dtype=np.uint64) * np.uint64(150))
    h = C.c_void_p()
    assert capi.load().smr_reads_pack(letters.tobytes(), o.ctypes.data, BATCH, C.byref(h)) == 0
    r = smr.Reads(h)
    reads.append(r)
    tot += r.total_len
say("%d batches of %d reads packed: %.1f s" % (NB, BATCH, time.time() - t))
ms = smr.minimal_score(0.618874, 0.343238, info, NB * BATCH, tot)

for cfg in sys.argv[1:] or ["base"]:
    name, _, envs = cfg.partition(":")
    env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        eng = smr.Engine(0)
        for s, ix in enumerate(parts):
            eng.upload_index(ix, s)
        for b, r in enumerate(reads):
            eng.select_batch(b); eng.upload_reads(r, 1)
        params = smr.default_params(minimal_score=ms)

        def step(b):
            eng.select_batch(b); eng.reset_state()
            smr.align_resident(eng, list(range(len(parts))), [params], with_cigar=os.environ.get("MB_CIGAR", "1") != "0")

        step(0)
        eng.prof_reset()
        t = time.perf_counter()
        for i in range(STEPS):
            step((i + 1) % NB)
        dt = time.perf_counter() - t
        p = eng.prof()
        eng.select_batch(1 % NB)
        al = eng.counters(1)["num_aligned"]
        say("== %s %s: %.2f M reads/s (%.1f ms per %d-read step); seed %.2f ms/launch, chain family %.1f ms/step, trace %.2f ms/step; sw_fwd %d spec %d used %d; aligned %d" % (
            name, env, STEPS * BATCH / dt / 1e6, dt / STEPS * 1e3, BATCH, p.seed_ms / max(p.seed_launches, 1), p.chain_ms / STEPS, p.trace_ms / STEPS,
            p.n_sw_fwd // STEPS, p.n_sw_spec // STEPS, p.n_sw_spec_used // STEPS, al))
        say("   hit lists of %d entries per search, candidate walk rounds per pass %s" % (p.hit_list_cap, eng.walk_rounds()))
        say("   per step: " + "  ".join("%s %.2f ms (x%d)" % (k, v["ms"] / STEPS, v["launches"] // STEPS) for k, v in eng.prof_kernels().items() if v["launches"]))
        eng.close()
    except Exception as x:  # noqa: BLE001
        say("== %s FAILED: %s" % (name, x))
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
say("done")
