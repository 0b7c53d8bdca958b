"""Differential fuzzing of the kernel sources against the CPU oracle, without a GPU: seeded random workloads (DB shape, read lengths, error and N rates,
seed length) x random points of the option space of the path (scoring scheme, num_seeds, min_lis, edges, best / num_alignments, strands, full search,
minoccur, strides, minimal score), each run through the oracle and through the kernels compiled for the host against the wave64 emulator (tests/emu);
records (Read::toBinString bytes) and counters must be equal.  TEST INFRASTRUCTURE (uses oracle/ and tests/helpers); the GPU parity tests cover a
fixed list of option sets -- this looks for the combinations nobody wrote down.

    python tools/fuzz_emu.py [first_seed [n_cases]]        # prints one line per case, the differing ones with everything needed to repeat them
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import sortmerna_amd as smr  # noqa: E402
from helpers import emu, orc  # noqa: E402
from helpers.workload import Workload  # noqa: E402

SCHEMES = [(2, -3, 5, 2), (2, -3, 5, 2), (2, -3, 3, 2), (3, -4, 6, 3), (5, -4, 5, 2), (1, -2, 3, 1), (2, -3, 4, 3), (4, -5, 7, 3), (2, -3, 10, 2), (1, -1, 2, 1),
           (2, -3, 3, 3), (2, -5, 2, 1)]      # the last two (round 6): schemes that go through the striped slow path (smr_sw_striped.hpp)


def case(seed, tmp):
    rng = np.random.Generator(np.random.PCG64(seed))
    pick = lambda xs: xs[int(rng.integers(0, len(xs)))]  # noqa: E731
    lnwin = pick([18, 18, 18, 18, 16, 14, 12, 12])       # (10 with 300-letter reads: every read meets most references -- minutes per case on the emulator)
    wk = dict(db_nt=int(pick([40_000, 80_000, 150_000, 300_000])), n_reads=int(pick([200, 400, 700, 33, 12])), read_len=int(pick([40, 75, 100, 150, 150, 220, 301, 301, 600, 1100])),
              frac_db=float(pick([0.2, 0.5, 0.8])), seed=seed, n_rate=float(pick([0.0, 0.002, 0.02, 0.06])), family_size=int(pick([1, 4, 40, 40, 200])),
              lnwin=lnwin, mean_len=int(pick([300, 1500])), db_kw=dict(sub_lo=float(pick([0.0, 0.01, 0.03])), sub_hi=float(pick([0.02, 0.06, 0.10])), indel=float(pick([0.0, 0.005, 0.02]))))
    if wk["read_len"] >= 600:
        wk["n_reads"] = min(wk["n_reads"], 60)
        wk["mean_len"] = 1500
    if wk["db_kw"]["sub_hi"] < wk["db_kw"]["sub_lo"]:
        wk["db_kw"]["sub_hi"] = wk["db_kw"]["sub_lo"] + 0.01
    wk["db_kw"]["min_len"] = min(400, wk["mean_len"])
    match, mismatch, go, ge = pick(SCHEMES)
    score_n = pick([mismatch, mismatch, 0, -1, -min(2 * go, 2 * ge, 127), 1])      # (+1: a positive N score -- the striped slow path since round 6)
    opts = dict(match=match, mismatch=mismatch, gap_open=go, gap_ext=ge, score_N=int(score_n),
                num_seeds=int(pick([1, 2, 2, 2, 3, 4])), min_lis=int(pick([1, 2, 2, 3, 4])), is_best=int(pick([1, 1, 0])), num_alignments=int(pick([0, 1, 1, 2, 3, 5, 8])),
                is_full_search=int(pick([0, 0, 1])), minoccur=int(pick([0, 0, 0, 1, 3])))
    if pick([0, 0, 1]):
        opts["edges"], opts["is_as_percent"] = int(pick([6, 8, 10, 10])), 1
    else:
        opts["edges"] = int(pick([1, 2, 4, 4, 10]))
    fr = pick([(1, 1), (1, 1), (1, 0), (0, 1)])
    opts["is_forward"], opts["is_reverse"] = fr
    half = lnwin // 2
    opts["skiplengths"] = pick([[lnwin, half, 3], [lnwin, half, 3], [lnwin, lnwin, 3], [lnwin, 6, 2], [half, half, half], [lnwin, half, 1]])
    delta = int(pick([0, 0, 0, -20, 15, 60]))
    wk["read_kw"] = dict(sub=float(pick([0.0, 0.005, 0.005, 0.03, 0.08])), indel=float(pick([0.0, 0.0001, 0.002, 0.01])))
    if pick([0, 0, 1]):
        wk["db_ambiguous"] = float(pick([0.0005, 0.003]))
    if pick([0, 0, 0, 1]):
        wk["max_mb"] = float(pick([0.4, 0.8, 1.5]))          # several index parts
    w = Workload(tmp, **wk)
    ms = max(1, int(w.minimal_score) + delta)
    return w, wk, opts, ms


def second_db(w, wk, tmp, seed):
    """a second reference DB for the same reads: the first one's sequences with 3 % of their letters changed and a fifth of them dropped -- reads align to
    both with different scores (best-N replacement across DBs, the per-DB counters moving with it: alignment.cpp:420-459)"""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0xDB2))
    out, keep = [], True
    for line in open(w.db, "rb"):
        if line.startswith(b">"):
            keep = rng.random() < 0.8
            if keep:
                out.append(line)
        elif keep:
            a = np.frombuffer(line.rstrip(b"\r\n"), dtype=np.uint8).copy()
            m = rng.random(len(a)) < 0.03
            a[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(m.sum()))]
            out.append(a.tobytes() + b"\n")
    os.makedirs(os.path.join(tmp, "db2"), exist_ok=True)
    db2 = os.path.join(tmp, "db2", "second.fasta")
    open(db2, "wb").write(b"".join(out))
    return Workload(os.path.join(tmp, "db2"), db_fasta=db2, seqs=w.seqs, lnwin=wk["lnwin"], max_mb=wk.get("max_mb", 3072.0))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    bad = 0
    import contextlib
    # FUZZ_ON_GPU=1: the same cases on the real device instead of the emulator (a case that is hours of emulation -- a dense family under the striped slow path -- is seconds there)
    with (contextlib.nullcontext() if os.environ.get("FUZZ_ON_GPU") == "1" else emu.active()):
        for seed in range(first, first + n):
            t = time.time()
            # the library's own switches (each one a different route to the same records), drawn per case: a context reads them when it is created
            rng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
            env = {}
            for name, vals in (("SMR_WALK_ROUNDS", [None, None, "1", "2", "3"]), ("SMR_WALK_K", [None, None, "1", "2", "8"]), ("SMR_WALK_ASSUME", [None, None, "0", "100"]),
                               ("SMR_WALK_SPLIT", [None, None, None, "0"]), ("SMR_HANDOVER", [None, None, None, "0"]), ("SMR_CAND_BLOOM", [None, None, None, "64"]),
                               ("SMR_PG_CAND_CAP", [None, None, None, "8"]), ("SMR_WALK_GATHER", [None, None, None, "0"]),
                               # round 6: repeated seeds searched once / large bins sorted by several blocks at thresholds of a test's size, one seed sort for all parts
                               # forced or off, single-hit windows with a segment after all
                               ("SMR_SEED_DEDUP", [None, None, "2", "16", "0"]), ("SMR_SEED_HOT_BIN", [None, None, "64"]), ("SMR_SEED_HOT_SUB", [None, None, "200"]),
                               ("SMR_SEED_SHARED", [None, None, "2", "0"]), ("SMR_SEG_INLINE", [None, None, None, "0"])):
                v = vals[int(rng.integers(0, len(vals)))]
                os.environ.pop(name, None)
                if v is not None:
                    os.environ[name] = env[name] = v
            e = smr.Engine(0)
            with tempfile.TemporaryDirectory(prefix="smr_fuzz_") as tmp:
                try:
                    w, wk, opts, ms = case(seed, tmp)
                    o_opts = dict(opts)
                    o_opts["lnwin"] = wk["lnwin"]
                    ws = [w]
                    if seed % 4 == 3:
                        ws.append(second_db(w, wk, tmp, seed))
                        if seed % 8 == 7:
                            ws.reverse()                    # the mutated DB first: the second one then replaces alignments with better ones
                    mss = [max(1, int(x.minimal_score) + (ms - int(w.minimal_score))) for x in ws]
                    run = orc.Run(w.seqs)
                    for k, x in enumerate(ws):
                        for part in range(x.stats.nparts):
                            po = orc.default_params(minimal_score=mss[k], **o_opts)
                            po.index_num, po.part, po.is_last_index_part = k, part, int(k == len(ws) - 1 and part == x.stats.nparts - 1)
                            run.align_part(x.prefix, x.db, x.stats, part, po)
                    recs_o = run.records()
                    ctr_o = dict(num_aligned=run.counters.num_aligned, num_short=run.counters.num_short, per_db=[int(run.counters.reads_matched_per_db[k]) for k in range(len(ws))])
                    run.close()
                    e.set_seed_mode(int(os.environ["FUZZ_SEED_MODE"]) if "FUZZ_SEED_MODE" in os.environ else (seed & 1 if seed % 5 == 0 else 0))      # 1: the DFS seed kernel
                    try:
                        ps = [smr.default_params(minimal_score=m, **opts) for m in mss]
                        smr.align(e, w.reads, [x.parts for x in ws], ps, max_alignments_per_read=(256 if opts["num_alignments"] == 0 else None))
                        recs_g, ctr_g = e.records(), e.counters(len(ws))
                    except smr.SmrError as x:
                        if "rounds to 0 letters" in str(x) or "max_alignments_per_read" in str(x):      # documented limits, said explicitly
                            print("seed %d refused: %s" % (seed, str(x)[:110]), flush=True)
                            e.close()
                            continue
                        raise
                    assert len(recs_g) == len(recs_o), (len(recs_g), len(recs_o))
                    diff = [i for i, (a, b) in enumerate(zip(recs_g, recs_o)) if a != b]
                    same_ctr = ctr_g["num_aligned"] == ctr_o["num_aligned"] and list(ctr_g["reads_matched_per_db"][:len(ws)]) == ctr_o["per_db"] and ctr_g["num_short"] == ctr_o["num_short"]
                    ok = not diff and same_ctr
                    print("seed %d %s: %d reads, %d aligned%s, %.1f s%s" % (seed, "ok" if ok else "DIFFERS", len(recs_o), ctr_o["num_aligned"], " (two DBs: %s)" % ctr_o["per_db"] if len(ws) > 1 else "", time.time() - t,
                                                                        "" if ok else "  records differing %d (first read %s), counters gpu %s oracle %s\n    workload %s\n    options %s minimal_score %d\n    switches %s" % (
                                                                            len(diff), diff[:1], {k: ctr_g[k] for k in ("num_aligned", "num_short")}, {k: ctr_o[k] for k in ("num_aligned", "num_short")}, wk, opts, ms, env)), flush=True)
                    bad += not ok
                except Exception as x:  # noqa: BLE001
                    print("seed %d ERROR %s: %s  (switches %s)" % (seed, type(x).__name__, x, env), flush=True)
                    bad += 1
            e.close()
    print("%d case(s), %d differing or failing" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
