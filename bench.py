#!/usr/bin/env python3
"""bench.py -- reads/sec of the MI355X hot path (libsmr_hip) on BASELINE.json's headline workload.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload illumina150|refs8|pacbio5k]
  (N > 1: one rank per GPU; started by torch.distributed.run, or by this script itself when no launcher is around it)

--workload: illumina150 (default) = BASELINE.json configs[2], the configuration the metric is quoted on (below);
            refs8    = configs[3]: the same reads against EIGHT resident reference DBs (bundled silva-arc-16s-id95 + 7 seeded synthetic ones
                       sized like the rRNA_databases set, incl. 5S / 5.8S families shorter than a read), (index, part) loop of
                       processor.cpp:219-277, reads_matched_per_db[8] through the counter reduce;
            pacbio5k = configs[4]: PacBio-like reads ~N(5000, 500) nt with 12 % errors against a 28S-like DB, long-read Smith-Waterman in
                       strips + wide traceback (the `--sam --blast 1` path needs every alignment's CIGAR).

Workload (BASELINE.json configs[2]): synthetic 150-nt Illumina-like reads (10 % sampled from
the DB with sequencing errors, 90 % random background) against an rRNA-like reference DB of the size of
smr_v4.3_default_db.fasta (140 Mnt; the real file is a release download that is not available offline, so a seeded
synthetic DB of families of mutated copies stands in -- `config.workload` says so).  A "step" is one pass of the
whole hot path -- both strands, all three seed passes, LIS chaining, Smith-Waterman, banded traceback, result fetch --
over ONE batch of `--batch-reads` reads (default 8 M: the larger the batch, the smaller the share of launch gaps and of the tail of the
persistent kernels -- 2 M: 23.5, 4 M: 24.7, 8 M: 25.3 M reads/s, profiles/r3s6_*) that is already resident in HBM; consecutive steps use
different batches and every step starts from a reset per-read state.  The 10 M read job of the config is 1.25 such steps.

Reads shard across GPUs (one rank per GPU, each with a full index replica, no data-path collective); the only
collectives are the two tiny all-reduces the reference's semantics need: global read totals before (they define
minimal_score, refstats.cpp:247-265) and the Readstats counters after (RCCL).

The JSON line also carries
  roofline      the seed-stage kernel with the most time in the timed steps: the algorithmic HBM bytes THAT kernel counts for itself on the
                device (smr_prof_kernels: tuples, directory words, strings looked at, accepted {rank, id}, hit segments) / its own HIP-event
                time, against the 8 TB/s HBM3E peak; `kernels` has every seed-stage kernel, `seed_stage` their sum, `counters` the issue-side
                view (VALU issue fraction, LDS bank conflicts) from profiles/sq_counters.json when it was measured on these sources; what
                the REFERENCE's traversal would move for the same reads (SURVEY.md 8d formula) only as `equivalent_rate`
  kernels       HIP-event time and launch count of each kernel family in the timed region
  cpu_baseline  the UNMODIFIED reference (oracle/_ref/sortmerna_ref) timed on this host's cores on a bounded sample of
                the same reads with the same index files (rank 0, N=1 only)
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

# Gumbel (lambda, K) of the scoring scheme 2/-3/5/2 for a near-uniform background, as the reference's vendored ALP
# computes them (refstats.cpp:194-233); inputs of smr_minimal_score.
GUMBEL = (0.618874, 0.343238)
HBM_PEAK_GBS = 8000.0
MAX_RESIDENT = 14          # the engine holds 16 batch slots; slot 15 is the CPU-baseline sample's


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench] " + msg, file=sys.stderr, flush=True)


def eng_counters_aligned(eng, b):
    eng.select_batch(b)
    return int(eng.counters(1)["num_aligned"])


def read_kvdb_dump(path):
    """key -> value of the dump the reference binary's KVDB stand-in writes (oracle/shim/rocksdb/db.h: u64 n, n x (u64 klen, key, u64 vlen, value))"""
    import struct
    d = open(path, "rb").read()
    (n,) = struct.unpack_from("<Q", d, 0)
    o, out = 8, {}
    for _ in range(n):
        (kl,) = struct.unpack_from("<Q", d, o)
        k = d[o + 8:o + 8 + kl]
        o += 8 + kl
        (vl,) = struct.unpack_from("<Q", d, o)
        out[k] = d[o + 8:o + 8 + vl]
        o += 8 + vl
    return out


def cpu_baseline(args, dbs, parts_per_db, sample, smr, eng, idx_slots):
    """Time the unmodified reference on a bounded sample (sample = (blob, offsets) of its reads); compare the ids it aligns (aligned.fq)
    and its per-read records (KVDB values) with the GPU path's."""
    ref_bin = os.path.join(HERE, "oracle", "_ref", "sortmerna_ref")
    strhash = os.path.join(HERE, "oracle", "_ref", "strhash")
    cores = os.cpu_count() or 1
    if not (os.path.isfile(ref_bin) and os.path.isfile(strhash)):
        return {"value": None, "unit": "reads/s", "cores": cores, "kind": "reference", "sample": "oracle/_ref/sortmerna_ref not built"}
    from sortmerna_amd import synth
    blob, offs = sample
    n = len(offs) - 1
    wd = tempfile.mkdtemp(prefix="smr_cpu_")
    try:
        reads = os.path.join(wd, "sample.fastq")
        synth.write_fastq_ragged(reads, blob, offs)
        idx = os.path.join(wd, "idx")
        os.makedirs(idx)
        t0 = time.time()
        for db, parts in zip(dbs, parts_per_db):
            h = subprocess.check_output([strhash, os.path.basename(db)]).decode().strip()
            smr.Index.write_files(parts, db, os.path.join(idx, h))
        log("index files for the reference written in %.1fs" % (time.time() - t0))
        threads = args.cpu_threads if args.cpu_threads > 0 else min(cores, 64)
        cmd = [ref_bin]
        for db in dbs:
            cmd += ["-ref", db]
        cmd += ["-reads", reads, "-workdir", os.path.join(wd, "run"), "-idx-dir", idx, "-threads", str(threads), "-v"] + WORKLOADS[args.workload]["ref_opts"]
        t0 = time.time()
        dump = os.path.join(wd, "kvdb_dump.bin")             # the reference's per-read records (Read::toBinString values), written by the KVDB stand-in it is built with
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500, env=dict(os.environ, SMR_KVDB_DUMP=dump))
        out = p.stdout.decode("latin-1")
        wall = time.time() - t0
        m = re.findall(r"done index: \d+ part: \d+ in ([0-9.eE+-]+) sec", out)
        if p.returncode != 0 or not m:
            return {"value": None, "unit": "reads/s", "cores": threads, "kind": "reference", "sample": "reference run failed rc=%d: %s" % (p.returncode, out[-300:])}
        align_s = sum(float(x) for x in m)
        res = {"value": n / align_s, "unit": "reads/s", "cores": threads, "kind": "reference",
               "sample": "first %d reads of batch 0, same index files, alignment stage only (%.2f s over %d (index, part) passes; whole process %.1f s incl. index load and reports)" % (n, align_s, len(m), wall)}
        logp = os.path.join(wd, "run", "out", "aligned.log")
        if os.path.isfile(logp):
            t = open(logp).read()
            ms = [int(x) for x in re.findall(r"Minimal SW score based on E-value = (\d+)", t)]
            na = re.search(r"Total reads passing E-value threshold = (\d+)", t)
            if len(ms) == len(dbs) and na:
                # same sample through the GPU path with the reference's own minimal_score per DB: ids and records must be equal
                import ctypes as C
                h = C.c_void_p()
                assert eng.L.smr_reads_pack(blob, offs.ctypes.data, n, C.byref(h)) == 0
                r = smr.Reads(h)
                eng.select_batch(15)
                eng.upload_reads(r, 1)
                smr.align_resident(eng, idx_slots, [smr.default_params(minimal_score=x) for x in ms], with_cigar=True)
                gpu_ids = set(i for i in range(n) if eng.is_hit(i))
                gpu_ctr = eng.counters(len(dbs))
                ref_ids = None
                for nm in ("aligned.fq", "aligned.fastq"):
                    fq = os.path.join(wd, "run", "out", nm)
                    if os.path.isfile(fq):
                        with open(fq, "rb") as f:
                            ref_ids = set(int(l[2:].split()[0]) for k, l in enumerate(f) if k % 4 == 0)
                res["parity"] = {"reference_aligned": int(na.group(1)), "gpu_aligned": int(gpu_ctr["num_aligned"]),
                                 "minimal_score": ms,
                                 "aligned_read_ids_equal": (ref_ids == gpu_ids) if ref_ids is not None else None,
                                 "ids_only_reference": len(ref_ids - gpu_ids) if ref_ids is not None else None,
                                 "ids_only_gpu": len(gpu_ids - ref_ids) if ref_ids is not None else None}
                if os.path.isfile(dump):
                    # records, not only ids: every read's Read::toBinString value (alignments with coordinates, scores, CIGARs, best-N bookkeeping)
                    # as the reference stored it against smr_result_record of the same read, byte for byte.
                    # The reference names a read '<slot>_<number within the slot>' (readfeed.cpp:793): one slot per thread, contiguous record
                    # ranges of the file in slot order (readfeed.cpp:1253-1277); its log gives every slot's read count.
                    ref_rec = read_kvdb_dump(dump)
                    per_slot = dict((int(a), int(b)) for a, b in re.findall(r"EOF reached\. Slot: (\d+) Total reads: (\d+)", out))
                    if sorted(per_slot) == list(range(threads)) and sum(per_slot.values()) == n:
                        eng.fetch()
                        n_rec = n_bad = i = 0
                        first_bad = None
                        for sl in range(threads):
                            for k in range(per_slot[sl]):
                                a = ref_rec.get(b"%d_%d" % (sl, k), b"")
                                b = eng.record(i)
                                n_rec += 1 if a else 0
                                if a != b:
                                    n_bad += 1
                                    first_bad = i if first_bad is None else first_bad
                                i += 1
                        res["parity"].update({"records_compared": n, "reference_records": n_rec, "records_differing": n_bad, "records_equal": n_bad == 0,
                                              "first_differing_read": first_bad})
                    else:
                        res["parity"]["records_equal"] = None
                        res["parity"]["records_note"] = "the reference's log did not give the read count of every slot: %r" % (per_slot,)
                if len(dbs) > 1:
                    res["parity"]["gpu_reads_matched_per_db"] = [int(x) for x in gpu_ctr["reads_matched_per_db"][:len(dbs)]]
                r.free()
        return res
    finally:
        shutil.rmtree(wd, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------------------------------------
# workloads (BASELINE.json configs[2], [3], [4]; SURVEY.md 8d "Configs restated as concrete inputs")
# ---------------------------------------------------------------------------------------------------------------------------------------
WORKLOADS = {
    # (cpu_sample_reads: 2 M reads = ~45 s of the reference's 64 threads on this workload -- round 5 timed 200 000 reads in 4.6 s, and a ratio should not rest on start-up)
    "illumina150": {"batch_reads": 8_000_000, "cpu_sample_reads": 2_000_000, "ref_opts": ["-fastx"],
                    "options": "default options (--fastx, best 1)"},
    "refs8": {"batch_reads": 8_000_000, "cpu_sample_reads": 100_000, "ref_opts": ["-fastx"],
              "options": "default options (--fastx, best 1), 8 --ref"},
    "pacbio5k": {"batch_reads": 50_000, "cpu_sample_reads": 4_000, "ref_opts": ["-fastx", "-sam", "-blast", "1"],
                 "options": "--sam --blast 1 (every alignment with its CIGAR), best 1"},
    # BASELINE configs[1] on REAL data: the reference's bundled 100 000 amplicon reads (scripts/t3.jinja) against its bundled silva-arc-16s-id95
    # (the id85 file the config names is not in the repository), the fixture of tests/golden/config2.  One pass over 100 000 reads is launch
    # bound, so a batch is the read set 80 times over (8 M reads); the CPU-baseline sample is the read set itself, all 100 000 records compared
    "config2": {"batch_reads": 8_000_000, "cpu_sample_reads": 100_000, "ref_opts": ["-fastx"],
                "options": "default options (--fastx, best 1), classify-only"},
}
CONFIG2_DIR = os.path.join(HERE, "tests", "golden", "config2")
# the 8-ref set: sizes (nt) and typical sequence lengths of sortmerna's rRNA_databases files; only silva-arc-16s-id95 is bundled with
# this repository (tests/golden/config2, the reference's own file), the others are seeded synthetic families of the same size
REFS8 = [("silva-bac-16s-like", 19_000_000, 1500, 400, 101), ("silva-arc-16s-id95", None, None, None, None), ("silva-euk-18s-like", 13_000_000, 1800, 400, 103),
         ("silva-bac-23s-like", 12_000_000, 2900, 400, 104), ("silva-arc-23s-like", 700_000, 2900, 400, 105), ("silva-euk-28s-like", 14_000_000, 3600, 400, 106),
         ("rfam-5s-like", 7_000_000, 119, 90, 107), ("rfam-5.8s-like", 2_000_000, 154, 120, 108)]


def workload_dbs(args, synth, cache, rank):
    """-> [(name, fasta path)]; rank 0 writes the files"""
    out = []
    if args.workload == "illumina150":
        specs = [("synth_rrna_db_%d" % args.db_nt, args.db_nt, 1500, 400, 42)]
    elif args.workload == "pacbio5k":
        specs = [("synth_28s_like_%d" % args.db_nt, args.db_nt, 5500, 400, 77)]
    elif args.workload == "config2":
        specs = [("silva-arc-16s-id95", None, None, None, None)]
    else:
        sc = args.db_nt / 140_000_000.0                       # (--db-nt scales the synthetic members; the tests use tiny ones)
        specs = [(n, (int(nt * sc) if nt else None), ml, mn, sd) for n, nt, ml, mn, sd in REFS8]
    for name, nt, mean_len, min_len, seed in specs:
        path = os.path.join(cache, name + ("_%d" % nt if nt and args.workload == "refs8" else "") + ".fasta")
        if rank == 0 and not os.path.isfile(path):
            if nt is None:                                   # the bundled real DB (a test-sized run, --db-nt below 14 M, takes its first 150 sequences)
                import gzip
                src = os.path.join(HERE, "tests", "golden", "config2", "silva-arc-16s-id95.fasta.gz")
                with gzip.open(src, "rb") as f, open(path + ".tmp", "wb") as g:
                    if args.db_nt >= 14_000_000 or args.workload == "config2":
                        shutil.copyfileobj(f, g)
                    else:
                        nseq = 0
                        for line in f:
                            nseq += line.startswith(b">")
                            if nseq > 150:
                                break
                            g.write(line)
            else:
                synth.make_db(path + ".tmp", nt, seed=seed, mean_len=mean_len, min_len=min_len, tag=name.replace("-", "_") + "_f")
            os.replace(path + ".tmp", path)
        out.append((name, path))
    return out


def load_all_codes(synth, paths):
    """codes / offsets of all DBs together (reads are sampled from their union, uniformly over sequences)"""
    import numpy as np
    cs, os_ = [], [np.zeros(1, dtype=np.int64)]
    base = 0
    for p in paths:
        c, o = load_codes_iupac(p)
        cs.append(c)
        os_.append(o[1:] + base)
        base += int(o[-1])
    return np.concatenate(cs), np.concatenate(os_)


def load_codes_iupac(path):
    """like synth.load_db_codes, for FASTA with lower case / IUPAC letters / multi-line records (the bundled real DB): other letters -> A"""
    import numpy as np
    seqs, cur = [], []
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if cur:
                    seqs.append(b"".join(cur))
                cur = []
            else:
                cur.append(line.strip())
    if cur:
        seqs.append(b"".join(cur))
    lut = np.zeros(256, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
        lut[ord(chr(c).lower())] = i
    lut[ord("U")] = lut[ord("u")] = 3
    offs = np.zeros(len(seqs) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(x) for x in seqs])
    return lut[np.frombuffer(b"".join(seqs), dtype=np.uint8)], offs


_CONFIG2_READS = None


def config2_reads():
    """the bundled amplicon reads: (blob of ASCII letters, offsets uint64[n+1])"""
    global _CONFIG2_READS
    if _CONFIG2_READS is None:
        import gzip
        import numpy as np
        seqs, cur = [], []
        with gzip.open(os.path.join(CONFIG2_DIR, "set2_environmental_study_550_amplicon.fasta.gz"), "rb") as f:
            for line in f:
                if line.startswith(b">"):
                    if cur:
                        seqs.append(b"".join(cur))
                    cur = []
                else:
                    cur.append(line.strip())
        if cur:
            seqs.append(b"".join(cur))
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(x) for x in seqs], dtype=np.uint64)
        _CONFIG2_READS = (b"".join(seqs), offs)
    return _CONFIG2_READS


def make_batch(args, synth, codes, offs, n, seed):
    """-> (blob of ASCII letters, offsets uint64[n+1])"""
    import numpy as np
    if args.workload == "config2":
        # the read set over and over, whole copies first (read i of the batch = read i mod 100 000 of the file)
        blob, o = config2_reads()
        nr = len(o) - 1
        reps, rest = divmod(n, nr)
        lens = np.diff(o)
        all_lens = np.concatenate([np.tile(lens, reps), lens[:rest]])
        oo = np.zeros(n + 1, dtype=np.uint64)
        oo[1:] = np.cumsum(all_lens, dtype=np.uint64)
        return blob * reps + blob[:int(o[rest])], oo
    if args.workload == "pacbio5k":
        L = args.long_read_len
        return synth.make_long_reads(codes, offs, n, mean_len=L, sd_len=L // 10, min_len=L // 5, max_len=6 * L, seed=seed)
    letters = synth.make_reads_fast(codes, offs, n, read_len=args.read_len, frac_db=0.10, seed=seed, sub=0.005, indel=0.0001, n_rate=0.001)
    return letters.tobytes(), (np.arange(n + 1, dtype=np.uint64) * np.uint64(args.read_len))


def make_shard(args, synth, codes, offs, rank, world):
    """--scaling strong: the job is ONE fixed read set of --total-reads reads, made of 8 seeded units (N units when 8 is not a multiple of N);
    rank r of N takes the units [r U / N, (r + 1) U / N) -- contiguous record ranges of the same file whatever N is, like the reference's
    split of a reads file over its threads (readfeed.cpp:1253-1277)"""
    import numpy as np
    units = 8 if 8 % world == 0 else world
    per = args.total_reads // units
    mine = range(rank * units // world, (rank + 1) * units // world)
    blobs, offl, base = [], [np.zeros(1, dtype=np.uint64)], 0
    for u in mine:
        n = per + (args.total_reads - per * units if u == units - 1 else 0)
        b, o = make_batch(args, synth, codes, offs, n, 777 + u)
        blobs.append(b)
        offl.append(o[1:] + np.uint64(base))
        base += int(o[-1])
    return b"".join(blobs), np.concatenate(offl)


def self_launch(n):
    """`python bench.py --gpus N ...` without a launcher: start the N ranks (one per GPU) through torch.distributed.run on this node,
    exactly as the driver's wrapped command would, and hand their output and exit code through.  Rank 0 prints the JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    log("--gpus %d without a launcher: starting %d ranks through torch.distributed.run (port %d)" % (n, n, port))
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE)
    for line in p.stdout:                                   # stdout carries rank 0's JSON line and nothing else (gloo's connection chatter goes to stderr)
        (sys.stdout if line.lstrip().startswith(b"{") else sys.stderr).buffer.write(line)
    sys.stdout.flush()
    return p.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="illumina150")
    ap.add_argument("--batch-reads", type=int, default=0, help="reads per resident batch (0 = the workload's default: 8 M short reads, 50 k long reads)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--long-read-len", type=int, default=5000, help="pacbio5k: mean read length (sd = a tenth of it, clipped to [a fifth, six times])")
    ap.add_argument("--db-nt", type=int, default=0, help="size of the synthetic DB (0 = the workload's: 140 Mnt; pacbio5k 14 Mnt; refs8 scales its 7 synthetic members by db_nt / 140 M)")
    ap.add_argument("--cpu-sample-reads", type=int, default=0, help="reads of the CPU-baseline sample (0 = the workload's default)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the reference CPU baseline (0 = min(host cores, 64))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cigar", action="store_true", help="skip the banded traceback (not the reference's behaviour)")
    ap.add_argument("--profile-run", action="store_true",
                    help="for rocprofv3 counter passes: only warm-up + timed steps (no exact-count pass, no PCIe leg, no CPU baseline), so every dispatch is a timed-path dispatch")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every rank aligns its own batches of --batch-reads reads; strong: ONE job of --total-reads reads (BASELINE configs[3]: 10 M) is split over the ranks, "
                         "a step = one pass of every rank over its shard")
    ap.add_argument("--total-reads", type=int, default=10_000_000, help="--scaling strong: reads of the whole job")
    ap.add_argument("--n1-value", type=float, default=0.0, help="--scaling strong: the value of the same command at --gpus 1; the line then carries efficiency_vs_n1 = value / (N x that)")
    ap.add_argument("--resident-batches", type=int, default=0,
                    help="distinct read batches kept resident in HBM (1..%d); step i runs on batch i %% this; 0 = as many as hold 16 M reads, between 2 and 8" % MAX_RESIDENT)
    args = ap.parse_args()

    t_process = time.time()                                  # set-up (DBs, index replica, resident batches) is timed per rank and reported: config.setup_s
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))          # `python bench.py --gpus N` as typed: this process becomes the launcher of N ranks
    if world != args.gpus:
        args.gpus = world
    if world > 1:                                            # the ranks of one job share this host: every rank's loaders / builders / packers get cores / N threads
        os.environ.setdefault("SMR_HOST_THREADS", str(max(1, (os.cpu_count() or 8) // world)))
    args.steps = max(args.steps, 1)
    args.warmup = max(args.warmup, 0)
    W = WORKLOADS[args.workload]
    args.batch_reads = args.batch_reads or W["batch_reads"]
    if args.scaling == "strong":
        units = 8 if 8 % world == 0 else world
        args.batch_reads = (args.total_reads // units) * (units // world) + args.total_reads % units      # the last rank's shard (the largest); config.batch_reads reports it
        args.resident_batches = 1
    args.cpu_sample_reads = args.cpu_sample_reads or W["cpu_sample_reads"]
    args.db_nt = args.db_nt or (14_000_000 if args.workload == "pacbio5k" else 3_770_000 if args.workload == "config2" else 140_000_000)
    if args.resident_batches <= 0:
        args.resident_batches = max(2, min(8, 16_000_000 // max(args.batch_reads, 1)))

    import numpy as np
    import torch
    import sortmerna_amd as smr
    from sortmerna_amd import shard, synth

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU fallback of the product path)")
    # test knobs (single-GPU dry run of the multi-rank path): SMR_BENCH_BACKEND=gloo puts the tiny collectives on the CPU,
    # SMR_BENCH_DEVICE=k maps every rank to GPU k
    backend = os.environ.get("SMR_BENCH_BACKEND", "nccl")
    if "SMR_BENCH_DEVICE" in os.environ:
        local = int(os.environ["SMR_BENCH_DEVICE"])
    cdev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)

    def barrier():
        if dist is not None:
            if backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    # ---------------- workload: DBs (rank 0 writes the FASTA files) + indexes (every rank builds its replicas on its own GPU) ----------------
    cache = os.path.join(tempfile.gettempdir(), "smr_bench_%s_%d" % (args.workload, args.db_nt))
    os.makedirs(cache, exist_ok=True)
    t0 = time.time()
    dbl = workload_dbs(args, synth, cache, rank)
    dbs = [p for _, p in dbl]
    log("%d DB file(s) ready (%.1fs)" % (len(dbs), time.time() - t0))
    barrier()
    t0 = time.time()
    eng = smr.Engine(local)
    # sorting / ids / positions / mini-tries as device kernels (seconds for 140 Mnt).  With several ranks every rank builds
    # the same indexes from the same FASTA files on its own GPU, concurrently: no index files, no N-fold host parsing of them
    parts_per_db, idx_slots, infos = [], [], []
    built = "device (smr_index_build_gpu)"
    for db in dbs:
        try:
            parts = smr.Index.build_gpu(eng, db, 18, 3072.0, 10000)
        except smr.SmrError as e:
            log("device index build failed (%s): host builder" % e)
            parts = smr.Index.build(db, 18, 3072.0, 10000, 0)
            built = "host (smr_index_build)"
        sl = []
        for ix in parts:                                     # every part of every --ref stays resident (the engine has 64 slots)
            slot = sum(len(x) for x in idx_slots) + len(sl)
            eng.upload_index(ix, slot)
            sl.append(slot)
        parts_per_db.append(parts)
        idx_slots.append(sl)
        infos.append(parts[0].info())
    index_built = "%s, %.1f s for %d DB(s)" % (built, time.time() - t0, len(dbs))
    parts = [ix for pp in parts_per_db for ix in pp]
    info = infos[0]
    log("indexes ready: %d DB(s), %d part(s), tries %.0f MB, positions %.0f MB, %d refs (%.1fs)" % (
        len(dbs), len(parts), sum(p.info().trie_words for p in parts) * 4 / 1e6, sum(p.info().n_pos for p in parts) * 8 / 1e6,
        sum(int(i.numseq) for i in infos), time.time() - t0))

    # ---------------- reads: W + K different batches per rank, resident in HBM ----------------
    t0 = time.time()
    codes, offs = load_all_codes(synth, dbs)
    n_total = args.warmup + args.steps              # step i (warmup first, then timed) runs on resident batch i % nb
    nb = max(1, min(n_total, args.resident_batches, MAX_RESIDENT))
    sample0 = None
    tot_reads = tot_len = 0
    min_len, max_len = 1 << 30, 0
    import ctypes as C
    shard_reads = []                                         # reads of every resident batch of this rank
    for b in range(nb):
        if args.scaling == "strong":
            blob, o = make_shard(args, synth, codes, offs, rank, world)
        else:
            blob, o = make_batch(args, synth, codes, offs, args.batch_reads, 1234 + 1000 * rank + b)
        nrb = len(o) - 1
        shard_reads.append(nrb)
        if b == 0:
            ns = min(nrb, args.cpu_sample_reads)
            sample0 = (blob[:int(o[ns])], o[:ns + 1].copy())
        h = C.c_void_p()
        rc = eng.L.smr_reads_pack(blob, o.ctypes.data, nrb, C.byref(h))
        assert rc == 0
        r = smr.Reads(h)
        eng.select_batch(b)
        eng.upload_reads(r, 1)
        tot_reads += r.count
        tot_len += r.total_len
        min_len, max_len = min(min_len, r.min_len), max(max_len, r.max_len)
        if b == nb - 1:
            last_packed = r                                  # kept on the host for the PCIe-inclusive measurement below
        else:
            r.free()
    del codes, offs
    log("%d batch(es) of %s reads resident, %.0f nt per read (%.1fs)" % (nb, "/".join(str(x) for x in sorted(set(shard_reads))), tot_len / max(tot_reads, 1), time.time() - t0))
    reads_per_step = shard.reduce_counters([shard_reads[0]], device=cdev)[0] if args.scaling == "strong" else args.gpus * args.batch_reads

    # C1: global read totals -> the same minimal_score (per DB) on every rank (refstats.cpp:247-265)
    g_reads, g_len, _, _ = shard.global_read_totals(tot_reads, tot_len, min_len, max_len, device=cdev)
    gumbel = GUMBEL
    if args.workload == "config2":                           # the reference's own Gumbel parameters for this DB (its log: tests/golden/config2/config2.json)
        g2 = json.load(open(os.path.join(CONFIG2_DIR, "config2.json")))["runs"]["default"]
        gumbel = (g2["lambda"], g2["K"])
    mss = [smr.minimal_score(gumbel[0], gumbel[1], i, g_reads, g_len) for i in infos]
    plist = [smr.default_params(minimal_score=m) for m in mss]
    ms = mss[0]

    def step(b):
        eng.select_batch(b)
        eng.reset_state()
        smr.align_resident(eng, idx_slots, plist, with_cigar=not args.no_cigar)

    setup_s = time.time() - t_process                       # this rank: process start -> everything resident, before the first step
    setup_max = shard.time_max(setup_s, device=cdev)
    setup_min = -shard.time_max(-setup_s, device=cdev)
    log("set-up of this rank %.1f s (ranks: %.1f .. %.1f s)" % (setup_s, setup_min, setup_max))
    for i in range(args.warmup):
        step(i % nb)
    timed = [i % nb for i in range(args.warmup, n_total)]        # the resident batch of every timed step
    uses = [timed.count(b) for b in range(nb)]
    # Algorithmic bytes of the timed steps: a workload property, counted once per distinct batch by the per-lane DFS seed kernel whose
    # work counters follow the reference's sequential scan exactly (untimed; the timed steps use the pigeonhole kernel, same results).
    eng.set_seed_mode(1)
    exact = [0] * 6
    exact_aligned = None if args.profile_run else 0
    for b in range(nb):
        if uses[b] == 0 or args.profile_run:
            continue
        eng.prof_reset()
        step(b)
        pe = eng.prof()
        for k, v in enumerate([pe.n_windows, pe.n_lookup, pe.n_node, pe.n_entry, pe.n_hit, pe.n_read_bytes]):
            exact[k] += int(v) * uses[b]
        exact_aligned += eng_counters_aligned(eng, b) * uses[b]
    eng.set_seed_mode(0)
    # N > 1, weak scaling: the N = 1 configuration of THIS run -- rank 0 alone, its GPU, its batches, the same timed steps, while the other ranks wait --
    # so that the line carries efficiency_vs_n1 = value / (N x that) from one process tree and one DB cache (the driver computes its own from its N = 1 run)
    n1_value = None
    if world > 1 and args.scaling == "weak" and not args.profile_run:
        barrier()
        if rank == 0:
            t1 = time.perf_counter()
            for b in timed:
                step(b)
            torch.cuda.synchronize()
            n1_value = args.steps * args.batch_reads / (time.perf_counter() - t1)
            log("N = 1 configuration (rank 0 alone): %.3g reads/s" % n1_value)
    eng.prof_reset()
    barrier()
    t0 = time.perf_counter()
    for b in timed:
        step(b)
    barrier()
    dt = time.perf_counter() - t0
    dt = shard.time_max(dt, device=cdev)

    # C2: Readstats counters of the timed steps, summed over ranks (RCCL).  Every step starts from a reset state, so a batch's
    # counter block holds the counts of its last step; a batch used u times contributes u times.
    n_db = len(dbs)
    ctr = np.zeros(2 + n_db, dtype=np.int64)
    for b in range(nb):
        if uses[b] == 0:
            continue
        eng.select_batch(b)
        c = eng.counters(n_db)
        ctr += uses[b] * np.array([c["num_aligned"], c["num_short"]] + list(c["reads_matched_per_db"][:n_db]), dtype=np.int64)
    ctr_t = shard.reduce_counters(ctr.tolist(), device=cdev)
    pr = eng.prof()
    assert exact_aligned is None or int(ctr[0]) == exact_aligned, "the two seed kernels disagree on num_aligned: %d vs %d" % (int(ctr[0]), exact_aligned)
    kp = eng.prof_kernels()                                  # per kernel family: HIP-event ms, launches, algorithmic bytes (device counters)
    kp_names = list(kp)
    prof = torch.tensor([pr.seed_ms, pr.chain_ms, pr.trace_ms, pr.seed_launches, pr.chain_launches, pr.trace_launches] +
                        exact + [pr.n_sw_fwd, pr.n_sw_rev, pr.n_sw_cells, pr.n_sw_spec, pr.n_sw_spec_used] +
                        [x for k in kp_names for x in (kp[k]["ms"], kp[k]["launches"], kp[k]["bytes"])], dtype=torch.float64, device=cdev)
    if dist is not None:
        dist.all_reduce(prof)
    prof = [float(x) for x in prof.cpu()]
    kp = {k: {"ms": prof[17 + 3 * i], "launches": prof[18 + 3 * i], "bytes": prof[19 + 3 * i]} for i, k in enumerate(kp_names)}      # sums over ranks

    # informational: the same step when the boundary hands over a HOST buffer (packed batch -> smr_reads_upload: allocations + H2D over
    # PCIe + state reset), serial, no overlap with the previous batch.  Never `value`.
    t0 = time.perf_counter()
    for _ in range(0 if args.profile_run else 2):
        eng.select_batch(nb - 1)
        eng.upload_reads(last_packed, 1)
        smr.align_resident(eng, idx_slots, plist, with_cigar=not args.no_cigar)
    torch.cuda.synchronize()
    pcie_rate = None if args.profile_run else 2 * shard_reads[nb - 1] / (time.perf_counter() - t0)
    last_packed.free()

    if rank == 0:
        reads_timed = args.steps * int(reads_per_step)
        seed_ms, chain_ms, trace_ms, seed_l, chain_l, trace_l = prof[0:6]
        n_lookup, n_node, n_entry, n_hit, n_read_bytes = prof[7], prof[8], prof[9], prof[10], prof[11]
        # what the reference's traversal would have moved (SURVEY.md 8d formula on the exact counters of the DFS kernel): reported as an
        # EQUIVALENT rate only -- k_seed_pg does not perform that traversal, so this is not a roofline numerator
        b_ref = n_read_bytes + 12 * n_lookup + 16 * n_node + 8 * n_entry + 8 * n_hit
        # Roofline numerators: the algorithmic HBM bytes of the SHIPPED kernels, each counted by the kernel itself on the device
        # (include/smr_hip.h smr_prof_kernels), over that kernel's own HIP-event time in the timed steps.  Ranks run
        # concurrently: sums over ranks of bytes / sums of ms = the per-GPU rate.
        traffic_all, traffic_note = None, "no PMC measurement for these kernel sources and this workload (profiles/hbm_traffic.json)"
        try:
            sys.path.insert(0, os.path.join(HERE, "tools"))
            import pmc_traffic
            tj = json.load(open(os.path.join(HERE, "profiles", "hbm_traffic.json")))
            w = tj["workload"]
            if (w["read_len"], w["db_nt"], w["batch_reads"]) == (args.read_len, args.db_nt, args.batch_reads) and tj.get("kernel_src_sha") == pmc_traffic.kernel_src_sha():
                traffic_all = tj["per_launch_bytes"]
                traffic_note = "REPLAYED from profiles/hbm_traffic.json (a PMC pass of these kernel sources on this workload, not measured in this run): " + tj["note"]
        except Exception:
            pass
        sw16_pmc = None
        counters_all, counters_note = None, "no counter pass for these kernel sources and this workload (profiles/sq_counters.json)"
        try:
            import pmc_traffic
            sj = json.load(open(os.path.join(HERE, "profiles", "sq_counters.json")))
            w = sj["workload"]
            if (w["read_len"], w["db_nt"], w["batch_reads"]) == (args.read_len, args.db_nt, args.batch_reads) and sj.get("kernel_src_sha") == pmc_traffic.kernel_src_sha():
                counters_all = sj["per_kernel"]
                counters_note = "REPLAYED from profiles/sq_counters.json (counter passes of these kernel sources on this workload, not measured in this run): " + sj["source"]
                sw16_pmc = sj.get("k_sw16")
        except Exception:
            pass
        seed_k = [k for k in kp if k.startswith("k_seed")]
        rk = {}
        for k in seed_k:
            v = kp[k]
            if v["launches"] <= 0 or v["ms"] <= 0:
                continue
            gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            rk[k] = {"avg_launch_ms": v["ms"] / v["launches"], "launches": v["launches"] / args.gpus, "algorithmic_bytes_per_launch": v["bytes"] / v["launches"],
                     "achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "traffic": (traffic_all or {}).get(k), "counters": (counters_all or {}).get(k)}
        dom = max(rk, key=lambda k: rk[k]["avg_launch_ms"] * rk[k]["launches"]) if rk else None
        st_bytes = sum(kp[k]["bytes"] for k in seed_k)
        st_ms = sum(kp[k]["ms"] for k in seed_k)
        roof = {"kernel": dom, "bound": "hbm", "achieved": rk[dom]["achieved"] if dom else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (rk[dom]["achieved"] / HBM_PEAK_GBS) if dom else 0.0, "traffic": rk[dom]["traffic"] if dom else None, "traffic_note": traffic_note,
                # the issue side of the same kernel (rocprofv3 derived metrics, tools/pmc_sq.py): % of cycles its vector / scalar ALUs issue, lane utilisation, LDS bank conflicts
                "valu_issue_frac": (rk[dom]["counters"]["VALUBusy"] / 100.0) if dom and rk[dom]["counters"] and "VALUBusy" in rk[dom]["counters"] else None,
                "lds_bank_conflict_frac": (rk[dom]["counters"]["LDSBankConflict"] / 100.0) if dom and rk[dom]["counters"] and "LDSBankConflict" in rk[dom]["counters"] else None,
                "counters": rk[dom]["counters"] if dom else None, "counters_note": counters_note,
                "other_kernels_counters": {k: v for k, v in (counters_all or {}).items() if not k.startswith("k_seed")} or None,
                "algorithmic_bytes_per_launch": rk[dom]["algorithmic_bytes_per_launch"] if dom else 0.0, "avg_launch_ms": rk[dom]["avg_launch_ms"] if dom else 0.0,
                "kernels": rk,
                "seed_stage": {"achieved": st_bytes / max(st_ms * 1e-3, 1e-12) / 1e9, "frac": st_bytes / max(st_ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_launch": st_bytes / max(seed_l, 1), "avg_launch_ms": st_ms / max(seed_l, 1), "bytes_per_read": st_bytes / reads_timed},
                "equivalent_rate": {"GB/s": (b_ref / max(st_ms * 1e-3, 1e-12) / 1e9) if b_ref else None, "bytes_per_read": b_ref / reads_timed,
                                    "note": "bytes the REFERENCE's trie traversal would move for these reads (SURVEY.md 8d formula on exact counters) / the seed stage's time; "
                                            "not a roofline figure: the shipped search reaches the same hits without that traversal"},
                "note": "kernel = the seed-stage kernel with the most time in the timed steps; achieved = the algorithmic HBM bytes THAT kernel counts for itself on the device "
                        "(tuples, directory words, strings looked at, accepted {rank,id}, hit segments; smr_prof_kernels) / its own HIP-event time; kernels{} has every seed-stage "
                        "kernel, seed_stage their sum"}
        # VALU model of the Smith-Waterman kernel: a wave64 VALU instruction occupies a SIMD for 4 cycles ->
        # 256 CU x 4 SIMD x 2.4 GHz / 4 = 6.14e11 wave-instructions/s; one systolic step costs 13 R + 14 (packed, R = ceil(m/128) cell pairs)
        # or 20 R + 15 (32-bit, R = ceil(m/64) cells) instructions and there are n + ceil(m/R) - 1 steps for an m x n problem
        mean_len = int(round(g_len / max(g_reads, 1)))
        m_sw, n_sw = mean_len, mean_len + 8
        if eng.sw_mode() >= 1:
            if m_sw > 512:
                # (reads beyond 512 letters: strips of 128 virtual lanes x R rows, R = the cheapest of 8, 10 .. 24 by the step cost counted in the
                # disassembly, 14 R + 35 instructions, n + 127 steps per strip: sw_long_rows in csrc/smr_chain.hpp)
                r_sw = min(range(8, 25, 2), key=lambda r: ((m_sw + 128 * r - 1) // (128 * r)) * (14 * r + 35))
                instr = ((m_sw + 128 * r_sw - 1) // (128 * r_sw)) * (n_sw + 127) * (14 * r_sw + 35)
            else:
                r_sw = min((m_sw + 127) // 128, 4)
                instr = (n_sw + (m_sw + r_sw - 1) // r_sw - 1) * (13 * r_sw + 14)
        else:
            r_sw = (m_sw + 63) // 64
            instr = (n_sw + 63) * (20 * r_sw + 15)
        sw_peak_gcups = m_sw * n_sw / (instr / 6.144e11) / 1e9
        # the four-problem kernel (sw_wave_x4): 32 virtual lanes per problem, R = 3 / 5 / 8 rows each, n + ceil(m/R) - 1 steps of 13 R + 13 instructions for FOUR problems
        r4 = 3 if m_sw <= 96 else (5 if m_sw <= 160 else 8)
        instr4 = (n_sw + (m_sw + r4 - 1) // r4 - 1) * (13 * r4 + 13)
        sw4_peak_gcups = 4 * m_sw * n_sw / (instr4 / 6.144e11) / 1e9 if m_sw <= 256 else None
        # the sixteen-problem kernel of the split walk (k_sw16, smr_walk.hpp): 8 virtual lanes per problem, R = 13 / 19 / 32 rows each (by the longest
        # read of the batch), n + 7 steps of 13 R + 33 instructions (end cells; 10 R + 28 where only the score is asked) for SIXTEEN problems
        r16 = 13 if max_len <= 104 else (19 if max_len <= 152 else (26 if max_len <= 208 else 32))      # (the instantiations k_sw16<13|19|26|32> the host picks from)
        sw16_peak_gcups = 16 * m_sw * n_sw / ((n_sw + 7) * (13 * r16 + 33) / 6.144e11) / 1e9 if max_len <= 256 else None
        out = {
            "metric": {"illumina150": "reads/sec (150 bp vs smr_v4.3_default_db-sized DB)", "refs8": "reads/sec (150 bp vs the 8-ref rRNA set)",
                       "pacbio5k": "reads/sec (5 kb PacBio-like reads vs a 28S-like DB, every alignment with its CIGAR)",
                       "config2": "reads/sec (set2 amplicon reads vs silva-arc-16s-id95, classify-only)"}[args.workload], "value": reads_timed / dt, "unit": "reads/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u8/i32",
            "data": "real (the reference's bundled fixture, the read set repeated to fill a batch)" if args.workload == "config2" else "synthetic",
            "config": {"workload": {
                "illumina150": "BASELINE configs[2]: synthetic 150-nt Illumina-like reads (10%% from DB, 90%% background) vs seeded synthetic rRNA-like DB "
                               "of %d nt standing in for smr_v4.3_default_db.fasta (absent offline); " % args.db_nt,
                "refs8": "BASELINE configs[3]: the same reads (10%% from the union of the DBs) vs EIGHT resident reference DBs: the bundled silva-arc-16s-id95 + 7 seeded synthetic "
                         "families sized like the rRNA_databases set (%s; the real files are absent offline); " % ", ".join("%s %.1f Mnt" % (n, int(i.full_len) / 1e6) for (n, _), i in zip(dbl, infos)),
                "config2": "BASELINE configs[1] on real data: the reference's bundled set2_environmental_study_550_amplicon reads (100 000 reads of 50-200 nt; a batch = the set %d times over) "
                           "vs its bundled silva-arc-16s-id95 (the id85 file of the config is not in the repository); " % max(1, args.batch_reads // 100_000),
                "pacbio5k": "BASELINE configs[4]: synthetic PacBio-like reads ~N(%d, %d) nt," % (args.long_read_len, args.long_read_len // 10) + " 12%% errors (6%% ins, 4%% del, 2%% sub), all sampled from a seeded synthetic 28S-like DB "
                            "of %d nt (silva-euk-28s-id98 is absent offline); " % args.db_nt}[args.workload] + W["options"],
                       "name": args.workload, "batch_reads": args.batch_reads, "resident_batches": nb, "read_len": (args.read_len if args.workload != "pacbio5k" else mean_len), "db_nt": args.db_nt, "index_parts": len(parts),
                       "n_dbs": n_db, "db_seqs": sum(int(i.numseq) for i in infos), "minimal_score": [int(x) for x in mss] if n_db > 1 else int(ms), "sharding": "reads, %d rank(s), index replicated" % args.gpus, "nranks": (dist.get_world_size() if dist is not None else 1),
                       "collectives": {"backend": (backend if dist is not None else None), "what": "all-reduce of the read totals before, of the Readstats counters after; none on the data path",
                                       "rccl_version": (".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None)},
                       "total_reads": (args.total_reads if args.scaling == "strong" else None),
                       "setup_s": {"slowest_rank": setup_max, "fastest_rank": setup_min,
                                   "what": "process start -> DB files, index replica built on the rank's GPU and resident batches generated + uploaded, all ranks concurrently on the shared host"},
                       "cigar": not args.no_cigar, "index_build": index_built,
                       "sw_kernel": ("packed 16-bit (v_pk): the candidate walk in rounds, its windows scored sixteen per wave by k_sw16 (reads <= 256 nt with a k_cand record); "
                                     "four per wave / single problems on 128 virtual lanes inside k_chain for the others") if eng.sw_mode() >= 1 else "32-bit",
                       "walk_rounds_per_pass": eng.walk_rounds(), "hit_list_entries_per_search": int(eng.prof().hit_list_cap)},
            "pcie_inclusive_reads_per_s_per_gpu": pcie_rate,
            "counters": {"reads": reads_timed, "num_aligned": int(ctr_t[0]), "num_short": int(ctr_t[1]), "reads_matched_per_db": [int(x) for x in ctr_t[2:2 + n_db]]},
            "work_per_read": {"windows": prof[6] / reads_timed, "lookups": n_lookup / reads_timed, "nodes": n_node / reads_timed,
                              "entries": n_entry / reads_timed, "hits": n_hit / reads_timed},
            "roofline": roof,
            "kernels": {"k_seed": {"ms": seed_ms / args.gpus, "launches": seed_l / args.gpus},
                        "k_chain": {"ms": chain_ms / args.gpus, "launches": chain_l / args.gpus,
                                    "sw_fwd": prof[12], "sw_rev": prof[13], "sw_cells": prof[14], "k_sw16_valu_per_cell_pair_pmc": sw16_pmc, "gcups": prof[14] / max(chain_ms / args.gpus, 1e-9) / 1e6,
                                    "valu_model_peak_gcups": sw_peak_gcups * args.gpus,
                                    "valu_model_x4_peak_gcups": sw4_peak_gcups * args.gpus if sw4_peak_gcups else None,
                                    "valu_model_x16_peak_gcups": sw16_peak_gcups * args.gpus if sw16_peak_gcups else None,
                                    "valu_model_frac_x16": (prof[14] / max(chain_ms / args.gpus, 1e-9) / 1e6 / (sw16_peak_gcups * args.gpus)) if sw16_peak_gcups else None,
                                    "round_kernels_ms_per_step": {k: kp[k]["ms"] / args.gpus / args.steps for k in ("k_cand", "k_walk", "k_sw16", "k_wnext", "k_chain", "k_begins") if k in kp},
                                    # against the kernel that scores most windows: the four-problem kernel where reads fit it (<= 256 nt), else the single-problem strips
                                    "valu_model_frac": prof[14] / max(chain_ms / args.gpus, 1e-9) / 1e6 / ((sw4_peak_gcups or sw_peak_gcups) * args.gpus),
                                    "valu_model_frac_single_problem_kernel": prof[14] / max(chain_ms / args.gpus, 1e-9) / 1e6 / (sw_peak_gcups * args.gpus),
                                    "sw_scored_ahead": prof[15], "sw_scored_ahead_used": prof[16],
                                    "note": "gcups = DP cells of the sequential walk's ssw_align calls / whole k_chain time (candidate search, LIS, bookkeeping included); "
                                            "valu_model_* = what the single-problem / four-problem packed kernel alone could do at one VALU op per SIMD per 4 cycles; valu_model_frac is against the "
                                            "four-problem kernel's model when the reads fit it"},
                        "k_trace": {"ms": trace_ms / args.gpus, "launches": trace_l / args.gpus}},
        }
        if args.scaling == "strong" and args.n1_value > 0:
            out["efficiency_vs_n1"] = out["value"] / (args.gpus * args.n1_value)
        if n1_value:
            out["n1_value_rank0_alone"] = n1_value
            out["efficiency_vs_n1"] = out["value"] / (args.gpus * n1_value)
        if args.gpus == 1 and not args.no_cpu_baseline and not args.profile_run:
            try:
                out["cpu_baseline"] = cpu_baseline(args, dbs, parts_per_db, sample0, smr, eng, idx_slots)
            except Exception as e:  # the baseline must never lose the measured GPU number
                out["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": os.cpu_count(), "kind": "reference", "sample": "failed: %r" % (e,)}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
