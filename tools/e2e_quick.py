"""The end-to-end run of tools/e2e_cpp.py for a GPU session with well under a minute left: 140 Mnt DB, index built on the device and written
as reference-format files while the 10 M-read FASTQ is being written, then ONE run of examples/build/smr_align_mgpu --fastx."""
import os
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import sortmerna_amd as smr  # noqa: E402
from sortmerna_amd import synth  # noqa: E402
from e2e_cpp import write_fastq_fast  # noqa: E402

T0 = time.time()
d = tempfile.mkdtemp(prefix="smr_e2eq_")
db = os.path.join(d, "db.fasta")
synth.make_db(db, 140_000_000, seed=42)
print("[prep] DB %.1f s" % (time.time() - T0), flush=True)
prefix = os.path.join(d, "idx")
fq = os.path.join(d, "reads.fastq")


def index():
    eng = smr.Engine(0)
    parts = smr.Index.build_gpu(eng, db, 18, 3072.0, 10000)
    smr.Index.write_files(parts, db, prefix)
    eng.close()
    print("[prep] index files %.1f s" % (time.time() - T0), flush=True)


def reads():
    codes, offs = synth.load_db_codes(db)
    done = 0
    while done < 10_000_000:
        letters = synth.make_reads(codes, offs, 2_000_000, read_len=150, frac_db=0.10, seed=1234 + done, sub=0.005, indel=0.0001, n_rate=0.001)
        write_fastq_fast(fq, letters, done)
        done += 2_000_000
    print("[prep] reads %.1f s" % (time.time() - T0), flush=True)


th = [threading.Thread(target=index), threading.Thread(target=reads)]
for x in th:
    x.start()
for x in th:
    x.join()
exe = os.path.join(ROOT, "examples", "build", "smr_align_mgpu")
flat = os.path.join(d, "flat")
# run 1: reference-format index files (the slow load), writes the flat cache beside the alignment; run 2: the same command, index from the cache
for run in (1, 2):
    out = os.path.join(d, "out%d" % run)
    os.makedirs(out)
    t = time.time()
    p = subprocess.run([exe, "--ref", db, "--idx", prefix, "--idx-flat", flat, "--gumbel", "0.618874", "0.343238", "--reads", fq, "--out", out, "--gpus", "1", "--chunk-reads", "2000000", "--fastx"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, SMR_IB_TIMING="1"))
    print(p.stdout.decode()[-3000:])
    print("[run %d] wall %.1f s, rc %d; total %.1f s" % (run, time.time() - t, p.returncode, time.time() - T0), flush=True)
