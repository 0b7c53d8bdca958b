"""One-off check on the kernel emulator (no GPU): 5-12.5 kb PacBio-like reads (indels, substitutions, both strands) against a
synthetic DB of 14 kb sequences: kernel sources == C oracle, record for record.  Covers the packed SW kernel with 512-row strips
(reads up to ~8 kb), the 32-bit kernel beyond, long tracebacks, and LDS sizes above 64 KB (whose launch attribute only the GPU can test).

    python tools/emu_long_reads_check.py
"""
import sys, time, tempfile
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import sortmerna_amd as smr
from sortmerna_amd import synth
from helpers import emu
from helpers.workload import Workload
emu.build()
with emu.active():
    w = Workload(tempfile.mkdtemp(prefix="smr_long_"), db_nt=400_000, n_reads=20, seed=43, family_size=4, mean_len=14000)
    codes, offs = synth.load_db_codes(w.db)
    rng = np.random.Generator(np.random.PCG64(7))
    seqs=[]
    for i in range(6):
        sq = int(rng.integers(0, len(offs) - 1))
        L = int(offs[sq+1]-offs[sq]); ln = min(L, [5000, 7000, 9000, 11000, 12500, 6000][i])
        st = int(offs[sq] + rng.integers(0, L - ln + 1))
        out=[]
        for c in codes[st:st+ln]:
            u = rng.random()
            if u < 0.03: continue
            if u < 0.06: out.append(int(rng.integers(0,4)))
            out.append(int((c + rng.integers(1,4)) & 3) if u > 0.96 else int(c))
        s = "".join("ACGT"[c] for c in out)
        if i % 2: s = s[::-1].translate(str.maketrans("ACGT","TGCA"))
        seqs.append(s)
    print("read lengths", [len(s) for s in seqs])
    w.seqs = seqs; w.reads = smr.Reads.from_seqs(seqs)
    w.minimal_score = smr.minimal_score(0.618874, 0.343238, w.parts[0].info(), len(seqs), sum(map(len, seqs)))
    eng = smr.Engine(0)
    t=time.time(); recs_o, ctr_o = w.oracle_records(); print("oracle %.1fs aligned %d"%(time.time()-t, ctr_o["num_aligned"]))
    t=time.time(); recs_g, ctr_g = w.gpu_records(eng); print("emu %.1fs aligned %d"%(time.time()-t, ctr_g["num_aligned"]))
    bad=[i for i in range(len(seqs)) if recs_o[i]!=recs_g[i]]
    print("bad", bad)
    from helpers import refrun
    for i in bad[:2]:
        print(refrun.parse_record(recs_g[i])); print(refrun.parse_record(recs_o[i]))
