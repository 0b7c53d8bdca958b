"""Scratch driver for the first GPU bring-up (prints mismatch details). Usage: python tools_gpu_debug.py"""
import json
import os
import sys
import time
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
import sortmerna_amd as smr
from helpers import refrun
from helpers.workload import Workload

os.makedirs("gpurun_out", exist_ok=True)
t0 = time.time()
w = Workload(tempfile.mkdtemp(), db_nt=300_000, n_reads=4000, seed=5)
print("workload built in %.1fs, minimal_score %d, parts %d" % (time.time() - t0, w.minimal_score, w.stats.nparts), flush=True)
t0 = time.time()
recs_o, ctr_o = w.oracle_records()
print("oracle %.1fs" % (time.time() - t0), ctr_o, flush=True)
e = smr.Engine(0)
t0 = time.time()
recs_g, ctr_g = w.gpu_records(e, with_cigar=True)
print("gpu %.2fs" % (time.time() - t0), ctr_g, flush=True)
p = e.prof()
print("prof seed_ms %.3f (%d) chain_ms %.3f (%d) trace_ms %.3f (%d) windows %d lookup %d node %d entry %d hit %d swf %d swr %d" % (
    p.seed_ms, p.seed_launches, p.chain_ms, p.chain_launches, p.trace_ms, p.trace_launches, p.n_windows, p.n_lookup, p.n_node, p.n_entry,
    p.n_hit, p.n_sw_fwd, p.n_sw_rev))
bad = [i for i in range(len(recs_o)) if recs_o[i] != recs_g[i]]
print("mismatching records: %d of %d (oracle aligned %d)" % (len(bad), len(recs_o), sum(1 for r in recs_o if r)))
for i in bad[:8]:
    print("read", i, "len", len(w.seqs[i]))
    print("  gpu", refrun.parse_record(recs_g[i]))
    print("  orc", refrun.parse_record(recs_o[i]))
json.dump({"bad": len(bad), "n": len(recs_o)}, open("gpurun_out/debug1.json", "w"))
