"""tests/golden/score_split.json: the minimal SW score the UNMODIFIED reference computes under -score_split (refstats.cpp:247-265: the read
totals divided by its number of processing threads), for the synthetic golden inputs with 1, 2, 4 and 7 threads -- next to the run without
the option.  Run in the build container (needs oracle/_ref/sortmerna_ref):   python tests/golden/make_golden_score_split.py"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from helpers import refrun  # noqa: E402

db, reads = os.path.join(HERE, "syn_db.fasta"), os.path.join(HERE, "syn_reads.fasta")
out = {"db": "syn_db.fasta", "reads": "syn_reads.fasta", "runs": []}
with tempfile.TemporaryDirectory() as d:
    idx = os.path.join(d, "idx")
    for threads, extra in [(1, []), (1, ["-score_split", "1"]), (2, ["-score_split", "1"]), (4, ["-score_split", "1"]), (7, ["-score_split", "1"])]:
        r = refrun.run_reference([db], [reads], os.path.join(d, "w%d_%d" % (threads, len(extra))), extra=extra, threads=threads, idx_dir=idx)
        assert r.rc == 0, r.stdout[-2000:]
        rs = refrun.parse_readstats(r.kvdb[b"Readstats"]) if b"Readstats" in r.kvdb else None
        out["runs"].append({"threads": threads, "score_split": bool(extra), "lambda": r.log["lambda"][0], "K": r.log["K"][0], "minimal_score": r.log["minimal_score"][0],
                            "num_aligned": r.log["num_aligned"], "all_reads_count": rs["all_reads_count"] if rs else None, "all_reads_len": rs["all_reads_len"] if rs else None})
        print(out["runs"][-1])
json.dump(out, open(os.path.join(HERE, "score_split.json"), "w"), indent=1)
