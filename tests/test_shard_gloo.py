"""CPU, world_size 2 over gloo: the N > 1 path.  Each rank takes its record range of the golden read file, the ranks
agree on minimal_score through the C1 all-reduce, align their shard (the oracle stands in for the GPU kernels -- this
is a test of the sharding logic), and the C2 all-reduce of the counters plus the concatenated per-read records must
equal the single-process reference results."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import sortmerna_amd as smr
    from sortmerna_amd import shard
    from helpers import golden, orc
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    g = golden.load()["syn_default"]
    db, reads_path, all_seqs = golden.inputs("syn_default")
    first, count = shard.shard_range(len(all_seqs), rank, world)
    r = smr.Reads.from_fastx(reads_path, first, count) if count else smr.Reads.from_seqs([])
    assert r.count == count
    tot = shard.global_read_totals(r.count, r.total_len, r.min_len, r.max_len)
    parts = smr.Index.build(db, 18, 3072.0, 10000, 1)
    prefix = os.path.join(tmp, "idx_rank%d" % rank)
    smr.Index.write_files(parts, db, prefix)
    ms = smr.minimal_score(g["log"]["lambda"][0], g["log"]["K"][0], parts[0].info(), tot[0], tot[1])
    st = orc.load_stats(prefix)
    run = orc.Run(all_seqs[first:first + count])
    run.align_part(prefix, db, st, 0, orc.default_params(minimal_score=ms))
    ctr = shard.reduce_counters([run.counters.num_aligned, run.counters.num_short, run.counters.reads_matched_per_db[0]])
    q.put((rank, first, count, tot, ms, ctr, run.records()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_read_sharding_world_size_n(tmp_path, world):
    from helpers import golden
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = golden.load()["syn_default"]
    exp = golden.records("syn_default")
    recs = []
    for rank, first, count, tot, ms, ctr, rr in got:
        assert first == len(recs) and count == len(rr)
        assert tot[:2] == (g["readstats"]["all_reads_count"], g["readstats"]["all_reads_len"])
        assert tot[2:] == (g["readstats"]["min_read_len"], g["readstats"]["max_read_len"])
        assert ms == g["log"]["minimal_score"][0]
        assert ctr == [g["readstats"]["num_aligned"], g["readstats"]["num_short"], g["readstats"]["reads_matched_per_db"][0]]
        recs += rr
    assert recs == exp


def test_shard_range_covers_everything():
    from sortmerna_amd import shard
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            rs = [shard.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and sum(c for _, c in rs) == n
            for (f0, c0), (f1, _) in zip(rs, rs[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in rs) - min(c for _, c in rs) <= 1
