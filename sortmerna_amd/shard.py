"""Read sharding across GPUs: reads are independent given (index, references, minimal_score), so the
host splits the record range, every rank keeps a full index replica, and there is NO data-path collective.  The two
collectives the reference's semantics need are tiny:
  C1 (before)  global read totals -> one minimal_score for all ranks   (Readstats ctor main.cpp:77-78, refstats.cpp:247-265)
  C2 (after)   Readstats counters summed over ranks                    (readstats.hpp:77-85)
Both go through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
import torch


def shard_range(n_records, rank, world):
    """Contiguous record range [first, first+count) of `rank`; the reference splits the read file the same way into one
    byte range per thread (readfeed.cpp:1253-1277), record-aligned."""
    base, rem = divmod(n_records, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def global_read_totals(count, total_len, min_len, max_len, device="cpu"):
    """C1: -> (all_reads_count, all_reads_len, min_read_len, max_read_len) over all ranks."""
    dist = _dist()
    if dist is None:
        return int(count), int(total_len), int(min_len), int(max_len)
    s = torch.tensor([count, total_len], dtype=torch.int64, device=device)
    # a rank with an empty shard must not win the min
    lo = torch.tensor([min_len if count else (1 << 62)], dtype=torch.int64, device=device)
    hi = torch.tensor([max_len], dtype=torch.int64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    mn = int(lo[0])
    return int(s[0]), int(s[1]), (0 if mn == (1 << 62) else mn), int(hi[0])


def reduce_counters(values, device="cpu"):
    """C2: element-wise sum over ranks of a list of non-negative integer counters."""
    dist = _dist()
    t = torch.tensor(list(values), dtype=torch.int64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.cpu()]


def time_max(seconds, device="cpu"):
    dist = _dist()
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])
