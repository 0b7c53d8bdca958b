"""bench.py's whole control flow (workload generation, resident batches, counting pass, timed steps, JSON line, CPU-baseline leg)
executed without a GPU: torch.cuda's three calls are stubbed and the binding is routed to the kernel emulator (tests/emu), on a tiny
workload.  Checks the contract of the JSON line, not any number in it."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import pytest

from helpers import emu, paths

KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline", "cpu_baseline"]


@pytest.mark.parametrize("steps,warmup,workload", [(2, 1, "illumina150"), (20, 5, "illumina150"), (1, 0, "refs8"), (2, 1, "pacbio5k"), (1, 1, "config2")])          # (20, 5) = the driver's own command line of round 1, which aborted
def test_bench_control_flow_on_the_emulator(monkeypatch, tmp_path, steps, warmup, workload):
    baseline = paths.have_ref_bin() and steps <= 2               # with the reference binary at hand the CPU-baseline leg runs too
    import torch
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    monkeypatch.setenv("SMR_BENCH_BACKEND", "gloo")          # the (world-size-1) reductions on CPU tensors
    import tempfile
    monkeypatch.setattr(tempfile, "tempdir", str(tmp_path))
    nreads = 60 if workload == "pacbio5k" else (300 if steps > 2 else 1500)      # (config2: the first 1500 reads of the bundled read set vs the whole bundled DB)
    argv = ["bench.py", "--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--batch-reads", str(nreads), "--db-nt", "1000000" if workload == "refs8" else "150000",
            "--cpu-sample-reads", str(nreads), "--cpu-threads", "2", "--workload", workload, "--long-read-len", "600"]
    if not baseline:
        argv.append("--no-cpu-baseline")
    monkeypatch.setattr(sys, "argv", argv)
    sys.path.insert(0, paths.REPO)
    import importlib
    bench = importlib.import_module("bench")
    buf = io.StringIO()
    with emu.active(), redirect_stdout(buf):
        bench.main()
    line = [l for l in buf.getvalue().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    for k in KEYS:
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == steps and out["warmup"] == warmup and out["unit"] == "reads/s" and out["value"] > 0
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert "workload" in out["config"] and "sw_kernel" in out["config"] and out["config"]["index_build"].startswith("device")
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert out["counters"]["reads"] == steps * nreads and out["config"]["name"] == workload
    assert out["data"].startswith("real" if workload == "config2" else "synthetic")
    if workload == "refs8":
        assert out["config"]["n_dbs"] == 8 and len(out["counters"]["reads_matched_per_db"]) == 8 and sum(out["counters"]["reads_matched_per_db"]) == out["counters"]["num_aligned"] > 0
    assert out["config"]["resident_batches"] == min(steps + warmup, 8)       # (tiny batches: the automatic choice is 8)
    assert abs(out["ms_per_step"] * steps / 1e3 * out["value"] - steps * nreads) < 1e-3
    assert out["kernels"]["k_chain"]["valu_model_peak_gcups"] > 0
    if baseline:
        cb = out["cpu_baseline"]
        assert cb["kind"] == "reference" and cb["unit"] == "reads/s" and cb["cores"] == 2 and cb["value"] and cb["value"] > 0, cb
        assert cb["parity"]["aligned_read_ids_equal"] is True and cb["parity"]["reference_aligned"] == cb["parity"]["gpu_aligned"] > 0, cb
        assert cb["parity"]["records_equal"] is True and cb["parity"]["reference_records"] == cb["parity"]["gpu_aligned"], cb
    else:
        assert out["cpu_baseline"] is None


def test_unwrapped_multi_rank_command(tmp_path):
    """`python bench.py --gpus 2 --steps 2 --warmup 1` exactly as the driver types it -- no torchrun around it: bench.py must start its own
    two ranks (round 2 exited with a usage message here).  Ranks on the emulator, collectives on gloo; one JSON line, from rank 0."""
    import subprocess
    emu.build()
    env = dict(os.environ, TMPDIR=str(tmp_path), SMR_BENCH_BACKEND="gloo", SMR_BENCH_DEVICE="0")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(paths.REPO, "tests", "helpers", "bench_on_emu.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch-reads", "1200", "--db-nt", "150000", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    for k in KEYS:
        assert k in out, k
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["counters"]["reads"] == 2 * 2 * 1200 and out["cpu_baseline"] is None
    assert out["config"]["nranks"] == 2
    # the N = 1 configuration timed by rank 0 alone in the same run (round 6): efficiency from one process tree
    assert out["n1_value_rank0_alone"] > 0 and abs(out["efficiency_vs_n1"] - out["value"] / (2 * out["n1_value_rank0_alone"])) < 1e-9
    r = out["roofline"]
    assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12


@pytest.mark.parametrize("ranks", [1, 2])
def test_strong_scaling_splits_one_job_over_the_ranks(tmp_path, ranks):
    """`--scaling strong --total-reads T`: the job is the same T reads whatever the number of ranks (8 seeded units, contiguous unit ranges per
    rank); the line says "strong", counts T reads per step, and the Readstats counters of the 2-rank run equal those of the 1-rank run."""
    import subprocess
    emu.build()
    env = dict(os.environ, TMPDIR=str(tmp_path), SMR_BENCH_BACKEND="gloo", SMR_BENCH_DEVICE="0")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(paths.REPO, "tests", "helpers", "bench_on_emu.py"), "--gpus", str(ranks), "--steps", "1", "--warmup", "0",
           "--scaling", "strong", "--total-reads", "2403", "--db-nt", "150000", "--no-cpu-baseline", "--n1-value", "1000.0"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["scaling"] == "strong" and out["n_gpus"] == ranks and out["counters"]["reads"] == 2403 and out["config"]["total_reads"] == 2403
    assert abs(out["efficiency_vs_n1"] - out["value"] / (ranks * 1000.0)) < 1e-9
    ref = tmp_path.parent / "strong_num_aligned.txt"         # the first of the two runs leaves its count for the second
    if ref.exists():
        assert int(ref.read_text()) == out["counters"]["num_aligned"] > 0
    else:
        ref.write_text(str(out["counters"]["num_aligned"]))


def test_the_traffic_file_belongs_to_the_seed_kernels_in_the_tree():
    """bench.py fills roofline.traffic from profiles/hbm_traffic.json only when the file's hash of the seed-stage sources is the tree's
    (tools/pmc_traffic.py).  A stale file is not an error of the code -- the bench then prints traffic: null with a note -- so this only
    warns: re-run `tools/gpu_session_r6.sh <tag> pmc` on the GPU box and copy <tag>/hbm_traffic.json into profiles/."""
    import json
    import sys
    import warnings
    sys.path.insert(0, os.path.join(paths.REPO, "tools"))
    import pmc_traffic
    tj = json.load(open(os.path.join(paths.REPO, "profiles", "hbm_traffic.json")))
    assert set(tj["workload"]) == {"batch_reads", "read_len", "db_nt"} and tj["per_launch_bytes"]
    if tj["kernel_src_sha"] != pmc_traffic.kernel_src_sha():
        warnings.warn("profiles/hbm_traffic.json was measured on other seed-stage sources (%s, tree: %s): bench.py will print roofline.traffic = null"
                      % (tj["kernel_src_sha"], pmc_traffic.kernel_src_sha()))
