"""The C++17 host driver over the C ABI (examples/smr_align.cpp): builds with plain g++ against include/smr_hip.h, fails loudly without a
GPU, and on the GPU box reproduces the reference's per-read records for a one-DB and a two-DB golden case."""
import os
import struct
import subprocess

import pytest

from helpers import golden, paths, refrun

EXE = os.path.join(paths.REPO, "examples", "build", "smr_align")


def build_driver():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    lib = os.path.join(paths.REPO, "sortmerna_amd", "lib")
    import sortmerna_amd.capi as capi
    capi.load()                                   # makes sure libsmr_hip.so is built
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", os.path.join(paths.REPO, "examples", "smr_align.cpp"),
                           "-I", os.path.join(paths.REPO, "include"), "-L", lib, "-lsmr_hip", "-Wl,-rpath," + lib, "-o", EXE])
    return EXE


def test_driver_builds_and_fails_loudly_without_gpu(tmp_path):
    exe = build_driver()
    assert "usage: smr_align" in subprocess.check_output([exe, "--help"]).decode()
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    db, rd, _ = golden.inputs("syn_default")
    p = subprocess.run([exe, "--ref", db, "--reads", rd, "--out", str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"--gumbel LAMBDA K is required" in p.stderr          # no silent default for the Gumbel parameters
    p = subprocess.run([exe, "--ref", db, "--gumbel", "0.6", "0.33", "--reads", rd, "--out", str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"no CPU fallback" in p.stderr


def _check_case(exe, case, tmp_path):
    g = golden.load()[case]
    dbs, rd, seqs = golden.inputs(case)
    if not isinstance(dbs, list):
        dbs = [dbs]
    cmd = [exe, "--reads", rd, "--out", str(tmp_path)]
    for k, db in enumerate(dbs):
        cmd += ["--ref", db, "--gumbel", repr(g["log"]["lambda"][k]), repr(g["log"]["K"][k])]
    if "num_alignments" in g["params"]:
        cmd += ["-num_alignments", str(g["params"]["num_alignments"])]
    if case == "syn_default":
        cmd += ["--fastx", "--other", "--sam", "--blast", "1 qstrand cigar"]
    subprocess.check_call(cmd)
    if case == "syn_default":            # the reference's own report files for the same options (row-exact except the e-value's last digit)
        sam = [l.rstrip("\n") for l in open(tmp_path / "aligned.sam") if not l.startswith("@")]
        assert sam == [l for l in g["sam"] if not l.startswith("@")]
        blast = [l.rstrip("\n").split("\t") for l in open(tmp_path / "aligned.blast")]
        exp_b = [l.split("\t") for l in g["blast"]]
        assert [a[:10] + a[11:] for a in blast] == [b[:10] + b[11:] for b in exp_b]
        assert [l.split()[0][1:] for l in open(tmp_path / "aligned.fa") if l.startswith(">")] == g["aligned_ids"]
        assert [l.split()[0][1:] for l in open(tmp_path / "other.fa") if l.startswith(">")] == g["other_ids"]
    kv = refrun.parse_kvdb_dump(str(tmp_path / "records.bin"))
    got = [kv.get(b"0_%d" % i, b"") for i in range(len(seqs))]
    exp = golden.records(case)
    bad = [i for i in range(len(seqs)) if got[i] != exp[i]]
    assert not bad, "%d records differ, first %d" % (len(bad), bad[0])
    summary = open(tmp_path / "summary.txt").read()
    assert "Total reads passing E-value threshold = %d" % g["readstats"]["num_aligned"] in summary
    if case == "syn_default":            # second run: `-blast 0 -sam -SQ` and aligned.log against tests/golden/reports2 (written by the reference)
        r2 = os.path.join(paths.REPO, "tests", "golden", "reports2")
        out2 = tmp_path / "pw"
        os.makedirs(out2)
        subprocess.check_call([exe, "--reads", rd, "--out", str(out2), "--ref", dbs[0], "--gumbel", repr(g["log"]["lambda"][0]), repr(g["log"]["K"][0]),
                               "--blast", "0", "--sam", "-SQ"])
        strip_e = lambda t: [l.split("\tExpect:")[0] + "\t" + l.split("\t")[-1] if l.startswith("Score: ") else l for l in t.split("\n")]
        assert strip_e(open(out2 / "aligned.blast").read()) == strip_e(open(os.path.join(r2, case + ".pairwise.txt")).read())
        sq = lambda p: [l for l in open(p).read().splitlines() if l.startswith("@SQ") or l.startswith("@HD")]
        assert sq(out2 / "aligned.sam") == sq(os.path.join(r2, case + ".sam_header.txt"))
        body = lambda t: [l.replace(os.path.dirname(dbs[0]) + "/", "") for l in t.split("\n")[3:-3]]     # without the command line and the time stamp
        assert body(open(out2 / "aligned.log").read()) == body(open(os.path.join(r2, case + ".log.txt")).read())


def _check_paired(exe, tmp_path, full=True):
    """two mate files, -paired_in -out2: the files of tests/golden/paired (written by the reference) and its per-read records"""
    import json
    import struct
    pd = os.path.join(paths.REPO, "tests", "golden", "paired")
    g = json.load(open(os.path.join(pd, "paired.json")))
    db = os.path.join(paths.REPO, "tests", "golden", "real_db.fasta")
    log = g["two_files"]["log"]
    for variant in ("paired_in_out2", "sout")[:2 if full else 1]:
        out = tmp_path / variant
        os.makedirs(out)
        subprocess.check_call([exe, "--ref", db, "--gumbel", repr(log["lambda"][0]), repr(log["K"][0]), "--reads", os.path.join(pd, "paired_1.fastq"),
                               "--reads", os.path.join(pd, "paired_2.fastq"), "--out", str(out), "--fastx", "--other"] + g[variant]["options"])
        got = {fn: [l.split()[0][1:] for l in open(out / fn).readlines()[0::4]] for fn in sorted(os.listdir(out)) if fn.endswith(".fq")}
        assert got == g[variant]["files"], variant
    # one interleaved file
    inter = str(tmp_path / "interleaved.fastq")
    a, b2 = open(os.path.join(pd, "paired_1.fastq")).readlines(), open(os.path.join(pd, "paired_2.fastq")).readlines()
    with open(inter, "w") as f:
        for i in range(len(a) // 4):
            f.writelines(a[4 * i:4 * i + 4])
            f.writelines(b2[4 * i:4 * i + 4])
    for variant in ("interleaved_paired_in", "interleaved_paired_out_out2")[:2 if full else 1]:
        o2 = tmp_path / variant
        os.makedirs(o2)
        subprocess.check_call([exe, "--ref", db, "--gumbel", repr(log["lambda"][0]), repr(log["K"][0]), "--reads", inter, "--out", str(o2), "--fastx", "--other"] + g[variant]["options"])
        got = {fn: [l.split()[0][1:] for l in open(o2 / fn).readlines()[0::4]] for fn in sorted(os.listdir(o2)) if fn.endswith(".fq")}
        assert got == g[variant]["files"], variant
    kv = refrun.parse_kvdb_dump(str(out / "records.bin"))
    b = open(os.path.join(pd, "paired.records.bin"), "rb").read()
    (n,) = struct.unpack_from("<I", b, 0)
    o = 4
    for k in range(n):
        (l,) = struct.unpack_from("<I", b, o)
        assert kv.get(b"%d_%d" % (k & 1, k >> 1), b"") == b[o + 4:o + 4 + l], k
        o += 4 + l
    assert "Total reads = %d" % n in open(out / "aligned.log").read()


def test_driver_paired_reads_on_the_kernel_emulator(tmp_path):
    _check_paired(_emu_driver(), tmp_path, full=os.environ.get("SMR_EMU_FULL", "0") == "1")


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["syn_default", "syn_all", "two_db_default"])
def test_driver_reproduces_reference_records(case, tmp_path):
    _check_case(build_driver(), case, tmp_path)


@pytest.mark.parametrize("case", ["syn_default", "two_db_default"])
def test_driver_on_the_kernel_emulator(case, tmp_path):
    """the same driver source linked with tests/emu's host build of the kernels (development aid, see tests/test_emu_kernels.py)"""
    _check_case(_emu_driver(), case, tmp_path)


def _emu_driver():
    from helpers import emu
    lib = emu.build()
    exe = os.path.join(os.path.dirname(lib), "smr_align_emu")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", os.path.join(paths.REPO, "examples", "smr_align.cpp"),
                           "-I", os.path.join(paths.REPO, "include"), lib, "-Wl,-rpath," + os.path.dirname(lib), "-o", exe])
    return exe


# ---- the multi-GPU C++ host (examples/smr_align_mgpu.cpp): one thread + one smr_ctx per rank, RCCL all-reduces for the read totals and the counters ----
MGPU = os.path.join(paths.REPO, "examples", "build", "smr_align_mgpu")


def build_mgpu():
    lib = os.path.join(paths.REPO, "sortmerna_amd", "lib")
    import sortmerna_amd.capi as capi
    capi.load()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "-std=c++17", "-O2", os.path.join(paths.REPO, "examples", "smr_align_mgpu.cpp"), "-I", os.path.join(paths.REPO, "include"),
                           "-L", lib, "-lsmr_hip", "-lrccl", "-Wl,-rpath," + lib, "-o", MGPU])
    return MGPU


def _check_mgpu(case, tmp_path, extra):
    g = golden.load()[case]
    dbs, rd, seqs = golden.inputs(case)
    if not isinstance(dbs, list):
        dbs = [dbs]
    cmd = [build_mgpu(), "--reads", rd, "--out", str(tmp_path)] + extra
    for k, db in enumerate(dbs):
        cmd += ["--ref", db, "--gumbel", repr(g["log"]["lambda"][k]), repr(g["log"]["K"][k])]
    if "num_alignments" in g["params"]:
        cmd += ["-num_alignments", str(g["params"]["num_alignments"])]
    if case == "syn_default":
        cmd += ["--fastx", "--other", "--sam", "--blast", "1 qstrand cigar"]
    out = subprocess.check_output(cmd).decode()
    assert "[timing]" in out
    if case == "syn_default":            # the shards' report files, merged in rank order, are the reference's own files for the same options
        sam = [l.rstrip("\n") for l in open(tmp_path / "aligned.sam") if not l.startswith("@")]
        assert sam == [l for l in g["sam"] if not l.startswith("@")]
        assert len([l for l in open(tmp_path / "aligned.sam") if l.startswith("@HD")]) == 1
        blast = [l.rstrip("\n").split("\t") for l in open(tmp_path / "aligned.blast")]
        exp_b = [l.split("\t") for l in g["blast"]]
        assert [a[:10] + a[11:] for a in blast] == [b[:10] + b[11:] for b in exp_b]
        assert [l.split()[0][1:] for l in open(tmp_path / "aligned.fa") if l.startswith(">")] == g["aligned_ids"]
        assert [l.split()[0][1:] for l in open(tmp_path / "other.fa") if l.startswith(">")] == g["other_ids"]
    kv = refrun.parse_kvdb_dump(str(tmp_path / "records.bin"))
    got = [kv.get(b"0_%d" % i, b"") for i in range(len(seqs))]
    exp = golden.records(case)
    bad = [i for i in range(len(seqs)) if got[i] != exp[i]]
    assert not bad, "%d records differ, first %d" % (len(bad), bad[0])
    summary = open(tmp_path / "summary.txt").read()
    assert "Total reads passing E-value threshold = %d" % g["readstats"]["num_aligned"] in summary
    for k, db in enumerate(dbs):
        assert "%s\t%d" % (db, g["readstats"]["reads_matched_per_db"][k]) in summary
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["syn_default", "two_db_default"])
def test_mgpu_host_one_rank_rccl(case, tmp_path):
    """world size 1 through the real RCCL path: ncclCommInitAll, the all-reduce of the read totals, the in-place all-reduce on the device counter block"""
    out = _check_mgpu(case, tmp_path, ["--gpus", "1", "--reduce", "rccl"])
    assert "RCCL reduction" in out


@pytest.mark.gpu
@pytest.mark.parametrize("case,ranks,chunk", [("syn_default", 2, 0), ("syn_default", 2, 50), ("syn_all", 3, 40), ("syn_all", 1, 16), ("two_db_default", 2, 64)])
def test_mgpu_host_shards_and_chunks_on_one_device(case, ranks, chunk, tmp_path):
    """the N-rank path on one GPU (every rank thread gets device 0; the two reductions go through the host because RCCL needs a device per rank):
    record-range shards, any number of chunks streamed through three recycled batch slots (upload | align | records + report rows overlapped),
    counters summed over chunks on the device and over ranks, records and report files concatenated in rank order"""
    _check_mgpu(case, tmp_path, ["--gpus", str(ranks), "--devices", ",".join(["0"] * ranks), "--reduce", "host", "--chunk-reads", str(chunk)])


def _device_count():
    import sortmerna_amd.capi as capi
    return int(capi.load().smr_device_count())


@pytest.mark.multigpu
@pytest.mark.gpu
@pytest.mark.parametrize("case,chunk", [("syn_default", 0), ("syn_all", 40), ("two_db_default", 64)])
def test_mgpu_host_all_devices_rccl(case, chunk, tmp_path):
    """On a node with more than one GPU: one rank per device, the read totals and the counter block reduced by ncclAllReduce with world > 1 (the
    one-device box of the builder only ever runs world 1: this test is for the driver's multi-GPU node).  Records, counters and the merged
    report files must still be the reference's own."""
    n = _device_count()
    if n < 2:
        pytest.skip("one GPU: RCCL with more than one rank needs a device per rank (%d visible)" % n)
    out = _check_mgpu(case, tmp_path, ["--gpus", str(n), "--reduce", "rccl", "--chunk-reads", str(chunk)])
    assert "RCCL reduction" in out and "ranks %d" % n in out
