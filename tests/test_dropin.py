"""The drop-in boundary, compiled and run: oracle/_ref/sortmerna_gpu is the REFERENCE -- its CLI, option parsing, indexer,
Readfeed, Refstats with the ALP Gumbel parameters, KVDB, summary and report writers, compiled from its sources where they lie -- with
align() (processor.cpp:173-285) replaced by the INTEGRATION.md binding over include/smr_hip.h (oracle/dropin/align_gpu.cpp), built by
`make -C oracle dropin`.  It must write the same files as the unmodified binary (oracle/_ref/sortmerna_ref) for the same command line:
aligned/other FASTX, BLAST, SAM, the KVDB values, and aligned.log except the lines that hold the command line, pid and time.
On the GPU box both prebuilt binaries are run (`-m gpu`); without a GPU the same objects are linked against the kernel emulator's build
of the library (needs /root/reference for the objects)."""
import os
import subprocess

import pytest

from helpers import paths, refrun

REF_GPU = os.path.join(paths.ORACLE_DIR, "_ref", "sortmerna_gpu")
CASES = {
    "t0": dict(refs=["t0_ref.fasta"], reads=["t0_read.fasta"], extra=["-fastx", "-other", "-blast", "1 cigar qcov qstrand", "-sam", "-SQ"]),
    "t9": dict(refs=["t9_ref.fasta"], reads=["t9_reads.fasta"], extra=["-fastx", "-blast", "1", "-sam", "-num_alignments", "3"]),
    "real_two_db": dict(refs=["syn_db.fasta", "real_db.fasta"], reads=["two_db_reads.fasta"], extra=["-fastx", "-other", "-blast", "1 cigar", "-sam"]),
    "paired": dict(refs=["real_db.fasta"], reads=["paired/paired_1.fastq", "paired/paired_2.fastq"], extra=["-fastx", "-other", "-paired_in", "-out2", "-blast", "1", "-sam"]),
}


def run_binary(exe, case, wd, env_extra=None):
    c = CASES[case]
    cmd = [exe]
    for r in c["refs"]:
        cmd += ["-ref", os.path.join(paths.GOLDEN, r)]
    for r in c["reads"]:
        cmd += ["-reads", os.path.join(paths.GOLDEN, r)]
    cmd += ["-workdir", str(wd), "-threads", "1"] + c["extra"]
    env = dict(os.environ)
    env["SMR_KVDB_DUMP"] = os.path.join(str(wd), "kvdb_dump.bin")
    env.update(env_extra or {})
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1800)
    assert p.returncode == 0, p.stdout.decode("latin-1")[-3000:]
    out = {}
    od = os.path.join(str(wd), "out")
    for fn in sorted(os.listdir(od)):
        b = open(os.path.join(od, fn), "rb").read()
        if fn.endswith(".log"):           # without the command line, the process id and the time stamps
            b = b"\n".join(l for l in b.split(b"\n") if not (b"Command:" in l or b"Process pid" in l or b"sortmerna" in l.lower() or b"20" in l[:24] and b":" in l[:24])
                           and str(wd).encode() not in l)
        if fn.endswith(".sam"):
            b = b"\n".join(l for l in b.split(b"\n") if not l.startswith(b"@PG"))
        out[fn] = b
    return out, refrun.parse_kvdb_dump(env["SMR_KVDB_DUMP"]), p.stdout.decode("latin-1")


def compare(exe_gpu, case, tmp_path, env_extra=None, min_chunks=1):
    ref_files, ref_kv, _ = run_binary(paths.REF_BIN, case, tmp_path / "ref")
    gpu_files, gpu_kv, log = run_binary(exe_gpu, case, tmp_path / "gpu", env_extra)
    assert "Starting alignment (libsmr_hip)" in log
    import re
    m = re.search(r"reads in (\d+) chunk\(s\)", log)
    assert m and int(m.group(1)) >= min_chunks, log[-600:]
    assert sorted(ref_files) == sorted(gpu_files)
    assert any(fn.startswith("aligned") and fn.endswith(".blast") for fn in ref_files)
    for fn in ref_files:
        assert gpu_files[fn] == ref_files[fn], "%s: %s differs from the reference's (%d vs %d bytes)" % (case, fn, len(gpu_files[fn]), len(ref_files[fn]))
    rk = {k: v for k, v in ref_kv.items() if k[:1].isdigit()}
    gk = {k: v for k, v in gpu_kv.items() if k[:1].isdigit()}
    assert rk == gk and len(rk) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_reference_cli_with_the_gpu_in_the_middle(case, tmp_path):
    if not (os.path.isfile(REF_GPU) and os.path.isfile(paths.REF_BIN)):
        pytest.skip("oracle/_ref/sortmerna_gpu / sortmerna_ref are not in this snapshot (built by `make -C oracle ref dropin` where /root/reference exists)")
    compare(REF_GPU, case, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("case,chunk", [("real_two_db", 100), ("paired", 64)])
def test_reads_stream_through_the_gpu_in_chunks(case, chunk, tmp_path):
    """a reads file several times larger than a chunk: the binding pulls Readfeed::next() chunk by chunk (reader thread | one worker per GPU |
    KVDB writer thread); every output file and every KVDB value is still the unmodified binary's"""
    if not (os.path.isfile(REF_GPU) and os.path.isfile(paths.REF_BIN)):
        pytest.skip("oracle/_ref/sortmerna_gpu / sortmerna_ref are not in this snapshot")
    compare(REF_GPU, case, tmp_path, {"SMR_DROPIN_CHUNK": str(chunk)}, min_chunks=4)


@pytest.mark.multigpu
@pytest.mark.gpu
@pytest.mark.parametrize("case,chunk", [("real_two_db", 50), ("paired", 32)])
def test_dropin_spreads_chunks_over_all_devices(case, chunk, tmp_path):
    """On a node with more than one GPU the binding starts one worker per visible device (smr_device_count) and hands the chunks round; the
    outputs must not depend on which device aligned which chunk.  Skipped on the builder's one-GPU box."""
    import sortmerna_amd.capi as capi
    n = int(capi.load().smr_device_count())
    if n < 2:
        pytest.skip("one GPU visible")
    if not (os.path.isfile(REF_GPU) and os.path.isfile(paths.REF_BIN)):
        pytest.skip("oracle/_ref/sortmerna_gpu / sortmerna_ref are not in this snapshot")
    compare(REF_GPU, case, tmp_path, {"SMR_DROPIN_CHUNK": str(chunk)}, min_chunks=4)


@pytest.mark.skipif(not (paths.have_reference() and paths.have_ref_bin()), reason="needs /root/reference and `make -C oracle ref`")
@pytest.mark.parametrize("case,chunk", [("t0", 0), ("paired", 0), ("paired", 64), ("real_two_db", -1)])
def test_reference_cli_with_the_kernel_emulator_in_the_middle(case, chunk, tmp_path):
    """(chunk 64: the 200 mate pairs stream through the emulator in 7 chunks; chunk -1: more index parts than the binding keeps resident --
    SMR_DROPIN_MAX_RESIDENT=1 for the two DBs of this case -- so every part is uploaded, used and unloaded per chunk of 300 reads)"""
    from helpers import emu
    lib = emu.build()
    subprocess.check_call(["make", "-s", "-j8", "-C", paths.ORACLE_DIR, "dropin", "SMRLIB=" + os.path.dirname(lib), "SMRNAME=smr_emu", "DROPIN_BIN=sortmerna_gpu_emu"])
    if chunk < 0:
        compare(os.path.join(paths.ORACLE_DIR, "_ref", "sortmerna_gpu_emu"), case, tmp_path, {"SMR_DROPIN_CHUNK": "300", "SMR_DROPIN_MAX_RESIDENT": "1"}, min_chunks=2)
        return
    compare(os.path.join(paths.ORACLE_DIR, "_ref", "sortmerna_gpu_emu"), case, tmp_path, {"SMR_DROPIN_CHUNK": str(chunk)} if chunk else None, min_chunks=4 if chunk else 1)
