"""GPU (`-m gpu`): the two device paths added after round 1's GPU budget was spent -- the packed 16-bit Smith-Waterman kernel
(smr_sw_pk.hpp) and the device index build (smr_ibuild.hpp).  Both were developed against the kernel emulator (tests/test_emu_kernels.py
runs these same test bodies); this file sorts after the other GPU tests on purpose, so that they run first."""
import pytest

import sortmerna_amd as smr
from helpers.workload import Workload

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    e = smr.Engine(0)      # raises without a GPU / without the HIP library: no CPU fallback
    yield e
    e.close()


@pytest.fixture(scope="module")
def wl(tmp_path_factory):
    return Workload(str(tmp_path_factory.mktemp("wl")))


def test_sw_kernels_equal_the_reference_ssw_c(engine):
    """smr_ssw_batch (32-bit and packed kernel) against the answers of the reference's own ssw.c (tests/golden/ssw_pairs.json)"""
    from helpers import sswgold
    assert sswgold.check(engine) == 320
    assert sswgold.check_x4(engine) > 150          # the four-problems-per-wave kernel that k_chain batches candidate windows with
    assert sswgold.check_striped(engine) == 600    # the slow path that reproduces ssw.c's stripe geometry, under the schemes that need it


def test_traceback_kernels_equal_the_reference_banded_sw(engine):
    """smr_cigar_batch (k_trace_band<8>, <16>, k_trace_wide) against the CIGARs of the reference's own banded_sw (tests/golden/trace_pairs.json.gz):
    gapless / single-indel short reads, multi-indel reads, windows narrower than their band, 0.7-3 kb noisy reads with long gaps, four scoring schemes"""
    from helpers import tracegold
    assert tracegold.check(engine) > 1800
    assert tracegold.check_variants(engine) > 150


def test_packed_smith_waterman_selfcheck_on_the_device(engine):
    """the packed 16-bit SW kernel against the 32-bit kernel, both on the GPU (smr_sw_selfcheck), and smr_create's own check passed"""
    assert engine.sw_selfcheck(2000, 3, 300) == 0
    assert engine.sw_selfcheck(1000, 4, 1200) == 0
    assert engine.sw_selfcheck(200, 5, 3500) == 0
    assert engine.sw_mode() == 2                    # the default: packed kernel, lane hand-over by wave_ror
    try:
        assert engine.sw_mode(1) == 1               # the readlane variant
        assert engine.sw_selfcheck(2000, 6, 300) == 0 and engine.sw_selfcheck(300, 7, 2000) == 0
    finally:
        engine.sw_mode(2)


def test_both_smith_waterman_kernels_give_the_same_records(engine, wl):
    recs = {}
    try:
        for mode in (0, 1, 2):
            assert engine.sw_mode(mode) == mode
            recs[mode], _ = wl.gpu_records(engine)
    finally:
        engine.sw_mode(2)
    assert recs[0] == recs[1] == recs[2]


def test_device_index_build_equals_the_host_build(engine, wl, tmp_path):
    """smr_index_build_gpu writes the same index files, byte for byte, as the host builder (one part, four parts
    with max_pos truncation, seed length 14), and the aligner gives the same records with it"""
    import hashlib
    import os
    from helpers import paths

    def digest(parts, db, tmp):
        os.makedirs(tmp, exist_ok=True)
        smr.Index.write_files(parts, db, os.path.join(tmp, "i"))
        h = hashlib.md5()
        for f in sorted(os.listdir(tmp)):
            h.update(f.encode())
            h.update(open(os.path.join(tmp, f), "rb").read())
        return h.hexdigest()

    cases = [(wl.db, 3072.0, 18, 10000), (wl.db, 1.0, 18, 5), (os.path.join(paths.REPO, "tests", "golden", "real_db.fasta"), 3072.0, 14, 10000)]
    for k, (db, mb, L, mp) in enumerate(cases):
        host = smr.Index.build(db, L, mb, mp, 0)
        dev = smr.Index.build_gpu(engine, db, L, mb, mp)
        assert len(host) == len(dev)
        assert digest(dev, db, str(tmp_path / ("d%d" % k))) == digest(host, db, str(tmp_path / ("h%d" % k)))
    dev = smr.Index.build_gpu(engine, wl.db, 18, 3072.0, 10000)
    p = smr.default_params(minimal_score=wl.minimal_score)
    smr.align(engine, wl.reads, [dev], [p], with_cigar=True)
    recs_dev = engine.records()
    recs_host, _ = wl.gpu_records(engine)
    assert recs_dev == recs_host


def test_cpp_driver_with_two_mate_files(tmp_path):
    """examples/smr_align.cpp with two mate files (two resident batches) and -paired_in -out2 / -sout: the reference's output files and records"""
    import test_cpp_driver as drv
    drv._check_paired(drv.build_driver(), tmp_path)


@pytest.mark.parametrize("lnwin,db_nt", [(18, 2_000_000), (14, 300_000)])
def test_pigeonhole_layout_built_on_the_device_equals_the_host_transform(engine, tmp_path, lnwin, db_nt):
    """the layout k_seed_pg reads is built by smr_index_upload on the device (smr_pgbuild.hpp); the host transform is its checker: word for word"""
    from sortmerna_amd import synth
    db = str(tmp_path / "db.fasta")
    synth.make_db(db, db_nt, seed=11, family_size=25)
    parts = smr.Index.build_gpu(engine, db, lnwin, 3072.0, 10000)
    for s, ix in enumerate(parts):
        engine.upload_index(ix, 8 + s)
        engine.check_device_index(ix, 8 + s)
        engine.unload_index(8 + s)
