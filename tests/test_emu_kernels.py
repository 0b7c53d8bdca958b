"""The kernel SOURCE (sortmerna_amd/csrc/*.hpp + smr_engine.hip), compiled for the host against the wave64 emulator of
tests/emu, must produce the oracle's / the reference's records too.  Same test bodies as the `-m gpu` parity tests
(test_gpu_parity.py, test_gpu_golden.py), a different `engine` fixture.  This is a development aid that runs without a GPU --
it catches logic errors and wave-divergent shuffles before GPU time is spent -- and not a substitute for the GPU tests:
timing, occupancy, LDS limits and memory-ordering effects of the real machine are not modelled.

By default a subset runs (pigeonhole seed kernel everywhere, the DFS kernel on the golden cases; ~2 min);
SMR_EMU_FULL=1 runs every GPU test body with both seed kernels (~13 min on 8 cores)."""
import os

import pytest

import sortmerna_amd as smr
from helpers import emu
from helpers.workload import Workload

from test_gpu_parity import (test_seed_scan_matches_oracle, test_multi_part_index, test_longer_reads, test_empty_batch,  # noqa: F401
                             test_seed_work_counters_match_oracle, test_batch_dominated_by_one_sequence, test_pigeonhole_seed_kernel_equals_the_dfs_kernel, test_small_candidate_pool_is_redone_and_grows,
                             test_percent_edges_on_kilobase_reads, test_mixed_read_lengths, test_reads_sharing_seeds_with_thousands_of_references,
                             test_optional_paths_of_the_candidate_stage_give_the_oracle_records, test_pigeonhole_search_bytes_equal_a_host_recount,
                             test_a_window_with_hundreds_of_hits, test_rounds_adapt_from_part_to_part_without_changing_a_record,
                             test_edges_outside_what_the_reference_defines_are_refused)
from test_gpu_parity import test_align_records_match_oracle as _align_body
from test_gpu_golden import test_gpu_records_equal_reference_records as _golden_body

FULL = os.environ.get("SMR_EMU_FULL", "0") == "1"
# round 6's paths: every variant with SMR_EMU_FULL=1 (and in the -m gpu suite), a slice of them by default (the CPU suite has a time budget)
from test_gpu_parity import (test_skewed_batch_sorted_by_several_blocks_and_searched_once_per_seed as _skew_body, SKEW_VARIANTS,      # noqa: E402
                             test_one_seed_sort_for_the_parts_and_references_of_a_batch as _shared_body,
                             test_schemes_under_which_ssw_c_leaves_the_affine_recurrence_give_the_oracle_records as _striped_body)


@pytest.mark.parametrize("env,mode", [(e, m) for e in SKEW_VARIANTS for m in (0, 1)] if FULL else [(SKEW_VARIANTS[0], 0), (SKEW_VARIANTS[2], 0), (SKEW_VARIANTS[1], 1)],
                         ids=lambda v: ",".join("%s=%s" % (k.replace("SMR_SEED_", ""), x) for k, x in v.items()) if isinstance(v, dict) else ("dfs" if v else "pg"))
def test_skewed_batch_sorted_by_several_blocks_and_searched_once_per_seed(tmp_path, monkeypatch, env, mode):
    _skew_body(tmp_path, monkeypatch, env, mode)


@pytest.mark.parametrize("mode", ["1", "0"] if FULL else ["1"], ids=lambda m: "shared" if m == "1" else "per-part")
def test_one_seed_sort_for_the_parts_and_references_of_a_batch(tmp_path, monkeypatch, mode):
    _shared_body(tmp_path, monkeypatch, mode)


@pytest.mark.parametrize("scoring", [{"gap_open": 3, "gap_ext": 3}, {"gap_open": 2, "gap_ext": 2}, {"mismatch": -5, "gap_open": 2, "gap_ext": 1}, {"score_N": 1}, {"gap_open": 3, "gap_ext": 3, "num_alignments": 3}]
                         if FULL else [{"score_N": 1, "gap_open": 3, "gap_ext": 3}],
                         ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
def test_schemes_under_which_ssw_c_leaves_the_affine_recurrence_give_the_oracle_records(engine, wl, scoring):
    _striped_body(engine, wl, scoring)


if FULL:
    from test_gpu_parity import (test_align_records_match_oracle, test_other_seed_lengths, test_non_default_strides,  # noqa: F401
                                 test_long_noisy_reads, test_long_reads_with_large_gaps, test_very_long_reads)
    from test_gpu_sw_and_index_build import test_device_index_build_equals_the_host_build as test_device_index_build_gpu_test_body  # noqa: F401
else:
    @pytest.mark.parametrize("opts", [{}, {"is_reverse": 0}], ids=["default", "F"])
    def test_align_records_match_oracle_subset(engine, wl, opts):
        _align_body(engine, wl, opts)


from test_gpu_parity import test_candidate_walk_in_rounds_gives_the_oracle_records as _walk_body, WALK_VARIANTS  # noqa: E402

# (the GPU suite runs every variant with five option sets; here every variant once and one variant with the other option sets, unless SMR_EMU_FULL=1)
_WALK_CASES = [(e, o) for e in WALK_VARIANTS for o in ([{}, {"num_alignments": 0}, {"is_best": 0, "num_alignments": 2}, {"num_seeds": 1}, {"min_lis": 3, "num_alignments": 2}] if FULL else [{}])]
if not FULL:
    _WALK_CASES += [(WALK_VARIANTS[4], o) for o in ({"num_alignments": 0}, {"is_best": 0, "num_alignments": 2}, {"num_seeds": 1})]


@pytest.mark.parametrize("env,opts", _WALK_CASES, ids=lambda v: ",".join("%s=%s" % (k.replace("SMR_WALK_", ""), x) for k, x in v.items()) or "default")
def test_candidate_walk_in_rounds_gives_the_oracle_records(wl, monkeypatch, env, opts):
    _walk_body(wl, monkeypatch, env, opts)


@pytest.fixture(scope="module", autouse=True)
def emulator():
    with emu.active() as lib:
        yield lib


@pytest.fixture(scope="module", params=[0, 1] if FULL else [0], ids=["pg", "dfs"] if FULL else ["pg"])
def engine(request, emulator):
    e = smr.Engine(0)
    e.set_seed_mode(request.param)
    yield e
    e.close()


@pytest.fixture(scope="module", params=[0, 1], ids=["pg", "dfs"])
def engine_both(request, emulator):
    e = smr.Engine(0)
    e.set_seed_mode(request.param)
    yield e
    e.close()


from helpers.cases import CASES  # noqa: E402


@pytest.mark.parametrize("case", CASES)
def test_emulated_kernels_equal_reference_records(engine_both, case, tmp_path):
    _golden_body(engine_both, case, tmp_path)


@pytest.fixture(scope="module")
def wl(tmp_path_factory, emulator):
    return Workload(str(tmp_path_factory.mktemp("wl")))


def test_packed_smith_waterman_equals_the_32bit_kernel(emulator):
    """smr_sw_selfcheck: the packed 16-bit SW kernel (smr_sw_pk.hpp: 128 virtual lanes, v_perm score lookup) against the 32-bit
    systolic kernel on seeded random pairs -- single strip (<= 128 / 256 rows), several strips, N in read and reference,
    two scoring schemes, forward and reverse pass."""
    e = smr.Engine(0)
    assert e.sw_mode() == 2                         # smr_create's own check passed (default: the wave_ror variant)
    for max_len, cases in ((100, 24), (250, 24), (700, 16), (1500, 8)):
        assert e.sw_selfcheck(cases, 11 + max_len, max_len) == 0
    assert e.sw_mode(1) == 1                        # the readlane variant of the packed kernel
    for max_len, cases in ((100, 24), (250, 24), (700, 16), (1500, 8)):
        assert e.sw_selfcheck(cases, 11 + max_len, max_len) == 0
    e.close()


@pytest.mark.skipif(not FULL, reason="SMR_EMU_FULL=1 (the SW-level tests above and every other emulator test already run the packed kernel)")
def test_both_smith_waterman_kernels_give_the_same_records(emulator, wl):
    e = smr.Engine(0)
    recs = {}
    for mode in (0, 1, 2):
        assert e.sw_mode(mode) == mode
        recs[mode], _ = wl.gpu_records(e)
    assert recs[0] == recs[1] == recs[2]
    e.close()


def _index_files_digest(parts, db, tmp):
    import hashlib
    os.makedirs(tmp, exist_ok=True)
    smr.Index.write_files(parts, db, os.path.join(tmp, "i"))
    h = hashlib.md5()
    for f in sorted(os.listdir(tmp)):
        h.update(f.encode())
        h.update(open(os.path.join(tmp, f), "rb").read())
    i = parts[0].info()
    return h.hexdigest(), len(parts), i.n_nodes, i.n_buckets, i.n_entries, i.n_ids, i.n_pos


@pytest.mark.parametrize("db,max_mb,lnwin,max_pos", [("t9_ref.fasta", 3072.0, 18, 10000), ("syn_db.fasta", 3072.0, 18, 10000),
                                                     ("syn_db.fasta", 0.15, 18, 3), ("real_db.fasta", 3072.0, 14, 10000)])
def test_device_index_build_equals_the_host_build(emulator, tmp_path, db, max_mb, lnwin, max_pos):
    """smr_index_build_gpu (sorting, ids, positions with max_pos truncation, mini-trie layout on the device; smr_ibuild.hpp) writes
    the same index files, byte for byte, as the host builder: one part / several parts, L = 18 / 14, IUPAC letters in the DB"""
    from helpers import paths
    path = os.path.join(paths.REPO, "tests", "golden", db)
    e = smr.Engine(0)
    host = smr.Index.build(path, lnwin, max_mb, max_pos, 0)
    dev = smr.Index.build_gpu(e, path, lnwin, max_mb, max_pos)
    assert _index_files_digest(dev, path, str(tmp_path / "d")) == _index_files_digest(host, path, str(tmp_path / "h"))
    for ix in dev:
        ix.selfcheck()
    e.close()


def test_smoke_entry_point_on_the_emulator(emulator):
    """__graft_entry__.smoke() end to end (it is written for cuda:0; here the binding is routed to the emulator build)"""
    import __graft_entry__ as entry
    entry.smoke()


def test_sw_kernels_equal_the_reference_ssw_c(emulator):
    """smr_ssw_batch (32-bit and packed kernel) against tests/golden/ssw_pairs.json = answers of the reference's own ssw.c (ssw_align, flag 2)
    for 320 seeded pairs: score1, ref_begin1/end1, read_begin1/end1"""
    from helpers import sswgold
    e = smr.Engine(0)
    assert sswgold.check(e) == 320
    assert sswgold.check_x4(e) > 150
    e.close()


def test_striped_slow_path_equals_the_reference_ssw_c(emulator):
    """smr_ssw_batch mode 4 (smr_sw_striped.hpp: ssw.c's stripe geometry on a wave) against tests/golden/ssw_pairs_striped.json = the answers of the
    reference's own ssw.c under six schemes, four of them ones under which the striped kernels leave the affine recurrence (gap_open <= gap_ext,
    2 gap < |mismatch|) and one with a positive score for N"""
    from helpers import sswgold
    e = smr.Engine(0)
    assert sswgold.check_striped(e, max_pairs=None if FULL else 25) == (600 if FULL else 150)
    e.close()


def test_traceback_kernels_equal_the_reference_banded_sw(emulator):
    """smr_cigar_batch (k_trace_band<8>, <16>, k_trace_wide: band doubling, several strips, gap_open < gap_ext) against
    tests/golden/trace_pairs.json.gz = CIGARs of the reference's own banded_sw for seeded pairs"""
    from helpers import tracegold
    e = smr.Engine(0)
    n = tracegold.check(e, kinds=["short", "tiny", "indels"])
    n += tracegold.check(e, kinds=["long"], max_pairs=3)
    assert n > 1700
    assert tracegold.check_variants(e) > 150
    e.close()


def test_chunked_pipeline_equals_one_batch(emulator, wl):
    """the host-side streaming pattern of INTEGRATION.md / examples/smr_align_mgpu.cpp: the reads in chunks (smr_reads_slice), chunk k+1 uploaded
    by a second thread into its own batch (smr_reads_upload_batch, upload stream) while chunk k is aligned; per-chunk records and counters
    add up to those of the whole batch"""
    import threading
    e = smr.Engine(0)
    for s, ix in enumerate(wl.parts):
        e.upload_index(ix, s)
    p = smr.default_params(minimal_score=wl.minimal_score)
    slots = list(range(len(wl.parts)))
    e.select_batch(0)
    e.upload_reads(wl.reads, 1)
    smr.align_resident(e, slots, [p])
    whole = e.records()
    whole_ctr = e.counters(1)
    n = wl.reads.count
    cuts = [0, n // 3, n // 3 + 1, 2 * n // 3, n]                   # four chunks, one of them a single read
    chunks = [wl.reads.slice(cuts[i], cuts[i + 1] - cuts[i]) for i in range(4)]
    assert sum(c.count for c in chunks) == n and chunks[1].count == 1
    e.select_batch(1)
    e.upload_reads(chunks[0], 1)
    got, aligned = [], 0
    for k in range(4):
        t = None
        if k + 1 < 4:
            t = threading.Thread(target=e.upload_reads_batch, args=(2 + k, chunks[k + 1], 1))
            t.start()
        e.select_batch(1 + k)
        smr.align_resident(e, slots, [p])
        e.n_reads = chunks[k].count
        got += e.records()
        aligned += e.counters(1)["num_aligned"]
        if t:
            t.join()
    assert got == whole and aligned == whole_ctr["num_aligned"]
    # a writer thread's view: the records of an earlier chunk by batch number while another batch is the selected one (slots recycled by the
    # streaming hosts: examples/smr_align_mgpu.cpp)
    e.select_batch(4)
    assert [e.record_batch(1, i) for i in range(chunks[0].count)] == whole[:cuts[1]]
    assert [e.record_batch(3, i) for i in range(chunks[2].count)] == whole[cuts[2]:cuts[3]]
    assert e.L.smr_device_count() >= 1
    with pytest.raises(smr.SmrError):
        e.upload_reads_batch(4, chunks[0], 1)                       # the selected batch must go through smr_reads_upload
    for c in chunks:
        c.free()
    e.close()


@pytest.mark.parametrize("lnwin,db_nt", [(18, 300_000), (14, 120_000), (10, 60_000)])
def test_pigeonhole_layout_built_on_the_device_equals_the_host_transform(emulator, tmp_path, lnwin, db_nt):
    """smr_index_upload builds the layout k_seed_pg reads (smr_pgbuild.hpp: DFS collection per mini-trie, two stable radix sorts, directory
    slots by the entries themselves) from the uploaded arena; the host transform smr_build_pigeonhole is its checker: every word equal --
    big blocks with directories, blocks of <= 4 entries, absent mini-tries, seed lengths with other h / pw"""
    from sortmerna_amd import synth
    db = str(tmp_path / "db.fasta")
    synth.make_db(db, db_nt, seed=11, family_size=25)
    e = smr.Engine(0)
    parts = smr.Index.build(db, lnwin, 3072.0, 10000, 0)
    for s, ix in enumerate(parts):
        e.upload_index(ix, s)
        e.check_device_index(ix, s)
    e.close()


def test_a_small_bloom_bitmap_in_k_cand_only_marks_more_reads(emulator, tmp_path, monkeypatch):
    """k_cand ends the pass of a read whose positions set no Bloom bit twice; a smaller bitmap (SMR_CAND_BLOOM words per read, 64 = 2 048 bits)
    collides more often, which hands more reads to k_chain's exact walk and changes no record."""
    from helpers.workload import Workload
    w = Workload(str(tmp_path), db_nt=300_000, n_reads=2000, frac_db=0.1, seed=23, family_size=4)
    exp, ctr = w.oracle_records()
    for words in ("64", "512"):
        monkeypatch.setenv("SMR_CAND_BLOOM", words)
        e = smr.Engine(0)
        got, c = w.gpu_records(e)
        assert got == exp, "SMR_CAND_BLOOM=%s: %d records differ" % (words, sum(1 for a, b in zip(got, exp) if a != b))
        assert c["num_aligned"] == ctr["num_aligned"]
        e.close()


def test_candidate_sets_from_k_cand_records_and_from_the_position_lists_agree(emulator, tmp_path, monkeypatch):
    """k_cand leaves the positions of a marked read as a record for k_chain (SMR_HANDOVER=1, the default); without it k_chain gathers them
    itself through hits -> list bounds -> positions.  Same records."""
    from helpers.workload import Workload
    w = Workload(str(tmp_path), db_nt=300_000, n_reads=2500, frac_db=0.15, seed=29, family_size=8)
    exp, ctr = w.oracle_records()
    for h in ("1", "0"):
        monkeypatch.setenv("SMR_HANDOVER", h)
        e = smr.Engine(0)
        got, c = w.gpu_records(e)
        assert got == exp, "SMR_HANDOVER=%s: %d records differ" % (h, sum(1 for a, b in zip(got, exp) if a != b))
        assert c["num_aligned"] == ctr["num_aligned"]
        p = e.prof()
        assert p.n_sw_fwd == ctr["n_sw_fwd"], (h, p.n_sw_fwd, ctr["n_sw_fwd"])
        e.close()


