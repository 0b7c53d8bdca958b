"""The kernel SOURCE (sortmerna_amd/csrc/*.hpp + smr_engine.hip), compiled for the host against the wave64 emulator of
tests/emu, must produce the oracle's / the reference's records too.  Same test bodies as the `-m gpu` parity tests
(test_gpu_parity.py, test_gpu_golden.py), a different `engine` fixture.  This is a development aid that runs without a GPU --
it catches logic errors and wave-divergent shuffles before GPU time is spent -- and not a substitute for the GPU tests:
timing, occupancy, LDS limits and memory-ordering effects of the real machine are not modelled.

By default a subset runs (work-queue seed kernel everywhere, the DFS kernel on the golden cases; ~2 min);
SMR_EMU_FULL=1 runs every GPU test body with both seed kernels (~10 min on 8 cores)."""
import os

import pytest

import sortmerna_amd as smr
from helpers import emu
from helpers.workload import Workload

from test_gpu_parity import (test_seed_scan_matches_oracle, test_align_records_match_oracle, test_multi_part_index,  # noqa: F401
                             test_longer_reads, test_other_seed_lengths, test_non_default_strides, test_empty_batch)
from test_gpu_golden import test_gpu_records_equal_reference_records as _golden_body

FULL = os.environ.get("SMR_EMU_FULL", "0") == "1"
if FULL:
    from test_gpu_parity import test_long_noisy_reads, test_long_reads_with_large_gaps  # noqa: F401


@pytest.fixture(scope="module", autouse=True)
def emulator():
    with emu.active() as lib:
        yield lib


@pytest.fixture(scope="module", params=[0, 1] if FULL else [0], ids=["bfs", "dfs"] if FULL else ["bfs"])
def engine(request, emulator):
    e = smr.Engine(0)
    e.set_seed_mode(request.param)
    yield e
    e.close()


@pytest.fixture(scope="module", params=[0, 1], ids=["bfs", "dfs"])
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
    assert e.sw_mode() == 1                         # smr_create's own check passed
    for max_len, cases in ((100, 24), (250, 24), (700, 16), (1500, 8)):
        assert e.sw_selfcheck(cases, 11 + max_len, max_len) == 0
    e.close()


def test_both_smith_waterman_kernels_give_the_same_records(emulator, wl):
    e = smr.Engine(0)
    recs = {}
    for mode in (0, 1):
        assert e.sw_mode(mode) == mode
        recs[mode], _ = wl.gpu_records(e)
    assert recs[0] == recs[1]
    e.close()
