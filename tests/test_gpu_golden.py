"""GPU (-m gpu): libsmr_hip, called through its C ABI, against the per-read records the UNMODIFIED reference produced
(tests/golden/*.records.bin): the reference's own t0/t2 and t9 inputs, the synthetic workload under 9 option variants,
a slice of the bundled silva-arc-16s DB with set4 reads (IUPAC letters in the references), and a two-DB run.
Bar: byte-exact Read::toBinString records (classification, scores, coordinates, CIGARs) and identical Readstats counters."""
import pytest

import sortmerna_amd as smr
from helpers import golden, refrun
from helpers.cases import CASES, gpu_run

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[0, 1], ids=["pg", "dfs"])
def engine(request):
    e = smr.Engine(0)      # raises without a GPU / without the HIP library: no CPU fallback
    e.set_seed_mode(request.param)
    yield e
    e.close()


@pytest.mark.parametrize("case", CASES)
def test_gpu_records_equal_reference_records(engine, case, tmp_path):
    g = golden.load()[case]
    o = gpu_run(engine, case, tmp_path)
    exp = golden.records(case)
    bad = [i for i, (a, b) in enumerate(zip(o["records"], exp)) if a != b]
    assert not bad, "%s: %d records differ, first %d\n gpu=%s\n ref=%s" % (
        case, len(bad), bad[0], refrun.parse_record(o["records"][bad[0]]), refrun.parse_record(exp[bad[0]]))
    assert o["num_aligned"] == g["readstats"]["num_aligned"] == g["log"]["num_aligned"]
    assert o["per_db"] == g["readstats"]["reads_matched_per_db"]
    assert o["num_short"] == g["readstats"]["num_short"]
