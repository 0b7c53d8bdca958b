"""A fixed slice of tools/fuzz_emu.py in the CPU suite: a dozen seeded (workload, options, library switches) cases, kernel sources on the emulator
against the oracle.  The tool itself runs thousands of cases; this keeps it alive and its generator covered."""
import os
import subprocess
import sys

from helpers import paths


def test_a_slice_of_the_differential_fuzzer():
    out = subprocess.run([sys.executable, os.path.join(paths.REPO, "tools", "fuzz_emu.py"), "9000", "12"], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "12 case(s), 0 differing or failing" in out.stdout
    assert out.stdout.count(" ok:") >= 9            # (a few of any dozen draw an input that is refused explicitly)


import pytest  # noqa: E402


@pytest.mark.skipif(not (paths.have_reference() and paths.have_ref_bin()), reason="needs /root/reference + oracle/_ref/sortmerna_ref")
def test_a_slice_of_the_oracle_against_the_reference_binary():
    """tools/fuzz_ref.py: the same kind of random (workload, option) cases through the UNMODIFIED reference binary and through the oracle -- records
    and counters equal.  (The fuzzer above referees the kernels with the oracle; this pins the referee to the reference over the random option space.)"""
    out = subprocess.run([sys.executable, os.path.join(paths.REPO, "tools", "fuzz_ref.py"), "9100", "9"], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert ", 0 differing or failing" in out.stdout
    assert out.stdout.count(" ok:") >= 7
