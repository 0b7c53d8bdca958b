"""The wave64 host emulator (tests/emu) checked against known answers, and its divergence detector against a kernel that
shuffles inside a divergent branch (the bug class that cost a GPU crash in round 1)."""
import os
import subprocess

from helpers import emu


def _selftest(tmp_path):
    exe = str(tmp_path / "emu_selftest")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-I", os.path.join(emu.EMU, "shim"), os.path.join(emu.EMU, "emu_runtime.cpp"),
                           os.path.join(emu.EMU, "selftest.cpp"), "-o", exe, "-lpthread", "-ldl"])
    return exe


def test_emulator_primitives_and_divergence_detection(tmp_path):
    exe = _selftest(tmp_path)
    r = subprocess.run([exe, "ok"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
    r = subprocess.run([exe, "diverge"], capture_output=True, text=True)
    assert r.returncode != 0 and "DIFFERENT wave-level operations" in r.stderr and "not detected" not in r.stdout, r.stdout + r.stderr
