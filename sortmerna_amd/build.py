"""Build libsmr_hip.so (HIP kernels + C ABI + host side) in-tree for gfx950 with hipcc."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsmr_hip.so")
SOURCES = ["smr_engine.hip", "smr_index.cpp", "smr_reads.cpp", "smr_report.cpp"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join("..", "..", "include", "smr_hip.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libsmr_hip cannot be built (there is no CPU fallback)")


def is_stale():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> sortmerna_amd/lib/libsmr_hip.so ; returns the path."""
    if not force and not is_stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", CSRC, "-o", LIB]
    cmd += os.environ.get("SMR_EXTRA_HIPCC_FLAGS", "").split()          # e.g. -DSMR_CHAIN_PHASES (debug instrumentation)
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-lpthread", "-lz"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
