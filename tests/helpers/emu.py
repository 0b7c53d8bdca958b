"""TEST INFRASTRUCTURE: builds tests/emu/_build/libsmr_emu.so -- the product's kernel + engine sources compiled by g++ for the
host against tests/emu/shim (a wave64 fiber emulator, see tests/emu/shim/hip/hip_runtime.h) -- and lets a test route the python
binding through it.  It exists so that the kernel source itself can be checked against the oracle without a GPU; it is not a
fallback of the product (nothing in sortmerna_amd/ knows about it) and `-m gpu` tests never use it."""
import contextlib
import ctypes
import fcntl
import os
import subprocess

from sortmerna_amd import capi

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EMU = os.path.join(ROOT, "tests", "emu")
CSRC = os.path.join(ROOT, "sortmerna_amd", "csrc")
LIB = os.path.join(EMU, "_build", "libsmr_emu.so")
_bound = None


def _sources():
    src = [os.path.join(EMU, "emu_runtime.cpp"), os.path.join(EMU, "shim", "hip", "hip_runtime.h"), os.path.join(EMU, "shim", "smr_device_ops.hpp"),
           os.path.join(ROOT, "include", "smr_hip.h")]
    src += [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    return src


def _fresh():
    return os.path.isfile(LIB) and all(os.path.getmtime(s) <= os.path.getmtime(LIB) for s in _sources())


def build(force=False):
    if not force and _fresh():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    # several test processes (pytest -n) may find the library stale at once: one builds, the others wait and find it fresh
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or not _fresh():
            tmp = "%s.%d.tmp" % (LIB, os.getpid())
            cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-I", os.path.join(EMU, "shim"), "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                   os.path.join(EMU, "emu_runtime.cpp"), "-x", "c++", os.path.join(CSRC, "smr_engine.hip"), "-x", "none", "-D__host__=", "-D__device__=",
                   os.path.join(CSRC, "smr_index.cpp"), os.path.join(CSRC, "smr_reads.cpp"), os.path.join(CSRC, "smr_report.cpp"),
                   "-o", tmp, "-lpthread", "-lz", "-ldl"] + os.environ.get("SMR_EMU_EXTRA_FLAGS", "").split()
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)             # new inode: a process that has the old library mapped keeps running
    return LIB


def lib():
    global _bound
    if _bound is None:
        _bound = capi.bind(ctypes.CDLL(build()))
    return _bound


@contextlib.contextmanager
def active():
    """Inside this context sortmerna_amd's binding talks to the emulator build instead of libsmr_hip.so."""
    saved = capi._lib
    capi._lib = lib()
    try:
        yield capi._lib
    finally:
        capi._lib = saved
