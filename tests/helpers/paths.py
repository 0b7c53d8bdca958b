"""Locations shared by the tests.  /root/reference exists only in the build container; the GPU box
gets the prebuilt checker binaries under oracle/_ref/ (git-ignored, shipped by gpurun)."""
import os

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REFERENCE = os.environ.get("SMR_REFERENCE", "/root/reference")
REF_DATA = os.path.join(REFERENCE, "data")
ORACLE_DIR = os.path.join(REPO, "oracle")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "sortmerna_ref")
ORACLE_SO = os.path.join(ORACLE_DIR, "_ref", "libsmr_oracle.so")
GOLDEN = os.path.join(REPO, "tests", "golden")


def have_reference():
    return os.path.isdir(REF_DATA)


def have_ref_bin():
    return os.path.isfile(REF_BIN) and os.access(REF_BIN, os.X_OK)
