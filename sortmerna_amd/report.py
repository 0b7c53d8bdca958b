"""ctypes wrapper of the report writers of libsmr_hip (smr_report_* in include/smr_hip.h): aligned/other FASTX, BLAST tabular, SAM
from the per-read records -- the reference's writeReports() pass (output.cpp:169-272)."""
import ctypes as C

from . import capi
from .engine import SmrError


def corrected_sizes(K, info, all_reads_count, all_reads_len):
    """Refstats::full_ref / full_read after the length correction (refstats.cpp:238-257)"""
    a, b = C.c_uint64(), C.c_uint64()
    capi.load().smr_refstats_corrected(K, info.bg, info.full_len, info.numseq, all_reads_count, all_reads_len, C.byref(a), C.byref(b))
    return a.value, b.value


class Report:
    def __init__(self, out_dir, is_fastq, fastx=True, other=True, blast_cols=None, sam=False):
        """blast_cols: None = no BLAST report, else a list out of "cigar", "qcov", "qstrand" (output order)"""
        self.L = capi.load()
        o = capi.ReportOpts()
        o.fastx, o.other, o.sam = int(fastx), int(other), int(sam)
        o.blast_tabular = int(blast_cols is not None)
        o.blast_cols = " ".join(blast_cols or []).encode()
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = self.L.smr_report_open(out_dir.encode(), C.byref(o), int(is_fastq), C.byref(h), err, 512)
        if rc != 0:
            raise SmrError("smr_report_open: %s (rc=%d)" % (err.value.decode(), rc))
        self.h = h

    def _chk(self, rc, what):
        if rc != 0:
            raise SmrError("%s: %s (rc=%d)" % (what, self.L.smr_report_last_error(self.h).decode(), rc))

    def set_db(self, index_num, lam, K, full_ref_corr, full_read_corr):
        self._chk(self.L.smr_report_set_db(self.h, index_num, lam, K, full_ref_corr, full_read_corr), "smr_report_set_db")

    def set_part(self, index_num, part, index):
        self._chk(self.L.smr_report_set_part(self.h, index_num, part, index.h), "smr_report_set_part")

    def add(self, header, seq, qual, record):
        self._chk(self.L.smr_report_add(self.h, header.encode(), seq.encode(), qual.encode() if qual else None, record, len(record)),
                  "smr_report_add")

    def close(self):
        if self.h:
            rc = self.L.smr_report_close(self.h)
            self.h = None
            if rc != 0:
                raise SmrError("smr_report_close rc=%d" % rc)
