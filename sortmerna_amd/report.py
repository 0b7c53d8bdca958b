"""ctypes wrapper of the report writers of libsmr_hip (smr_report_* in include/smr_hip.h): aligned/other FASTX, BLAST tabular, SAM
from the per-read records -- the reference's writeReports() pass (output.cpp:169-272)."""
import ctypes as C

from . import capi
from .engine import SmrError


def corrected_sizes(K, info, all_reads_count, all_reads_len, full_read_scale=1):
    """Refstats::full_ref / full_read after the length correction (refstats.cpp:238-257); full_read_scale: the number of processing threads of a
    reference run with -score_split (refstats.cpp:247), which the e-value of its BLAST report then carries too"""
    a, b = C.c_uint64(), C.c_uint64()
    capi.load().smr_refstats_corrected_split(K, info.bg, info.full_len, info.numseq, all_reads_count, all_reads_len, int(full_read_scale), C.byref(a), C.byref(b))
    return a.value, b.value


class Report:
    def __init__(self, out_dir, is_fastq, fastx=True, other=True, blast_cols=None, sam=False, blast_pairwise=False, sam_sq=False, cmdline=None,
                 paired_in=False, paired_out=False, out2=False, sout=False, zip_out=False):
        """blast_cols: None = no tabular BLAST report, else a list out of "cigar", "qcov", "qstrand" (output order);
        blast_pairwise: the `-blast 0` text instead; sam_sq: @SQ header lines (-SQ); cmdline: text of the SAM @PG CL: field"""
        self.L = capi.load()
        o = capi.ReportOpts()
        o.fastx, o.other, o.sam = int(fastx), int(other), int(sam)
        o.blast_tabular = int(blast_cols is not None)
        o.blast_cols = " ".join(blast_cols or []).encode()
        o.blast_pairwise, o.sam_sq = int(blast_pairwise), int(sam_sq)
        o.paired_in, o.paired_out, o.out2, o.sout = int(paired_in), int(paired_out), int(out2), int(sout)
        o.zip_out = int(zip_out)
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = self.L.smr_report_open(out_dir.encode(), C.byref(o), int(is_fastq), C.byref(h), err, 512)
        if rc != 0:
            raise SmrError("smr_report_open: %s (rc=%d)" % (err.value.decode(), rc))
        self.h = h
        if cmdline is not None:
            self._chk(self.L.smr_report_set_cmdline(self.h, cmdline.encode()), "smr_report_set_cmdline")

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

    def add_pair(self, mate1, mate2):
        """mate = (header, seq, qual, record): routed like the reference's paired FASTX reports (-paired_in / -paired_out / -out2 / -sout)"""
        (h1, s1, q1, r1), (h2, s2, q2, r2) = mate1, mate2
        self._chk(self.L.smr_report_add_pair(self.h, h1.encode(), s1.encode(), q1.encode() if q1 else None, r1, len(r1),
                                             h2.encode(), s2.encode(), q2.encode() if q2 else None, r2, len(r2)), "smr_report_add_pair")

    def close(self):
        if self.h:
            rc = self.L.smr_report_close(self.h)
            self.h = None
            if rc != 0:
                raise SmrError("smr_report_close rc=%d" % rc)


def write_summary(path, dbs, reads_files, total_reads, num_aligned, all_reads_len, min_read_len, max_read_len, seed_len=18, num_seeds=2, edges=4,
                  match=2, mismatch=-3, gap_open=5, gap_ext=2, score_N=-3, sam_sq=False, threads=1, cmdline="", pid="", timestamp=""):
    """aligned.log (summary.cpp:102-175).  dbs: list of dicts {ref_file, skiplengths, lam, K, minimal_score, reads_matched}"""
    L = capi.load()
    arr = (capi.SummaryDb * len(dbs))()
    for i, d in enumerate(dbs):
        arr[i].ref_file = d["ref_file"].encode()
        for k in range(3):
            arr[i].skiplengths[k] = d["skiplengths"][k]
        arr[i].lam, arr[i].K, arr[i].minimal_score, arr[i].reads_matched = d["lam"], d["K"], d["minimal_score"], d["reads_matched"]
    rf = (C.c_char_p * len(reads_files))(*[r.encode() for r in reads_files])
    s = capi.Summary()
    s.cmdline, s.pid, s.timestamp = cmdline.encode(), pid.encode(), timestamp.encode()
    s.seed_len, s.num_seeds, s.edges, s.match, s.mismatch, s.gap_open, s.gap_ext, s.score_N = seed_len, num_seeds, edges, match, mismatch, gap_open, gap_ext, score_N
    s.sam_sq, s.threads = int(sam_sq), threads
    s.reads_files, s.n_reads_files = rf, len(reads_files)
    s.total_reads, s.num_aligned, s.all_reads_len, s.min_read_len, s.max_read_len = total_reads, num_aligned, all_reads_len, min_read_len, max_read_len
    s.dbs, s.n_dbs = arr, len(dbs)
    rc = L.smr_summary_write(path.encode(), C.byref(s))
    if rc != 0:
        raise SmrError("smr_summary_write rc=%d" % rc)
