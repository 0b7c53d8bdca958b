"""ctypes wrapper of the CPU oracle (oracle/_ref/libsmr_oracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import paths


class Params(C.Structure):
    _fields_ = [
        ("lnwin", C.c_uint32), ("skiplengths", C.c_uint32 * 3), ("num_seeds", C.c_int32), ("min_lis", C.c_int32),
        ("edges", C.c_int32), ("is_as_percent", C.c_int32), ("match", C.c_int32), ("mismatch", C.c_int32),
        ("score_N", C.c_int32), ("gap_open", C.c_int32), ("gap_ext", C.c_int32), ("minimal_score", C.c_uint32),
        ("num_alignments", C.c_uint32), ("is_best", C.c_int32), ("is_full_search", C.c_int32),
        ("is_forward", C.c_int32), ("is_reverse", C.c_int32), ("minoccur", C.c_uint32),
        ("index_num", C.c_uint32), ("part", C.c_uint32), ("is_last_index_part", C.c_int32),
    ]


class Counters(C.Structure):
    _fields_ = [
        ("num_aligned", C.c_uint64), ("num_short", C.c_uint64), ("reads_matched_per_db", C.c_uint64 * 64),
        ("n_lookup", C.c_uint64), ("n_node", C.c_uint64), ("n_entry", C.c_uint64), ("n_hit", C.c_uint64),
        ("n_sw_fwd", C.c_uint64), ("n_sw_rev", C.c_uint64), ("n_traceback", C.c_uint64), ("n_windows", C.c_uint64),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("filesize", C.c_uint64), ("bg", C.c_double * 4), ("full_len", C.c_uint64), ("lnwin", C.c_uint32),
        ("numseq", C.c_uint64), ("nparts", C.c_uint16), ("part_start", C.c_uint64 * 256),
        ("part_bytes", C.c_uint64 * 256), ("part_numseq", C.c_uint32 * 256),
    ]


class SswResult(C.Structure):
    _fields_ = [
        ("score1", C.c_uint16), ("ref_begin1", C.c_int32), ("ref_end1", C.c_int32), ("read_begin1", C.c_int32),
        ("read_end1", C.c_int32), ("cigar_len", C.c_uint32), ("cigar", C.c_uint32 * 4096),
    ]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", paths.ORACLE_DIR, "oracle"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(paths.ORACLE_SO) or (
        os.path.getmtime(paths.ORACLE_SO) < os.path.getmtime(os.path.join(paths.ORACLE_DIR, "smr_oracle.c"))
    ):
        build()
    L = C.CDLL(paths.ORACLE_SO)
    L.orc_index_load.restype = C.c_void_p
    L.orc_index_load.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32]
    L.orc_index_free.argtypes = [C.c_void_p]
    L.orc_index_num_ids.restype = C.c_uint32
    L.orc_index_num_ids.argtypes = [C.c_void_p]
    L.orc_index_positions.restype = C.c_uint32
    L.orc_index_positions.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.orc_refs_load.restype = C.c_void_p
    L.orc_refs_load.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32]
    L.orc_refs_free.argtypes = [C.c_void_p]
    L.orc_refs_count.restype = C.c_uint32
    L.orc_refs_count.argtypes = [C.c_void_p]
    L.orc_refs_len.restype = C.c_uint32
    L.orc_refs_len.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_refs_seq.restype = C.c_void_p
    L.orc_refs_seq.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_stats_load.restype = C.c_int
    L.orc_stats_load.argtypes = [C.c_char_p, C.POINTER(Stats)]
    L.orc_minimal_score.restype = C.c_uint32
    L.orc_minimal_score.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), C.c_uint64, C.c_uint64,
                                    C.c_uint64, C.c_uint64, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.orc_batch_new.restype = C.c_void_p
    L.orc_batch_new.argtypes = [C.c_uint32]
    L.orc_batch_free.argtypes = [C.c_void_p]
    L.orc_batch_record.restype = C.c_size_t
    L.orc_batch_record.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
    L.orc_batch_is_hit.restype = C.c_int
    L.orc_batch_is_hit.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_align_part.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Params), C.c_char_p, C.c_void_p, C.c_uint32,
                                 C.c_void_p, C.POINTER(Counters)]
    L.orc_window_hits.restype = C.c_uint32
    L.orc_window_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p,
                                  C.c_uint32, C.POINTER(C.c_int)]
    L.orc_ssw.restype = C.c_int
    L.orc_ssw.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint8, C.c_uint8,
                          C.c_uint16, C.POINTER(SswResult)]
    _lib = L
    return L


def default_params(**kw):
    p = Params()
    p.lnwin = 18
    p.skiplengths[0], p.skiplengths[1], p.skiplengths[2] = 18, 9, 3
    p.num_seeds, p.min_lis, p.edges, p.is_as_percent = 2, 2, 4, 0
    p.match, p.mismatch, p.score_N, p.gap_open, p.gap_ext = 2, -3, -3, 5, 2
    p.minimal_score, p.num_alignments, p.is_best, p.is_full_search = 0, 1, 1, 0
    p.is_forward, p.is_reverse, p.minoccur = 1, 1, 0
    p.index_num, p.part, p.is_last_index_part = 0, 0, 1
    for k, v in kw.items():
        if k == "skiplengths":
            for i in range(3):
                p.skiplengths[i] = v[i]
        else:
            setattr(p, k, v)
    return p


def load_stats(prefix):
    st = Stats()
    rc = lib().orc_stats_load(prefix.encode(), C.byref(st))
    if rc != 0:
        raise IOError("cannot load %s.stats" % prefix)
    return st


def minimal_score(lam, K, st, all_reads_count, all_reads_len, evalue=1.0):
    fr = C.c_uint64()
    fd = C.c_uint64()
    ms = lib().orc_minimal_score(lam, K, st.bg, st.full_len, st.numseq, all_reads_count, all_reads_len, evalue,
                                 C.byref(fr), C.byref(fd))
    return ms, fr.value, fd.value


def pack_seqs(seqs):
    blob = "".join(seqs).encode("latin-1")
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    return blob, offs


class Run:
    """Runs the oracle over all (index, part) pairs like processor.cpp:align()."""

    def __init__(self, seqs):
        self.L = lib()
        self.seqs = seqs
        self.blob, self.offs = pack_seqs(seqs)
        self.batch = self.L.orc_batch_new(len(seqs))
        self.counters = Counters()

    def align_part(self, prefix, fasta, st, part, params):
        L = self.L
        ix = L.orc_index_load(prefix.encode(), part, st.lnwin)
        assert ix, "index load failed"
        rf = L.orc_refs_load(fasta.encode(), st.part_start[part], st.part_numseq[part])
        assert rf, "refs load failed"
        self.counters.num_short = 0  # processor.cpp:230
        L.orc_align_part(ix, rf, C.byref(params), self.blob, self.offs.ctypes.data, len(self.seqs), self.batch,
                         C.byref(self.counters))
        L.orc_index_free(ix)
        L.orc_refs_free(rf)

    def record(self, i):
        n = self.L.orc_batch_record(self.batch, i, None, 0)
        if n == 0:
            return b""
        buf = C.create_string_buffer(n)
        self.L.orc_batch_record(self.batch, i, buf, n)
        return buf.raw

    def records(self):
        return [self.record(i) for i in range(len(self.seqs))]

    def close(self):
        if self.batch:
            self.L.orc_batch_free(self.batch)
            self.batch = None
