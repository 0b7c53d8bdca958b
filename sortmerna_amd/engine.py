"""Host-side mirror of the reference's alignment driver for the hot path.

`align()` below has the shape of processor.cpp:align() (/root/reference/src/sortmerna/processor.cpp:173-285):
for every index and every part: load index + references, run the per-read path over all reads, keep per-read
state between parts -- except that the inner N x align2() thread loop is one call into libsmr_hip (GPU).
"""
import ctypes as C

import numpy as np

from . import capi


class SmrError(RuntimeError):
    pass


def default_params(**kw):
    p = capi.Params()
    capi.load().smr_params_default(C.byref(p))
    for k, v in kw.items():
        if k == "skiplengths":
            for i in range(3):
                p.skiplengths[i] = v[i]
        else:
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
    return p


class Index:
    """One (index, part) on the host: flattened lookup / mini-trie arena / positions CSR / reference bytes."""

    def __init__(self, handle):
        self.h = handle

    @staticmethod
    def load_files(prefix, part, ref_fasta):
        L = capi.load()
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = L.smr_index_load_files(prefix.encode(), part, ref_fasta.encode(), C.byref(h), err, 512)
        if rc != 0:
            raise SmrError("smr_index_load_files: %s (rc=%d)" % (err.value.decode(), rc))
        return Index(h)

    @staticmethod
    def build(ref_fasta, seed_win_len=18, max_file_size_mb=3072.0, max_pos=10000, threads=0):
        """-> list of Index (one per part)"""
        L = capi.load()
        cap = 256
        arr = (C.c_void_p * cap)()
        n = C.c_uint32()
        err = C.create_string_buffer(512)
        rc = L.smr_index_build(ref_fasta.encode(), seed_win_len, max_file_size_mb, max_pos, threads,
                               C.cast(arr, C.POINTER(C.c_void_p)), cap, C.byref(n), err, 512)
        if rc != 0:
            raise SmrError("smr_index_build: %s (rc=%d)" % (err.value.decode(), rc))
        return [Index(C.c_void_p(arr[i])) for i in range(n.value)]

    @staticmethod
    def build_gpu(engine, ref_fasta, seed_win_len=18, max_file_size_mb=3072.0, max_pos=10000):
        """like build(), with the sorting / id / position / mini-trie work on the device (smr_index_build_gpu)"""
        L = capi.load()
        cap = 256
        arr = (C.c_void_p * cap)()
        n = C.c_uint32()
        err = C.create_string_buffer(512)
        rc = L.smr_index_build_gpu(engine.h, ref_fasta.encode(), seed_win_len, max_file_size_mb, max_pos,
                                   C.cast(arr, C.POINTER(C.c_void_p)), cap, C.byref(n), err, 512)
        if rc != 0:
            raise SmrError("smr_index_build_gpu: %s (rc=%d)" % (err.value.decode(), rc))
        return [Index(C.c_void_p(arr[i])) for i in range(n.value)]

    @staticmethod
    def write_files(parts, ref_fasta, prefix):
        L = capi.load()
        arr = (C.c_void_p * len(parts))(*[p.h for p in parts])
        err = C.create_string_buffer(512)
        rc = L.smr_index_write_files(C.cast(arr, C.POINTER(C.c_void_p)), len(parts), ref_fasta.encode(), prefix.encode(), err, 512)
        if rc != 0:
            raise SmrError("smr_index_write_files: %s (rc=%d)" % (err.value.decode(), rc))

    def save(self, path, stamp=0):
        """flat cache of this part's host layout (smr_index_save)"""
        err = C.create_string_buffer(512)
        rc = capi.load().smr_index_save(self.h, path.encode(), stamp, err, 512)
        if rc != 0:
            raise SmrError("smr_index_save: %s (rc=%d)" % (err.value.decode(), rc))

    @staticmethod
    def load_flat(path, stamp=0):
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = capi.load().smr_index_load_flat(path.encode(), stamp, C.byref(h), err, 512)
        if rc != 0:
            raise SmrError("smr_index_load_flat: %s (rc=%d)" % (err.value.decode(), rc))
        return Index(h)

    def selfcheck(self):
        """the pigeonhole device layout holds the same entries, with their DFS ranks, as the reference-shaped one"""
        err = C.create_string_buffer(512)
        rc = capi.load().smr_index_selfcheck(self.h, err, 512)
        if rc != 0:
            raise SmrError("smr_index_selfcheck: %s (rc=%d)" % (err.value.decode(), rc))

    def info(self):
        i = capi.IndexInfo()
        capi.load().smr_index_get_info(self.h, C.byref(i))
        return i

    def free(self):
        if self.h:
            capi.load().smr_index_free(self.h)
            self.h = None


class Reads:
    def __init__(self, handle):
        self.h = handle

    @staticmethod
    def from_seqs(seqs):
        L = capi.load()
        blob = "".join(seqs).encode("latin-1")
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        if seqs:
            offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
        h = C.c_void_p()
        rc = L.smr_reads_pack(blob, offs.ctypes.data, len(seqs), C.byref(h))
        if rc != 0:
            raise SmrError("smr_reads_pack rc=%d" % rc)
        return Reads(h)

    @staticmethod
    def from_fastx(path, first=0, count=0):
        L = capi.load()
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = L.smr_reads_load_fastx(path.encode(), first, count, C.byref(h), err, 512)
        if rc != 0:
            raise SmrError("smr_reads_load_fastx: %s (rc=%d)" % (err.value.decode(), rc))
        return Reads(h)

    @staticmethod
    def from_fastx_mt(path, threads=0):
        """whole file, parsed and packed by `threads` threads (0 = all cores)"""
        L = capi.load()
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = L.smr_reads_load_fastx_mt(path.encode(), threads, C.byref(h), err, 512)
        if rc != 0:
            raise SmrError("smr_reads_load_fastx_mt: %s (rc=%d)" % (err.value.decode(), rc))
        return Reads(h)

    @staticmethod
    def from_fastx_text(path, threads=0):
        """from_fastx_mt + the file text kept, so that record_text(i) returns (header line, letters, quality) for the report writers"""
        L = capi.load()
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = L.smr_reads_load_fastx_text(path.encode(), threads, C.byref(h), err, 512)
        if rc != 0:
            raise SmrError("smr_reads_load_fastx_text: %s (rc=%d)" % (err.value.decode(), rc))
        return Reads(h)

    def slice(self, first, count):
        """records [first, first+count) as a batch of their own (the read shard of one rank / one pipeline chunk)"""
        h = C.c_void_p()
        rc = capi.load().smr_reads_slice(self.h, first, count, C.byref(h))
        if rc != 0:
            raise SmrError("smr_reads_slice rc=%d" % rc)
        return Reads(h)

    @property
    def is_fastq(self):
        return bool(capi.load().smr_reads_is_fastq(self.h))

    def record_text(self, i):
        L = capi.load()
        lens = (C.c_size_t * 3)()
        rc = L.smr_reads_record_text(self.h, i, None, 0, None, 0, None, 0, lens)
        if rc != 0:
            raise SmrError("smr_reads_record_text rc=%d (batch not loaded with from_fastx_text?)" % rc)
        bufs = [C.create_string_buffer(lens[k] + 1) for k in range(3)]
        L.smr_reads_record_text(self.h, i, bufs[0], lens[0] + 1, bufs[1], lens[1] + 1, bufs[2], lens[2] + 1, lens)
        return tuple(b.value.decode() for b in bufs)

    @property
    def digest(self):
        return capi.load().smr_reads_digest(self.h)

    @property
    def count(self):
        return capi.load().smr_reads_count(self.h)

    @property
    def total_len(self):
        return capi.load().smr_reads_total_len(self.h)

    @property
    def min_len(self):
        return capi.load().smr_reads_min_len(self.h)

    @property
    def max_len(self):
        return capi.load().smr_reads_max_len(self.h)

    def free(self):
        if self.h:
            capi.load().smr_reads_free(self.h)
            self.h = None


def minimal_score(lam, K, info, all_reads_count, all_reads_len, evalue=1.0, full_read_scale=1):
    """Refstats arithmetic (refstats.cpp:238-265) from Gumbel (lambda, K) + DB statistics + GLOBAL read totals; full_read_scale = the
    reference's processing threads under -score_split (refstats.cpp:247), else 1."""
    return capi.load().smr_minimal_score_split(lam, K, info.bg, info.full_len, info.numseq, all_reads_count, all_reads_len, evalue, full_read_scale)


def pigeonhole_layout(index):
    """(pg uint32[], root3 uint32[]) of a host index as the host transform builds it (smr_index_pigeonhole: a test seam; views into the index)"""
    L = capi.load()
    pg, r3 = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
    npg, nr3 = C.c_uint64(), C.c_uint64()
    err = C.create_string_buffer(512)
    rc = L.smr_index_pigeonhole(index.h, C.byref(pg), C.byref(npg), C.byref(r3), C.byref(nr3), err, 512)
    if rc != 0:
        raise SmrError("smr_index_pigeonhole: %s (rc=%d)" % (err.value.decode(), rc))
    return np.ctypeslib.as_array(pg, shape=(npg.value,)), np.ctypeslib.as_array(r3, shape=(nr3.value,))


class Engine:
    """One GPU.  Fails loudly when no HIP device / library is available (no CPU fallback)."""

    def __init__(self, device=0):
        self.L = capi.load()
        h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = self.L.smr_create(device, C.byref(h), err, 512)
        if rc != 0:
            raise SmrError("smr_create: %s (rc=%d)" % (err.value.decode(), rc))
        self.h = h
        self.n_reads = 0
        self._batch_n = {}
        self._cur = 0

    def _chk(self, rc, what):
        if rc != 0:
            raise SmrError("%s: %s (rc=%d)" % (what, self.L.smr_last_error(self.h).decode(), rc))

    def upload_index(self, index, slot=0):
        self._chk(self.L.smr_index_upload(self.h, index.h, slot), "smr_index_upload")

    def check_device_index(self, index, slot=0):
        """the pigeonhole layout built on the device for `slot` == the host transform of the same index (raises SmrError on a difference)"""
        self._chk(self.L.smr_index_check_device(self.h, slot, index.h), "smr_index_check_device")

    def unload_index(self, slot=0):
        self._chk(self.L.smr_index_unload(self.h, slot), "smr_index_unload")

    def select_batch(self, batch):
        self._chk(self.L.smr_batch_select(self.h, batch), "smr_batch_select")
        self.n_reads = self._batch_n.get(batch, 0)
        self._cur = batch

    def set_seed_mode(self, exact_counters):
        """0: pigeonhole seed search (default); 1: per-lane DFS kernel with reference-exact work counters (same results)"""
        self._chk(self.L.smr_set_seed_mode(self.h, int(bool(exact_counters))), "smr_set_seed_mode")

    def sw_mode(self, set_to=-1):
        """Smith-Waterman kernel in use: 1 = packed 16-bit (default when the device self-check passes), 0 = 32-bit; set_to 0/1 selects"""
        return self.L.smr_sw_mode(self.h, set_to)

    def walk_rounds(self):
        """rounds of the candidate walk the next align_part runs, per pass (smr_walk_rounds)"""
        out = (C.c_uint32 * 3)()
        self._chk(self.L.smr_walk_rounds(self.h, out), "smr_walk_rounds")
        return list(out)

    def sw_selfcheck(self, n_cases=256, seed=1, max_len=700):
        """packed vs 32-bit Smith-Waterman kernel on n_cases random pairs x 2 scoring schemes, on the device; returns the number of differing cases"""
        bad = C.c_uint64()
        self._chk(self.L.smr_sw_selfcheck(self.h, n_cases, seed, max_len, C.byref(bad)), "smr_sw_selfcheck")
        return bad.value

    def ssw_batch(self, reads, refs, match=2, mismatch=-3, score_N=-3, gap_open=5, gap_ext=2, filters=0, mode=1):
        """reads / refs: lists of byte strings in the 0..4 alphabet; -> int32 array (n, 5): score1, ref_begin1, ref_end1, read_begin1, read_end1
        (ssw_align with flag 2, without the CIGAR), computed by the 32-bit (mode 0) or the packed (mode 1) SW kernel"""
        import numpy as np
        n = len(reads)
        ro = np.zeros(n + 1, dtype=np.uint64); fo = np.zeros(n + 1, dtype=np.uint64)
        ro[1:] = np.cumsum([len(x) for x in reads]); fo[1:] = np.cumsum([len(x) for x in refs])
        rb = np.frombuffer(b"".join(reads) + b"\0", dtype=np.uint8).copy(); fb = np.frombuffer(b"".join(refs) + b"\0", dtype=np.uint8).copy()
        out = np.zeros((n, 5), dtype=np.int32)
        self._chk(self.L.smr_ssw_batch(self.h, n, rb.ctypes.data, ro.ctypes.data, fb.ctypes.data, fo.ctypes.data, match, mismatch, score_N,
                                       gap_open, gap_ext, filters, mode, out.ctypes.data), "smr_ssw_batch")
        return out

    def cigar_batch(self, reads, refs, scores, match=2, mismatch=-3, score_N=-3, gap_open=5, gap_ext=2):
        """reads / refs: the aligned spans (byte strings in the 0..4 alphabet), scores: their score1; -> list of u32 CIGAR arrays (len << 4 | op),
        what the reference's banded_sw returns for each triple (the traceback kernels behind smr_traceback)"""
        n = len(reads)
        ro = np.zeros(n + 1, dtype=np.uint64); fo = np.zeros(n + 1, dtype=np.uint64)
        ro[1:] = np.cumsum([len(x) for x in reads]); fo[1:] = np.cumsum([len(x) for x in refs])
        rb = np.frombuffer(b"".join(reads) + b"\0", dtype=np.uint8).copy(); fb = np.frombuffer(b"".join(refs) + b"\0", dtype=np.uint8).copy()
        sc = np.asarray(scores, dtype=np.uint16)
        off = np.zeros(n + 1, dtype=np.uint64)
        cap = int(ro[-1] + fo[-1]) + 4 * n + 16
        out = np.zeros(cap, dtype=np.uint32)
        self._chk(self.L.smr_cigar_batch(self.h, n, rb.ctypes.data, ro.ctypes.data, fb.ctypes.data, fo.ctypes.data, sc.ctypes.data, match, mismatch, score_N,
                                         gap_open, gap_ext, out.ctypes.data, cap, off.ctypes.data), "smr_cigar_batch")
        return [out[int(off[i]):int(off[i + 1])].copy() for i in range(n)]

    def upload_reads(self, reads, max_alignments_per_read=1):
        self._chk(self.L.smr_reads_upload(self.h, reads.h, max_alignments_per_read), "smr_reads_upload")
        self.n_reads = reads.count
        self._batch_n[self._cur] = reads.count

    def upload_reads_batch(self, batch, reads, max_alignments_per_read=1):
        """smr_reads_upload_batch: into batch `batch` without selecting it, on the context's upload stream (may run on a second host thread
        while the selected batch is being aligned)"""
        self._chk(self.L.smr_reads_upload_batch(self.h, batch, reads.h, max_alignments_per_read), "smr_reads_upload_batch")
        self._batch_n[batch] = reads.count

    def reset_state(self):
        self._chk(self.L.smr_state_reset(self.h), "smr_state_reset")

    def align_part(self, slot, params):
        self._chk(self.L.smr_align_part(self.h, slot, C.byref(params)), "smr_align_part")

    def traceback(self, slot, params):
        self._chk(self.L.smr_traceback(self.h, slot, C.byref(params)), "smr_traceback")

    def counters(self, n_db=1):
        out = (C.c_uint64 * (2 + n_db))()
        self._chk(self.L.smr_counters(self.h, out, n_db), "smr_counters")
        return dict(num_aligned=out[0], num_short=out[1], reads_matched_per_db=[out[2 + i] for i in range(n_db)])

    def counters_device(self):
        p = C.c_void_p()
        n = C.c_uint32()
        self._chk(self.L.smr_counters_device(self.h, C.byref(p), C.byref(n)), "smr_counters_device")
        return p.value, n.value

    def fetch(self):
        self._chk(self.L.smr_results_fetch(self.h), "smr_results_fetch")

    def record(self, i):
        n = self.L.smr_result_record(self.h, i, None, 0)
        if n == 0:
            return b""
        buf = C.create_string_buffer(n)
        self.L.smr_result_record(self.h, i, buf, n)
        return buf.raw

    def records(self):
        return [self.record(i) for i in range(self.n_reads)]

    def record_batch(self, batch, i):
        """record of read i of batch `batch`, whichever batch is selected (smr_result_record_batch: a writer thread's call)"""
        n = self.L.smr_result_record_batch(self.h, batch, i, None, 0)
        if n == 0:
            return b""
        buf = C.create_string_buffer(n)
        self.L.smr_result_record_batch(self.h, batch, i, buf, n)
        return buf.raw

    def is_hit(self, i):
        return bool(self.L.smr_result_is_hit(self.h, i))

    def seed_scan(self, slot, params, strand, pass_):
        n = C.c_uint64()
        self._chk(self.L.smr_seed_scan(self.h, slot, C.byref(params), strand, pass_, C.byref(n)), "smr_seed_scan")
        return n.value

    def seed_hits(self):
        n = C.c_uint64()
        self._chk(self.L.smr_seed_hits_fetch(self.h, None, 0, C.byref(n)), "smr_seed_hits_fetch")
        arr = np.zeros((max(n.value, 1), 3), dtype=np.uint32)
        self._chk(self.L.smr_seed_hits_fetch(self.h, arr.ctypes.data, n.value, C.byref(n)), "smr_seed_hits_fetch")
        return arr[: n.value]

    def seed_tuples(self):
        """(sorted tuples uint64[n], cbase uint32[nc + 1], meta dict) of the last seed-stage launch (smr_seed_tuples_fetch: a test seam)"""
        meta = (C.c_uint32 * 8)()
        self._chk(self.L.smr_seed_tuples_fetch(self.h, None, 0, None, 0, meta), "smr_seed_tuples_fetch")
        tup = np.zeros(max(meta[0], 1), dtype=np.uint64)
        cbase = np.zeros(meta[2] + 1, dtype=np.uint32)
        self._chk(self.L.smr_seed_tuples_fetch(self.h, tup.ctypes.data, len(tup), cbase.ctypes.data, len(cbase), meta), "smr_seed_tuples_fetch")
        return tup[: meta[0]], cbase, dict(n=meta[0], n_fwd=meta[1], nc=meta[2], fb=meta[3], cb=meta[4], nkh=meta[5], ccap=meta[6], redo=meta[7])

    def prof_reset(self):
        self._chk(self.L.smr_prof_reset(self.h), "smr_prof_reset")

    def prof(self):
        p = capi.Prof()
        self._chk(self.L.smr_prof_get(self.h, C.byref(p)), "smr_prof_get")
        return p

    def prof_kernels(self):
        """{kernel family: {"ms", "launches", "bytes"}} since the last prof_reset (smr_prof_kernels)"""
        a = (capi.Kprof * 16)()
        n = C.c_uint32()
        self._chk(self.L.smr_prof_kernels(self.h, a, 16, C.byref(n)), "smr_prof_kernels")
        return {a[i].name.decode(): {"ms": a[i].ms, "launches": int(a[i].launches), "bytes": int(a[i].bytes)} for i in range(n.value)}

    def close(self):
        if self.h:
            self.L.smr_destroy(self.h)
            self.h = None


def align(engine, reads, index_parts, params_per_index, with_cigar=True, max_alignments_per_read=None):
    """processor.cpp:align(): index_parts = [[Index part0, part1, ...] per --ref], params_per_index = [Params per --ref]
    (each carrying that DB's minimal_score).  Returns nothing; results stay in `engine` (fetch()/record())."""
    p0 = params_per_index[0]
    slots = max_alignments_per_read or (p0.num_alignments if p0.num_alignments > 0 else 32)
    engine.upload_reads(reads, slots)
    n_idx = len(index_parts)
    for idx_num, parts in enumerate(index_parts):
        for part, ix in enumerate(parts):
            p = params_per_index[idx_num]
            p.index_num = idx_num
            p.part = part
            p.is_last_index_part = int(idx_num == n_idx - 1 and part == len(parts) - 1)
            engine.upload_index(ix, 0)
            engine.align_part(0, p)
            if with_cigar:
                engine.traceback(0, p)
            engine.unload_index(0)
    engine.fetch()


def align_resident(engine, index_slots, params_per_index, with_cigar=True):
    """Same loop over (index, part) for reads AND index parts that are already resident in HBM: index_slots is either a
    flat list of slots (one --ref, its parts in order) or a list of such lists (one per --ref).  Acts on the selected
    batch; the caller resets its state first when the batch is reused."""
    if index_slots and not isinstance(index_slots[0], (list, tuple)):
        index_slots = [index_slots]
    n_idx = len(index_slots)
    for idx_num, slots in enumerate(index_slots):
        for part, slot in enumerate(slots):
            p = params_per_index[idx_num]
            p.index_num = idx_num
            p.part = part
            p.is_last_index_part = int(idx_num == n_idx - 1 and part == len(slots) - 1)
            engine.align_part(slot, p)
            if with_cigar:
                engine.traceback(slot, p)
    engine.fetch()
