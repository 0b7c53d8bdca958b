/*
 * smr_hip.h -- C ABI of libsmr_hip.so: the MI355X (gfx950) engine for SortMeRNA's per-read hot path.
 *
 * This is the drop-in boundary.  The reference is one C++17 executable with no plugin
 * API; the seam is the per-read call  traverse(opts, index, refs, readstats, refstats, read, isLastStrand)
 * made from align2()  (/root/reference/src/sortmerna/processor.cpp:85,104-161), once per
 * read x strand x index part, by the thread pool of align() (processor.cpp:173-285).  A maintainer
 * replaces that loop by batch calls into this library (binding sketch: INTEGRATION.md).
 *
 * Everything here is plain C: pointers, sizes, int error codes (0 = ok, <0 = error, message via
 * smr_last_error).  No exceptions, no C++ or torch types cross the ABI.
 *
 * What each entry point replaces in the reference:
 *   smr_index_*      Index::load + References::load          index.cpp:143-357, references.cpp:55-159
 *                    (+ our own builder = build_index()       indexdb.cpp:1119-2095)
 *   smr_reads_*      Readfeed::next -> Read(readstr).init()   readfeed.hpp:124, read.cpp:264-347
 *   smr_refstats_*   Refstats::load (.stats + minimal_score)  refstats.cpp:103-265  (Gumbel lambda,K are INPUTS:
 *                    the reference gets them from the vendored NCBI ALP library, refstats.cpp:194-233)
 *   smr_align_part   the N x align2() threads for one (index, part): traverse() -> traversetrie_align()
 *                    -> compute_lis_alignment() -> ssw_align()   paralleltraversal.cpp:81-298,
 *                    traverse_bursttrie.cpp:100-298, alignment.cpp:100-509, ssw.c:834-941
 *   smr_result_*     Read::toBinString() / kvdb.put()         read.cpp:429-462, processor.cpp:150-155
 *   smr_counters     Readstats atomics                        readstats.hpp:77-85
 *
 * Hard limits of this build (each is an explicit error -- SMR_ERR_CAPACITY / SMR_ERR_ARG with a message --, never a silent difference):
 *   reads                 <= 65 535 letters each; reads x windows of the finest pass < 2^31 per batch (150-nt reads: ~47 M; the benches use 8 M)
 *   resident batches      16 per context (smr_batch_select), resident index parts 64 per context (smr_index_upload slot 0..63)
 *   seed length           8..20, even; < 2^31 - 1 distinct seeds (ids) per index part
 *   seed hits             no limit: the lane-local hit lists grow to what a half-seed search can accept at most (31 L/2 - 20 strings, smr_prof.hit_list_cap)
 *   candidate references  <= 49 152 references sharing seeds with ONE read on one strand (the per-block global table of k_chain<EXT>)
 *   alignments per read   max_alignments_per_read given to smr_reads_upload (the reference's -num_alignments, or 256 for "all")
 *   scoring               match <= 127, |mismatch|, |score_N| <= 127, gaps <= 255.  With 2 * gap_open, 2 * gap_ext >= |mismatch|, gap_open > gap_ext and score_N <= 0 the
 *                         affine recurrence of the fast kernels equals the reference's striped kernels cell for cell; outside those conditions ssw.c's scores
 *                         depend on its SIMD stripe geometry (ssw.c:267,496 and its 16-bit lazy-F loop :496-507), and smr_align_part scores through a slow
 *                         path that reproduces that geometry (csrc/smr_sw_striped.hpp; about ten times slower; rounds 1 - 5 refused such schemes)
 *   edges                 1..10 letters or percent like the reference's --edges; a percentage must not round to 0 letters for any searchable read of the batch
 *   pools                 seed-hit pool <= 8 GiB, CIGAR pool < 2^32 words, pigeonhole arena < 2^34 words per part (all grown on demand)
 */
#ifndef SMR_HIP_H
#define SMR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMR_OK 0
#define SMR_ERR_ARG (-1)        /* bad argument / unsupported option value */
#define SMR_ERR_IO (-2)         /* file missing or malformed */
#define SMR_ERR_DEVICE (-3)     /* HIP error (no GPU, out of memory, kernel fault) */
#define SMR_ERR_CAPACITY (-4)   /* a device pool overflowed after automatic regrow attempts */
#define SMR_ERR_STATE (-5)      /* call order violated */

typedef struct smr_ctx smr_ctx;          /* one GPU: stream, resident index parts, resident read batch */
typedef struct smr_index smr_index;      /* HOST: one flattened index part + its reference sequences */
typedef struct smr_reads smr_reads;      /* HOST: a packed read batch */

/* Options of Runopts that reach the hot path (include/options.hpp:495-608; defaults options.cpp:1566-1758). */
typedef struct {
  uint32_t skiplengths[3];   /* -passes      pass strides; {0,0,0} => {L, L/2, 3}     refstats.cpp:159-166 */
  int32_t  num_seeds;        /* -num_seeds   2 */
  int32_t  min_lis;          /* -min_lis     2 */
  int32_t  edges;            /* -edges       4 */
  int32_t  is_as_percent;    /* -edges N%    0 */
  int32_t  match;            /* -match       2 */
  int32_t  mismatch;         /* -mismatch   -3 */
  int32_t  score_N;          /* -N           = mismatch */
  int32_t  gap_open;         /* -gap_open    5 */
  int32_t  gap_ext;          /* -gap_ext     2 */
  uint32_t num_alignments;   /* -num_alignments 1 (0 = all) */
  int32_t  is_best;          /* 1 unless -no-best */
  int32_t  is_full_search;   /* -full_search 0 */
  int32_t  is_forward;       /* -F */
  int32_t  is_reverse;       /* -R  (both 1 when neither given) */
  uint32_t minoccur;         /* 0 */
  /* per call of smr_align_part: */
  uint32_t minimal_score;    /* Refstats::minimal_score[index_num]            refstats.cpp:261-265 */
  uint32_t index_num;        /* position of the DB in --ref order */
  uint32_t part;             /* index part number */
  int32_t  is_last_index_part; /* last part of the last index                 paralleltraversal.cpp:294 */
} smr_params;

void smr_params_default(smr_params* p);
/* NULL when smr_align_part takes these options; otherwise the reason it will answer SMR_ERR_ARG (--edges outside 1..10: INTEGRATION.md "known
 * limits").  For a host's option parser, so that a driver says so before it loads anything.  (Scoring schemes under which ssw.c's striped kernels
 * leave the affine recurrence are no longer among them: round 6 scores them through the slow path of csrc/smr_sw_striped.hpp.) */
const char* smr_params_refused(const smr_params* p);

/* ------------------------------------------------------------------------------------------------
 * Index (host side).  An smr_index is one (index, part): 9-mer lookup, mini burst tries in a compact
 * word arena, positions CSR, reference sequences (0..4, 1 byte/nt).
 * ---------------------------------------------------------------------------------------------- */
/* Load a part written by the reference's indexer: <prefix>.kmer_P.dat / .bursttrie_P.dat / .pos_P.dat
 * (+ <prefix>.stats for the part's byte range in the FASTA).  Replaces Index::load + References::load. */
int smr_index_load_files(const char* prefix, uint32_t part, const char* ref_fasta, smr_index** out, char* err, size_t errcap);

/* Build ALL parts of the index of one FASTA ourselves (replaces build_index(), indexdb.cpp:1119-2095):
 * same 19-mer geometry, same forward/reverse mini-trie contents, ids = rank of the unique 18-mer
 * (any bijection is equivalent: ids are opaque keys into the positions table), positions in file
 * order truncated at max_pos.  n_parts_out parts are returned in parts_out[0..n_parts_out).
 * max_file_size_mb = the reference's -m (3072), seed_win_len = -L (18), max_pos = -max_pos (10000). */
int smr_index_build(const char* ref_fasta, uint32_t seed_win_len, double max_file_size_mb, uint32_t max_pos,
                    uint32_t threads, smr_index** parts_out, uint32_t cap_parts, uint32_t* n_parts_out,
                    char* err, size_t errcap);
/* Write one part in the REFERENCE's on-disk format (so the reference binary can consume our index). */
int smr_index_write_files(const smr_index* const* parts, uint32_t n_parts, const char* ref_fasta, const char* prefix,
                          char* err, size_t errcap);
/* Flat cache of one part's HOST layout (header + the arrays as they lie in memory, page aligned): smr_index_load_flat maps the file and
 * copies the arrays with all cores -- the reference-format files store no stream lengths (index.cpp:176-316) and cost a sequential walk
 * over GBs plus a parse (1.3 s for the 140 Mnt DB, 0.3 s from the cache).  `stamp` is the caller's key for "these reference files" (size
 * and mtime, a hash ...): load returns SMR_ERR_STATE when the file was written under another stamp, SMR_ERR_IO when it is absent or damaged;
 * the caller then loads the reference's files (or builds) and saves the cache for the next run. */
int smr_index_save(const smr_index*, const char* path, uint64_t stamp, char* err, size_t errcap);
int smr_index_load_flat(const char* path, uint64_t stamp, smr_index** out, char* err, size_t errcap);
/* Consistency check of the two device layouts of the mini-tries (reference-shaped arena for k_seed_search, pigeonhole arena for
 * k_seed_pg): the second must hold the same (candidate string, id) entries with their ranks in the DFS order of the first, sorted by its
 * two keys, under consistent directories.  0 = ok. */
int smr_index_selfcheck(smr_index*, char* err, size_t errcap);
void smr_index_free(smr_index*);

typedef struct {
  uint32_t lnwin;            /* L */
  uint32_t n_kmers;          /* 4^(L/2) */
  uint64_t trie_words;       /* u32 words in the mini-trie arena */
  uint32_t n_ids;            /* unique 18-mers */
  uint64_t n_pos;            /* total positions */
  uint32_t n_refs;           /* sequences in this part */
  uint64_t ref_bytes;        /* total nt in this part */
  uint64_t n_nodes, n_buckets, n_entries;
  double   bg[4];            /* A,C,G,T background frequencies of the whole DB (from .stats / builder) */
  uint64_t full_len;         /* total nt of the whole DB */
  uint64_t numseq;           /* total sequences of the whole DB */
  uint32_t n_parts;
} smr_index_info;
int smr_index_get_info(const smr_index*, smr_index_info* out);

/* Refstats::load arithmetic (refstats.cpp:238-265): minimal SW score for E-value `evalue` given the
 * Gumbel parameters of the (scoring scheme, background) pair and the GLOBAL read totals
 * (all_reads_count / all_reads_len are sums over every rank when reads are sharded).
 * lambda and K are INPUTS and there is no default: the reference computes them per DB and scoring scheme (-match / -mismatch / -gap_open /
 * -gap_ext, background frequencies) with its vendored NCBI ALP library (refstats.cpp:194-233) and prints them in its log ("Gumbel lambda",
 * "Gumbel K").  A different pair gives a different minimal_score, i.e. a different set of reads passes -- take them from the reference
 * (the compiled drop-in of INTEGRATION.md uses the reference's own Refstats object). */
uint32_t smr_minimal_score(double lambda, double K, const double bg[4], uint64_t full_ref_len, uint64_t numseq,
                           uint64_t all_reads_count, uint64_t all_reads_len, double evalue);
/* The same under -score_split (refstats.cpp:247: `full_read_scale = opts.is_score_split ? opts.num_proc_thread : 1`): the read totals are
 * divided by the number of processing threads the reference was run with; full_read_scale = 1 is smr_minimal_score. */
uint32_t smr_minimal_score_split(double lambda, double K, const double bg[4], uint64_t full_ref_len, uint64_t numseq,
                                 uint64_t all_reads_count, uint64_t all_reads_len, double evalue, uint32_t full_read_scale);

/* The length-corrected database / read sizes of Refstats (refstats.cpp:238-257) that the e-value of the BLAST report uses. */
void smr_refstats_corrected(double K, const double bg[4], uint64_t full_ref_len, uint64_t numseq, uint64_t all_reads_count, uint64_t all_reads_len,
                            uint64_t* full_ref_corr, uint64_t* full_read_corr);
void smr_refstats_corrected_split(double K, const double bg[4], uint64_t full_ref_len, uint64_t numseq, uint64_t all_reads_count, uint64_t all_reads_len,
                                  uint32_t full_read_scale, uint64_t* full_ref_corr, uint64_t* full_read_corr);

/* ------------------------------------------------------------------------------------------------
 * Reads (host side): 2-bit packed + ambiguity mask (Read::seqToIntStr: ACGT(U) -> 0..3, other -> 0
 * and the position is remembered, read.cpp:334-347).
 * ---------------------------------------------------------------------------------------------- */
/* seqs: concatenated ASCII sequences, offs[n+1] byte offsets. */
int smr_reads_pack(const char* seqs, const uint64_t* offs, uint32_t n_reads, smr_reads** out);
/* FASTA/FASTQ (optionally multi-line FASTA), plain text; [first, first+count) selects a record range
 * (count = 0 => to the end): the host-side read shard of one rank. */
int smr_reads_load_fastx(const char* path, uint64_t first, uint64_t count, smr_reads** out, char* err, size_t errcap);
/* The whole file (plain or gzip, izlib.cpp:95-210), parsed and packed by `threads` threads (0 = all cores) over byte ranges that start at record boundaries (the
 * reference splits the read file the same way into one range per thread, readfeed.cpp:1253-1277).  Same result as
 * smr_reads_load_fastx(path, 0, 0, ...). */
int smr_reads_load_fastx_mt(const char* path, uint32_t threads, smr_reads** out, char* err, size_t errcap);
void smr_reads_free(smr_reads*);
/* records [first, first + count) of a packed batch as a batch of their own (no text): the host-side split of the reads into one shard per
 * GPU (the reference splits the read file into one range per thread, readfeed.cpp:1253-1277) or into the chunks of an upload/align pipeline */
int smr_reads_slice(const smr_reads*, uint64_t first, uint64_t count, smr_reads** out);
/* like smr_reads_load_fastx_mt, and keeps the file text so that smr_reads_record_text can hand out every record's header line (as in the
 * file, with '>' / '@'), letters (line breaks removed) and quality line for the report writers; lens = {header, letters, quality} lengths;
 * a NULL / too small buffer is skipped / filled as far as it goes (always NUL-terminated) */
int smr_reads_load_fastx_text(const char* path, uint32_t threads, smr_reads** out, char* err, size_t errcap);
int smr_reads_is_fastq(const smr_reads*);
int smr_reads_record_text(const smr_reads*, uint32_t i, char* hdr, size_t hdr_cap, char* seq, size_t seq_cap, char* qual, size_t qual_cap, size_t lens[3]);
uint64_t smr_reads_digest(const smr_reads*);   /* hash of the packed batch (lengths, offsets, words): equal digests = same reads, same order */
uint32_t smr_reads_count(const smr_reads*);
uint64_t smr_reads_total_len(const smr_reads*);
uint32_t smr_reads_min_len(const smr_reads*);
uint32_t smr_reads_max_len(const smr_reads*);

/* ------------------------------------------------------------------------------------------------
 * Device context
 * ---------------------------------------------------------------------------------------------- */
int  smr_device_count(void);   /* HIP devices this process sees (0: none -- there is no CPU fallback); a host that spreads its read chunks over the GPUs of a node creates one context per device (the reference: one aligner thread per core, processor.cpp:248-256) */
int  smr_create(int device, smr_ctx** out, char* err, size_t errcap);
void smr_destroy(smr_ctx*);
const char* smr_last_error(const smr_ctx*);

/* Copy an index part to HBM; it stays resident under `slot` (0..63) until freed. */
int smr_index_upload(smr_ctx*, const smr_index*, int slot);
/* Test seam: the pigeonhole layout that smr_index_upload BUILT ON THE DEVICE for `slot` (from the uploaded lookup table and mini-trie arena;
 * it replaces a host pass over every mini-trie, indexdb.cpp's in-memory tries being what both start from) against the host transform of the
 * same index, word for word.  SMR_PG_HOST=1 makes smr_index_upload use the host transform instead. */
int smr_index_check_device(smr_ctx*, int slot, smr_index*);
/* Test seams of the roofline numerator's audit (tests/test_gpu_parity.py::test_pigeonhole_search_bytes_equal_a_host_recount): the pigeonhole
 * layout of a host index as the host transform builds it, and the sorted tuples of the context's last seed-stage launch with what decodes them
 * (meta = {tuples, forward tuples, coarse bins, fine key bits, char bits, forward keys, candidate records per wave, waves handed to the DFS kernel}). */
int smr_index_pigeonhole(smr_index*, const uint32_t** pg, uint64_t* pg_words, const uint32_t** root3, uint64_t* root3_words, char* err, size_t errcap);
int smr_seed_tuples_fetch(smr_ctx*, uint64_t* tuples, uint64_t cap_tuples, uint32_t* cbase, uint32_t cap_cbase, uint32_t meta[8]);
int smr_index_unload(smr_ctx*, int slot);

/* Several read batches (0..15) can be resident at once, so the host can upload batch k+1 while batch k is being
 * aligned (the reference's Readfeed hands reads to align2() one at a time, processor.cpp:104; here the unit is a
 * batch).  smr_batch_select picks the batch every later call acts on (default 0).  Each batch owns its per-read
 * state, its Readstats counter block and its CIGAR pool; the caller sums the counters of its batches. */
int smr_batch_select(smr_ctx*, int batch);

/* Seed-search kernel selection.  0 (default): the pigeonhole kernel k_seed_pg.  1: the per-lane DFS kernel k_seed_search for
 * every window; slower, but its work counters (smr_prof_get: n_node, n_entry) follow the reference's sequential scan exactly
 * (nothing after a 0-error match is counted) -- used to obtain the algorithmic byte counts of a workload.  Results are identical. */
int smr_set_seed_mode(smr_ctx*, int exact_counters);

/* Copy a read batch to HBM (into the selected batch) and allocate its persistent per-read state (what the reference keeps in
 * the KVDB between index parts, read.cpp:429-539).  Resets all state and counters. */
int smr_reads_upload(smr_ctx*, const smr_reads*, uint32_t max_alignments_per_read);
/* The same into batch `batch` (0..15) WITHOUT selecting it, on the context's second (upload) stream: a second host thread may call this
 * while the first one is inside smr_align_part / smr_traceback / smr_results_fetch of another batch -- upload of batch k+1 overlaps the
 * alignment of batch k (the reference's Readfeed/Processor overlap file reading with alignment the same way, readfeed.cpp, processor.cpp:248-256).
 * Fails with SMR_ERR_STATE when `batch` is the selected batch. */
int smr_reads_upload_batch(smr_ctx*, int batch, const smr_reads*, uint32_t max_alignments_per_read);
/* Forget all per-read results/counters of the resident batch (reads stay resident). */
int smr_state_reset(smr_ctx*);

/* The hot path for ONE (index, part) over the resident batch: both strands, all passes, LIS chaining,
 * Smith-Waterman scoring; commits per-read state exactly like processor.cpp:104-161 + kvdb.put.
 * Synchronous (returns after the GPU finished). */
int smr_align_part(smr_ctx*, int slot, const smr_params*);
/* Banded traceback -> CIGAR for every stored alignment that does not have one yet and whose
 * (index_num, part) is resident in `slot` (ssw.c:577-773).  Call after smr_align_part of that slot. */
int smr_traceback(smr_ctx*, int slot, const smr_params*);

/* Readstats counters (readstats.hpp:77-85): out[0]=num_aligned, out[1]=num_short (of the last part),
 * out[2+i]=reads_matched_per_db[i], i < n_db.  These are what the RCCL all-reduce sums across ranks. */
int smr_counters(smr_ctx*, uint64_t* out, uint32_t n_db);
/* Device pointer + count of the u64 counters block (for an in-place RCCL all-reduce by the caller). */
int smr_counters_device(smr_ctx*, void** dptr, uint32_t* n_u64);
/* d_acc[k] += counter k of the selected batch for k < n_u64 (<= the count smr_counters_device gives), on the device: a host that aligns its
 * shard chunk by chunk through a few recycled batches keeps one device block of sums (and all-reduces THAT over the ranks). */
int smr_counters_accumulate(smr_ctx*, void* d_acc, uint32_t n_u64);

/* Results.  smr_result_record writes Read::toBinString() bytes of read i (the KVDB value,
 * read.cpp:429-462; 0 bytes when the read has no alignment) and returns the size needed. */
int    smr_results_fetch(smr_ctx*);                       /* device -> host copy of all per-read results */
size_t smr_result_record(const smr_ctx*, uint32_t read_idx, uint8_t* buf, size_t cap);
/* The same for read i of batch `batch`, whichever batch is selected.  It only reads the host copy that smr_results_fetch made of that
 * batch, so a second host thread may serialise the records of batch k while the first one is aligning batch k+1 (the reference's writer
 * thread works the same way behind its aligners, output.cpp:169-272). */
size_t smr_result_record_batch(const smr_ctx*, int batch, uint32_t read_idx, uint8_t* buf, size_t cap);
int    smr_result_is_hit(const smr_ctx*, uint32_t read_idx);

/* Seed hits of the last smr_seed_scan call (kernel-level parity + roofline bench of the seed-scan kernel).
 * Runs ONLY the window-scan/burst-trie kernel for (strand, pass) over every read of the resident batch. */
int smr_seed_scan(smr_ctx*, int slot, const smr_params*, int strand, int pass, uint64_t* n_hits_out);
/* hits as (read_idx, id, win) triples in unspecified order */
int smr_seed_hits_fetch(smr_ctx*, uint32_t* triples, uint64_t cap_triples, uint64_t* n_out);

/* Timing/work counters accumulated since the last smr_prof_reset (HIP events on the engine's stream). */
typedef struct {
  double   seed_ms;  uint64_t seed_launches;   /* window-scan + burst-trie kernel */
  double   chain_ms; uint64_t chain_launches;  /* LIS chaining + SW scoring kernel */
  double   trace_ms; uint64_t trace_launches;  /* banded traceback kernel */
  /* exact work counters of the seed-scan kernel (SURVEY.md 8d byte formula) */
  uint64_t n_windows, n_lookup, n_node, n_entry, n_hit, n_read_bytes;
  uint64_t n_sw_fwd, n_sw_rev, n_sw_cells;     /* ssw_align calls of the sequential walk (forward / reverse passes) and their DP cells */
  uint64_t n_sw_spec, n_sw_spec_used;          /* forward passes scored ahead of the walk in four-problem batches, and how many of them the walk then asked for */
  uint64_t n_seed_redo;                        /* waves (64 searches) of the fast seed kernel whose candidate pool overflowed and that the per-lane DFS kernel searched again */
  uint64_t hit_list_cap;                       /* entries of the per-search hit lists in use: 4, doubled on demand up to 128, then 31 L/2 - 20 (twice that for the DFS kernel) = what a search can accept at most */
  uint64_t n_seed_shared;                      /* seed stages since smr_prof_reset whose searches walked the batch's SHARED sorted arrays (one sort for several index parts / --ref) */
  uint64_t n_seed_shared_builds;               /* ... and how often those six arrays were built */
} smr_prof;
/* smr_index_build with the per-occurrence work (sorting all (L+1)-mers, ids, position lists, mini-trie layout) done
 * on the device: same arguments (threads does not apply), same smr_index objects, byte-identical index files
 * (replaces build_index, indexdb.cpp:1119-2095). */
int smr_index_build_gpu(smr_ctx*, const char* ref_fasta, uint32_t lnwin, double max_mb, uint32_t max_pos,
                        smr_index** parts_out, uint32_t cap_parts, uint32_t* n_parts_out, char* err, size_t errcap);

/* Device self-check: n_cases seeded random (read, reference window) pairs, 1..max_len nt, for two scoring schemes: the packed 16-bit
 * Smith-Waterman kernel against the 32-bit kernel (score, end cell; forward and reverse pass), both on the GPU.  smr_create runs it
 * (SMR_SW_SELFCHECK=<cases>, 0 = skip) and falls back to the 32-bit kernel if any case differs; SMR_SW_PACKED=0 disables the packed kernel,
 * SMR_SW_PACKED=2 selects its wave_ror variant (the kernel checked is the selected one). */
int smr_sw_selfcheck(smr_ctx*, uint32_t n_cases, uint32_t seed, uint32_t max_len, uint64_t* n_bad);
int smr_sw_mode(smr_ctx*, int set_to);
/* The candidate walk (alignment.cpp:150-508) runs in rounds of walk kernel -> Smith-Waterman over a task list -> next list; the last round scores inside
 * the walk kernel, so the records never depend on the number.  out[pass] = rounds the next smr_align_part runs for that pass: at most 8 (SMR_WALK_ROUNDS=<n>
 * fixes it), lowered part by part towards what the previous part needed (an empty round still costs three launches). */
int smr_walk_rounds(const smr_ctx*, uint32_t out[3]);
/* The SW kernels at the ssw.h seam: for n independent pairs (read / reference window in the 0..4 alphabet, pair i = bytes [off[i], off[i+1])),
 * what ssw_align(prof, ref, refLen, gapO, gapE, flag = 2, filters, 0, 0) returns without the CIGAR (ssw.h:118-140, ssw.c:834-941):
 * out[5 i ..] = {score1, ref_begin1, ref_end1, read_begin1, read_end1}, begins = -1 when score1 < filters.  mode 0 / 1 / 2 = 32-bit / packed / packed wave_ror kernel, 3 = four pairs per wave (the fast kernels: only under the
 * schemes whose answers they share with ssw.c, SMR_ERR_ARG otherwise); mode 4 = the slow path that reproduces ssw.c's stripe geometry, any scheme. */
int smr_ssw_batch(smr_ctx*, uint32_t n_pairs, const uint8_t* reads, const uint64_t* read_off, const uint8_t* refs, const uint64_t* ref_off,
                  int match, int mismatch, int score_N, int gap_open, int gap_ext, uint32_t filters, int mode, int32_t* out);   /* set_to 0 / 1: use the 32-bit / the packed kernel; other values: query; returns the mode in use */
/* The traceback kernels at the same seam: for n independent triples (read window, reference window -- both exactly the aligned spans
 * [begin1, end1] -- and the alignment's score1) the CIGAR that banded_sw returns for them (ssw.c:577-773 as called from ssw_align,
 * ssw.c:919-926): BAM-style u32 operations (length << 4 | op, op 0/1/2 = M/I/D), pair i at cigar_out[cigar_off_out[i] .. cigar_off_out[i+1]).
 * Operations beyond cigar_cap are counted but not written.  Uses the same host logic and kernels as smr_traceback. */
int smr_cigar_batch(smr_ctx*, uint32_t n_pairs, const uint8_t* reads, const uint64_t* read_off, const uint8_t* refs, const uint64_t* ref_off,
                    const uint16_t* scores, int match, int mismatch, int score_N, int gap_open, int gap_ext,
                    uint32_t* cigar_out, uint64_t cigar_cap, uint64_t* cigar_off_out);
int smr_prof_reset(smr_ctx*);
int smr_prof_get(smr_ctx*, smr_prof* out);
/* The same period per kernel family (k_seed_keys, the tuple sort, k_seed_pg<0>, k_seed_pg<1>, k_seed_finish, k_cand, k_chain, k_begins,
 * k_trace): HIP-event time on the engine's stream, number of launches, and for the seed-stage kernels the ALGORITHMIC HBM bytes of what
 * the shipped kernels themselves do, from exact device counters: every tuple (12 B) written once by k_seed_keys next to its inputs
 * (read records, per-read state, two lookup words per window), read and written once by each of the two sort passes, read once by
 * k_seed_pg, which adds per search 8 B of block table, its directory words, 4 B per string looked at, 8 B per accepted {rank, id} and
 * the hit-segment words it reads and writes; k_seed_finish the segment words it gathers.  bytes = 0 where nothing is counted.
 * This -- not the traversal of the reference, which k_seed_pg does not perform -- is the numerator of bench.py's roofline. */
typedef struct { char name[32]; double ms; uint64_t launches; uint64_t bytes; } smr_kprof;
int smr_prof_kernels(smr_ctx*, smr_kprof* out, uint32_t cap, uint32_t* n_out);

/* ------------------------------------------------------------------------------------------------
 * Reports (host side): the reference's second pass over reads + KVDB (writeReports, output.cpp:169-272), fed by
 * smr_result_record.  Files in out_dir: aligned.fa|fq, other.fa|fq (report_fx_base.cpp:176-205), aligned.blast = BLAST tabular m8
 * with the optional columns of `-blast '1 cigar qcov qstrand'` (report_blast.cpp:253-354), aligned.sam (report_sam.cpp:64-152).
 * Rows are written per (index, part) in --ref order, within a part in the order the reads were added, like the reference's loop.
 * ---------------------------------------------------------------------------------------------- */
typedef struct smr_report smr_report;
typedef struct {
  int fastx;             /* -fastx   aligned.fa|fq */
  int other;             /* -other   other.fa|fq   */
  int blast_tabular;     /* -blast 1 ...           */
  char blast_cols[64];   /* optional BLAST columns, space separated, in output order: "cigar", "qcov", "qstrand" */
  int sam;               /* -sam                   */
  int blast_pairwise;    /* -blast 0: the BLAST-like pairwise text (report_blast.cpp:130-252) instead of tabular rows */
  int sam_sq;            /* -SQ: @SQ lines of every reference sequence in the SAM header (report_sam.cpp:155-211) */
  /* paired reads (two read files, or -paired_in / -paired_out): smr_report_add_pair routes the two mates like ReportFastx::append /
   * ReportFxOther::append (report_fastx.cpp:57-133, report_fx_other.cpp:49-113) */
  int paired_in;         /* -paired_in: a pair with one aligned mate goes to aligned.* entirely  */
  int paired_out;        /* -paired_out: ... goes to other.* entirely                              */
  int out2;              /* -out2: separate files for the mates: *_fwd / *_rev                     */
  int sout;              /* -sout: separate files for pairs and singletons: *_paired / *_singleton */
  int zip_out;           /* gzip every report file, names + ".gz" (the reference does so for gzip reads files or -zip-out 1, report_fx_base.cpp:94-95) */
} smr_report_opts;
int smr_report_open(const char* out_dir, const smr_report_opts*, int is_fastq, smr_report** out, char* err, size_t errcap);
/* per --ref: Gumbel parameters and the corrected sizes (smr_refstats_corrected); per (index, part): where its reference ids/sequences are */
int smr_report_set_db(smr_report*, uint32_t index_num, double lambda, double K, uint64_t full_ref_corr, uint64_t full_read_corr);
int smr_report_set_part(smr_report*, uint32_t index_num, uint32_t part, const smr_index*);
/* one read: its header line as in the file (with '>' / '@'), letters, quality (NULL for FASTA), and its record (NULL, 0: none) */
int smr_report_add(smr_report*, const char* header, const char* seq, const char* qual, const uint8_t* record, size_t record_len);
/* a pair of mates (read i of the first and of the second file / two consecutive records of an interleaved file) */
int smr_report_add_pair(smr_report*, const char* header1, const char* seq1, const char* qual1, const uint8_t* record1, size_t record1_len,
                        const char* header2, const char* seq2, const char* qual2, const uint8_t* record2, size_t record2_len);
int smr_report_set_cmdline(smr_report*, const char* cmdline);   /* text after "CL:" in the SAM @PG line (default "libsmr_hip") */
int smr_report_close(smr_report*);      /* writes aligned.blast / aligned.sam, closes the files, frees the object */
const char* smr_report_last_error(const smr_report*);

/* aligned.log: the run summary of Summary::to_string (summary.cpp:102-175), same text for the same numbers.
 * cmdline / pid / timestamp are printed as given (the reference prints its own command line, pid string and ctime()). */
typedef struct {
  const char* ref_file;      /* as given to --ref */
  uint32_t skiplengths[3];
  double lambda, K;          /* Gumbel parameters */
  uint32_t minimal_score;
  uint64_t reads_matched;    /* Readstats::reads_matched_per_db[i] */
} smr_summary_db;
typedef struct {
  const char* cmdline; const char* pid; const char* timestamp;
  uint32_t seed_len; int32_t num_seeds, edges, match, mismatch, gap_open, gap_ext, score_N; int32_t sam_sq; int32_t threads;
  const char* const* reads_files; uint32_t n_reads_files;
  uint64_t total_reads, num_aligned, all_reads_len; uint32_t min_read_len, max_read_len;
  const smr_summary_db* dbs; uint32_t n_dbs;
} smr_summary;
int smr_summary_write(const char* path, const smr_summary*);

/* Readstats persistence: the value Readstats::store_to_db puts into the KVDB after the alignment stage = Readstats::toBstring()
 * (readstats.cpp:133-174, 291-295) and its key = decimal std::hash of the '_'-joined basenames of the read files (readstats.cpp:82-91,
 * util.cpp:216-222).  Both return the size needed; the buffer is filled when it is large enough (the key NUL-terminated). */
size_t smr_readstats_record(uint64_t all_reads_count, uint64_t all_reads_len, uint32_t min_read_len, uint32_t max_read_len, uint64_t num_aligned,
                            uint64_t num_short, const uint64_t* reads_matched_per_db, uint32_t n_db, uint8_t* buf, size_t cap);
size_t smr_readstats_key(const char* const* reads_files, uint32_t n_files, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
