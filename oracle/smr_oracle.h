/*
 * smr_oracle.h -- TEST INFRASTRUCTURE, NOT THE PRODUCT.
 *
 * Plain-C CPU restatement of SortMeRNA's per-read hot path (reference v5.0.0 under /root/reference):
 *   traverse()                src/sortmerna/paralleltraversal.cpp:81-298
 *   traversetrie_align()      src/sortmerna/traverse_bursttrie.cpp:100-298 (LEV(1) tables :68-98)
 *   init_win_f / init_win_r   src/sortmerna/bitvector.cpp:57-132
 *   compute_lis_alignment()   src/sortmerna/alignment.cpp:100-509, find_lis :58-98
 *   ssw_init / ssw_align      src/sortmerna/ssw.c:788-941 (sw_sse2_byte :150-373, sw_sse2_word :399-575,
 *                             banded_sw :577-773)
 *   align2() loop body        src/sortmerna/processor.cpp:93-168
 *   Read encode/state         src/sortmerna/read.cpp:264-401,429-462,601-611
 *   Index::load               src/sortmerna/index.cpp:143-357
 *   References::load          src/sortmerna/references.cpp:55-159
 *   Refstats minimal_score    src/sortmerna/refstats.cpp:238-265
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * PARITY PIN: tests/test_oracle_vs_reference.py compares this restatement, record for record,
 * with the KVDB records (Read::toBinString bytes) produced by the unmodified reference built as
 * oracle/_ref/sortmerna_ref, and with the reference's golden vectors t0/t2/t9 (tests/golden/).
 */
#ifndef SMR_ORACLE_H
#define SMR_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_index orc_index;   /* one index part: lookup table, mini burst tries, positions */
typedef struct orc_refs  orc_refs;    /* reference sequences of one index part, 0..4 alphabet      */
typedef struct orc_batch orc_batch;   /* persistent per-read state (what the reference keeps in KVDB) */

/* Run options that reach the hot path (include/options.hpp:495-608, options.cpp:1566-1758). */
typedef struct {
  uint32_t lnwin;            /* seed length L (18)                          refstats.cpp:147 */
  uint32_t skiplengths[3];   /* pass strides {L, L/2, 3}                    refstats.cpp:159-166 */
  int32_t  num_seeds;        /* 2 */
  int32_t  min_lis;          /* 2 */
  int32_t  edges;            /* 4 */
  int32_t  is_as_percent;    /* 0 */
  int32_t  match, mismatch, score_N;   /* 2, -3, -3 */
  int32_t  gap_open, gap_ext;          /* 5, 2 */
  uint32_t minimal_score;    /* Refstats::minimal_score[index_num] */
  uint32_t num_alignments;   /* 1 */
  int32_t  is_best;          /* 1 */
  int32_t  is_full_search;   /* 0 */
  int32_t  is_forward, is_reverse;     /* both 1 when neither -F nor -R */
  uint32_t minoccur;         /* 0 */
  uint32_t index_num, part;  /* which (index, part) this call processes */
  int32_t  is_last_index_part; /* (index_num == last) && (part == last part) paralleltraversal.cpp:294 */
} orc_params;

typedef struct {
  uint64_t num_aligned;      /* Readstats::num_aligned */
  uint64_t num_short;        /* Readstats::num_short (caller resets per part, processor.cpp:230) */
  uint64_t reads_matched_per_db[64];
  /* work counters for the algorithmic-bytes formula */
  uint64_t n_lookup, n_node, n_entry, n_hit, n_sw_fwd, n_sw_rev, n_traceback, n_windows;
} orc_counters;

/* ---- loading ---- */
orc_index* orc_index_load(const char* prefix, uint32_t part, uint32_t lnwin);
void       orc_index_free(orc_index*);
uint32_t   orc_index_num_ids(const orc_index*);
uint32_t   orc_index_positions(const orc_index*, uint32_t id, uint32_t* pos_seq_pairs, uint32_t cap_pairs);

orc_refs*  orc_refs_load(const char* fasta, uint64_t start_part, uint32_t numseq_part);
void       orc_refs_free(orc_refs*);
uint32_t   orc_refs_count(const orc_refs*);
uint32_t   orc_refs_len(const orc_refs*, uint32_t i);
const char* orc_refs_seq(const orc_refs*, uint32_t i);

/* .stats file (indexdb.cpp:2033-2080 writer, refstats.cpp:103-186 reader) */
typedef struct {
  uint64_t filesize;
  double   bg[4];
  uint64_t full_len;
  uint32_t lnwin;
  uint64_t numseq;
  uint16_t nparts;
  uint64_t part_start[256], part_bytes[256];
  uint32_t part_numseq[256];
} orc_stats;
int orc_stats_load(const char* prefix, orc_stats* out);

/* refstats.cpp:238-265: length-corrected sizes + minimal SW score for an E-value. */
uint32_t orc_minimal_score(double lambda, double K, const double bg[4], uint64_t full_ref, uint64_t numseq,
                           uint64_t all_reads_count, uint64_t all_reads_len, double evalue,
                           uint64_t* full_ref_corr, uint64_t* full_read_corr);

/* ---- per-read state ---- */
orc_batch* orc_batch_new(uint32_t n_reads);
void       orc_batch_free(orc_batch*);
/* Read::toBinString() bytes of read i (0 if the read has no alignment); returns needed size. */
size_t     orc_batch_record(const orc_batch*, uint32_t i, uint8_t* buf, size_t cap);
int        orc_batch_is_hit(const orc_batch*, uint32_t i);

/* align2() over a batch for ONE (index, part): seqs = concatenated raw read sequences (ASCII),
 * offs[n+1] their offsets.  Mutates batch state exactly as processor.cpp:104-161 + kvdb.put. */
void orc_align_part(const orc_index*, const orc_refs*, const orc_params*,
                    const char* seqs, const uint64_t* offs, uint32_t n_reads,
                    orc_batch*, orc_counters*);

/* ---- unit-level entry points (kernel parity tests) ---- */
/* seed hits of ONE window: iseq = read in 0..3 alphabet; returns number of ids written
 * (paralleltraversal.cpp:131-249 for a single win_pos); *zero_err = accept_zero_kmer. */
uint32_t orc_window_hits(const orc_index*, const uint8_t* iseq, uint32_t win_pos, uint32_t lnwin,
                         uint32_t minoccur, int is_full_search, uint32_t* ids, uint32_t cap, int* zero_err);

/* the LEV(1) automaton over one complete (pattern, text) pair; see smr_oracle.c */
uint32_t orc_lev_accepts(uint32_t pchars, uint32_t tchars, uint32_t partialwin);
uint32_t orc_lev_alive_depth(uint32_t pchars, uint32_t tchars, uint32_t partialwin);

typedef struct {
  uint16_t score1; int32_t ref_begin1, ref_end1, read_begin1, read_end1;
  uint32_t cigar_len; uint32_t cigar[4096];
} orc_ssw_result;
/* ssw_init(read, readLen, mat, 5, 2) + ssw_align(..., flag=2, filters, 0, 0)  (alignment.cpp:365-381).
 * returns 0 if ssw_align would return NULL. */
int orc_ssw(const int8_t* read, int32_t readLen, const int8_t* ref, int32_t refLen, const int8_t* mat5x5,
            uint8_t gap_open, uint8_t gap_ext, uint16_t filters, orc_ssw_result* out);

#ifdef __cplusplus
}
#endif
#endif
