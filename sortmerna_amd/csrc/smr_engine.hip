// smr_engine.hip -- device context of libsmr_hip: HBM residency of index parts and the read batch,
// kernel orchestration for one (index, part), capacity regrow/retry, results and profiling.
// Compiled with hipcc --offload-arch=gfx950.  The C ABI is declared in include/smr_hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "smr_kernels.hpp"
#include "smr_ibuild.hpp"
#include "smr_pgbuild.hpp"
#include "smr_hostmem.hpp"

using namespace smr;

namespace {

struct DevIndex {
  bool used = false;
  Lookup* lookup = nullptr; uint32_t* trie = nullptr; uint32_t* pos_off = nullptr; uint2* pos_arr = nullptr;
  uint32_t* pg = nullptr; uint32_t* root3 = nullptr; uint32_t* lkc = nullptr;
  uint8_t* ref_seq = nullptr; uint64_t* ref_off = nullptr;
  uint32_t n_refs = 0, n_ids = 0, lnwin = 0;
  uint32_t ref_any_n = 1;     // does any reference hold an ambiguous letter (0 only when the upload looked and found none)
  uint64_t trie_words = 0, n_pos = 0, ref_bytes = 0, pg_words = 0;
};

struct EvMark { hipEvent_t e; int kind; };        // kind < 0: end of a run of intervals

}  // namespace

#define SMR_MAX_BATCHES 16
// kernel families timed apart (one HIP event between them on the engine's stream)
enum { KP_KEYS = 0, KP_SPLIT, KP_BINS, KP_PG0, KP_PG1, KP_FINISH, KP_CAND, KP_CHAIN, KP_BEGINS, KP_TRACE, KP_WALK, KP_SW16, KP_WNEXT, KP_COUNT };
static const char* const KP_NAME[KP_COUNT] = {"k_seed_keys", "k_seed_split", "k_seed_bins", "k_seed_pg<0>", "k_seed_pg<1>", "k_seed_finish", "k_cand", "k_chain", "k_begins", "k_trace",
                                              "k_walk", "k_sw16", "k_wnext"};

// One resident read batch: packed reads + everything the reference keeps per read in the KVDB (read.cpp:429-539)
// + its Readstats counter block + its CIGAR pool.  Several batches can be resident at once (the host uploads
// batch k+1 while batch k is being aligned); smr_batch_select picks the one the other calls act on.
struct Batch {
  bool used = false;
  uint32_t n = 0, max_len = 0, slots = 1;
  uint32_t min_ge[7] = {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u};     // the shortest read of at least 8, 10, ... 20 letters (~0: none): what `--edges N%` is smallest for
  uint64_t gen = 0;                        // counts the uploads and state resets of this batch (what a shared seed sort was built for)
  bool min_ge_known = false;               // (computed when a call with is_as_percent asks for it: one pass over the lengths, which the default options never need)
  uint32_t* d_words = nullptr; uint64_t* d_rec_off = nullptr; uint32_t* d_len = nullptr;
  RState* d_saved = nullptr; RState* d_work = nullptr; RWork* d_rw = nullptr;
  uint8_t* d_marks = nullptr;              // per read: k_chain has to walk it in this (strand, pass) (k_cand)
  AlignRec* d_saved_aln = nullptr; AlignRec* d_work_aln = nullptr;
  unsigned long long* d_ctr = nullptr;
  uint32_t* d_cigar = nullptr; uint64_t cigar_words = 0;
  // host copies of results
  std::vector<RState> h_state; std::vector<AlignRec> h_aln; std::vector<uint32_t> h_cigar;       // of the reads with alignments, packed
  std::vector<uint32_t> h_idx, h_map;                                                            // packed position -> read, read -> packed position (or ~0)
  uint32_t last_num_alignments = 1;
  unsigned long long redo_seen = 0, win_seen = 0;        // C_SEED_REDO / C_WINDOWS at the end of the previous part (smr_align_part sizes k_seed_pg's candidate pool from the increments)
  bool fetched = false;
  // capacities of the device arrays above (grow-only: a re-upload into the same batch allocates nothing unless it is larger)
  size_t cap_words = 0, cap_reads = 0, cap_aln = 0;
};

// (one seed sort for the index parts of a batch: see ensure_shared_sort)
struct SharedSet { SeedTup* srt = nullptr; uint16_t* wbin = nullptr; uint32_t* cbase = nullptr; uint32_t* sn = nullptr; uint32_t maxwin = 0; bool built = false; };
struct SharedSort {
  SharedSet set[2][3];
  const void* batch = nullptr; uint64_t gen = 0; uint32_t lnwin = 0, skip[3] = {0, 0, 0}, n = 0, max_len = 0;
  uint64_t cap[3] = {0, 0, 0};
  bool usable = false;
  uint32_t* abits = nullptr; size_t abits_words = 0;
};

struct smr_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;                        // last error; written under err_m (smr_reads_upload_batch runs on a second host thread)
  std::mutex err_m, sel_m;                // sel_m: which batch is selected (read by smr_reads_upload_batch)
  int n_cu = 256;
  DevIndex idx[64];
  Batch bt[SMR_MAX_BATCHES];
  Batch* b = &bt[0];
  // pools / scratch (shared by all batches: one batch is aligned at a time)
  uint32_t* d_pool = nullptr; uint64_t pool_words = 0;
  uint32_t hcap = 4;                      // lane-local hit list capacity of k_seed_search; doubles (and the part is redone) on overflow
  int seed_exact = 0;                     // 1: k_seed_search for every wave (exact work counters); 0: k_seed_pg (+ redo of the waves whose pool overflowed)
  // Bloom words per read in k_cand (a power of two, 64..512): fewer = more blocks of k_cand per CU, but more reads marked for k_chain by a false
  // collision.  Measured per 2 M-read launch (profiles/r3s18_*): 512 words k_cand 1.07 ms + k_chain 6.01 ms, 256: 0.59 + 6.06, 128: 0.48 + 6.07
  // (16 KB of LDS per block: the 8 blocks per CU that the wave slots allow)
  uint32_t cand_bloom = 128;
  // repeated seeds (k_seed_dedup): keys with at least this many tuples in a launch (and four times the average) are searched once per different seed; 0: off
  uint32_t hot_min = getenv("SMR_SEED_DEDUP") ? (uint32_t)std::max(0, atoi(getenv("SMR_SEED_DEDUP"))) : 1024u;
  // one seed sort for the index parts of a batch (SharedSort below): 0 off, 1 when the part in hand is not the batch's last, 2 always (tests)
  int seed_shared = getenv("SMR_SEED_SHARED") ? atoi(getenv("SMR_SEED_SHARED")) : 1;
  uint16_t* d_sw_scr = nullptr; uint32_t sw_scr_stride = 0; size_t sw_scr_words = 0;      // scratch rows of the striped Smith-Waterman slow path (smr_sw_striped.hpp), one per block
  int last_seed_slot = -1;                 // index slot of the last smr_seed_scan (smr_seed_hits_fetch translates its ids back)
  SharedSort* shared = nullptr;
  uint64_t n_seed_shared = 0, n_seed_shared_builds = 0;      // smr_prof
  uint32_t ccap = PG_CAND_CAP0;           // candidate records per wave of k_seed_pg; doubles when more than 1/64 of the waves of a part overflow
  // k_seed_pg: waves of the launch (0: one per wave chunk the batch can have; else a wave walks chunks it, it + grid, ...), XCD-aware chunk order
  uint32_t pg_grid = getenv("SMR_PG_GRID") ? (uint32_t)atoi(getenv("SMR_PG_GRID")) : 262144u;
  int pg_swz = getenv("SMR_PG_SWZ") ? atoi(getenv("SMR_PG_SWZ")) : 0;
  SeedBufs sb = {};                       // seed-stage scratch (smr_seed.hpp)
  uint64_t sb_slots = 0; uint32_t sb_nk = 0;
  uint32_t chain_blocks = 0;
  unsigned long long* d_tuples = nullptr; uint32_t chain_scap = 512;   // (pos, slot, win) tuples; slots of the candidate set S in LDS
  uint32_t keys_need = 0;
  bool chain_ext = false;                                              // a read's candidate set has outgrown the LDS table once: global tables are on
  uint32_t* d_stab = nullptr; unsigned long long* d_tuples2 = nullptr; // per block: CH_EXT_CAP-slot table (4 arrays), tuples grouped by member
  size_t chain_lds_attr = 0, begins_lds_attr = 0, split_lds_attr = 0, bins_lds_attr = 0;
  uint32_t* d_fidx = nullptr; RState* d_fstate = nullptr; AlignRec* d_faln = nullptr; size_t fetch_cap_r = 0, fetch_cap_a = 0;   // staging of smr_results_fetch
  int sw_mode = getenv("SMR_SW_PACKED") ? atoi(getenv("SMR_SW_PACKED")) : 2;   // 1 / 2: packed 16-bit Smith-Waterman kernels (smr_sw_pk.hpp; 2 = lane hand-over by wave_ror, measured faster) where they apply
  unsigned long long* d_keys = nullptr; uint32_t keys_cap = 0;
  unsigned long long* d_pairs = nullptr; uint32_t* d_lis = nullptr; uint32_t pairs_cap = 0;
  uint2* d_hits = nullptr; uint32_t hits_cap = 0;
  uint8_t* d_rdq = nullptr; size_t rdq_cap = 0;
  // k_cand -> k_chain hand-over (smr_chain.hpp): {offset, npos} per read, the records (SMR_HANDOVER=0 switches it off)
  int handover = getenv("SMR_HANDOVER") ? atoi(getenv("SMR_HANDOVER")) : 1;
  uint2* d_mrec = nullptr; uint32_t* d_mpool = nullptr; size_t mrec_cap = 0, mpool_words = 0;
  // the candidate walk in rounds (smr_walk.hpp): walk kernel -> Smith-Waterman over a task list -> next list; SMR_WALK_SPLIT=0: k_chain walks every marked read
  int walk_split = getenv("SMR_WALK_SPLIT") ? atoi(getenv("SMR_WALK_SPLIT")) : 1;
  uint32_t walk_rounds = getenv("SMR_WALK_ROUNDS") ? (uint32_t)std::max(1, std::min(32, atoi(getenv("SMR_WALK_ROUNDS")))) : 8u;      // the last one scores in the kernel
  uint32_t walk_k = getenv("SMR_WALK_K") ? (uint32_t)std::max(1, std::min((int)WK_MAX, atoi(getenv("SMR_WALK_K")))) : 4u;           // tasks a read leaves per round, at least (smr_walk.hpp walk_tasks_per_read)
  int walk_gather = getenv("SMR_WALK_GATHER") ? atoi(getenv("SMR_WALK_GATHER")) : 1;     // 0: only reads with a record of k_cand go through the rounds (at most 64 positions)
  uint32_t walk_assume = getenv("SMR_WALK_ASSUME") ? (uint32_t)atoi(getenv("SMR_WALK_ASSUME")) : 3u;                                 // round 0 predicts "aligns" from this many seeds of the best candidate
  uint2* d_wlist[2] = {nullptr, nullptr}; WState* d_wstate[2] = {nullptr, nullptr}; WTask* d_wtask[2] = {nullptr, nullptr}; uint2* d_wres[2] = {nullptr, nullptr};
  uint32_t* d_wtidx = nullptr; uint32_t* d_wslow = nullptr; unsigned long long* d_wctr = nullptr; size_t walk_cap = 0; uint32_t walk_kcap = 0, walk_rcap = 0;
  size_t walk_lds_attr = 0, pg_lds_attr = 0, search_lds_attr = 0;
  // rounds per (strand, pass): without SMR_WALK_ROUNDS the number adapts to what the previous part needed (the last round with more than a few
  // reads listed + the closing one: an empty round still costs three launches, ~70 us of stream time; 8 -> 4 rounds = 3 % of the bench step)
  bool walk_rounds_fixed = getenv("SMR_WALK_ROUNDS") != nullptr;
  uint32_t walk_need[3] = {0, 0, 0};
  unsigned long long* d_wstat = nullptr; uint32_t wstat_n = 0; int wstat_pass[8] = {}; uint32_t wstat_rm[8] = {};
  int* d_bound = nullptr; size_t bound_cap = 0;                        // strip-boundary rows of the SW kernels (reads of more than one strip), per block
  uint32_t* d_tasks = nullptr; uint64_t tasks_cap = 0;
  uint8_t* d_trflags = nullptr; uint64_t trflags_bytes = 0;            // direction flags of k_trace_wide (one tile per block)
  int* d_trrows = nullptr; uint64_t trrows_ints = 0;                   // its DP rows when the band does not fit LDS
  // profiling
  std::vector<EvMark> events; std::vector<hipEvent_t> ev_pool;
  double kp_ms[KP_COUNT] = {}; uint64_t kp_l[KP_COUNT] = {};       // HIP-event time and launches per kernel family (smr_prof_kernels)
  hipStream_t upload_stream = nullptr;     // smr_reads_upload_batch: H2D of batch k+1 while batch k is aligned on `stream`
  unsigned long long* d_ctr_snap = nullptr; // counters of the selected batch at the start of smr_align_part (restored when an attempt is redone)
};
struct KpSave { double ms[KP_COUNT]; uint64_t l[KP_COUNT]; };

#define SEED_REDO_CAP 16384u

namespace {

#define HIPCHK(ctx, call)                                                                        \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      std::lock_guard<std::mutex> l_((ctx)->err_m);                                              \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
      return SMR_ERR_DEVICE;                                                                     \
    }                                                                                            \
  } while (0)

// every write of the context's error string goes through here (smr_reads_upload_batch runs on a second host thread)
void set_err(smr_ctx* c, const std::string& msg) { std::lock_guard<std::mutex> l_(c->err_m); c->err = msg; }

template <class T> int dev_alloc(smr_ctx* c, T** p, size_t count) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (count == 0) count = 1;
  HIPCHK(c, hipMalloc((void**)p, count * sizeof(T)));
  return SMR_OK;
}
template <class T> void dev_free(T** p) { if (*p) { (void)hipFree(*p); *p = nullptr; } }


const char* scheme_unsupported(int mismatch, int score_N, int gap_open, int gap_ext);
DParams make_dparams(const smr_ctx* c, const DevIndex& di, const smr_params* p) {
  DParams P;
  memset(&P, 0, sizeof P);
  P.lnwin = di.lnwin; P.partialwin = di.lnwin / 2;
  if (p->skiplengths[0] == 0 || p->skiplengths[1] == 0 || p->skiplengths[2] == 0) {   // refstats.cpp:159-166
    P.skip[0] = di.lnwin; P.skip[1] = di.lnwin / 2; P.skip[2] = 3;
  } else { P.skip[0] = p->skiplengths[0]; P.skip[1] = p->skiplengths[1]; P.skip[2] = p->skiplengths[2]; }
  P.num_seeds = p->num_seeds; P.min_lis = p->min_lis; P.edges = p->edges; P.is_as_percent = p->is_as_percent;
  P.match = p->match; P.mismatch = p->mismatch; P.score_N = p->score_N; P.gap_open = p->gap_open; P.gap_ext = p->gap_ext;
  P.minimal_score = p->minimal_score; P.num_alignments = p->num_alignments;
  P.is_best = p->is_best; P.is_full_search = p->is_full_search; P.is_forward = p->is_forward; P.is_reverse = p->is_reverse;
  P.minoccur = p->minoccur; P.index_num = p->index_num; P.part = p->part; P.is_last_index_part = p->is_last_index_part;
  P.slots = c->b->slots;
  // a scheme under which ssw.c's striped kernels leave the affine recurrence: the slow path that reproduces their stripe geometry (smr_sw_striped.hpp)
  P.sw_mode = scheme_unsupported(p->mismatch, p->score_N, p->gap_open, p->gap_ext) ? -1 : c->sw_mode;
  P.sw_scratch = c->d_sw_scr; P.sw_scratch_stride = c->sw_scr_stride;
  return P;
}

// nullptr when the FAST Smith-Waterman kernels here (the affine recurrence H = max(0, diag + s, E, F)) give ssw.c's results under the scheme, else why not
// -- such a scheme goes through the slow path that reproduces ssw.c's stripe geometry (smr_sw_striped.hpp; rounds 1 - 5 refused it).
// (1) The reference's striped kernels store E before the lazy-F loop has raised H (ssw.c:267,496): a gap in one sequence directly after a gap in
// the other is missed when the second crosses a SIMD stripe.  Under 2 * gap_open >= |mismatch| and 2 * gap_ext >= |mismatch| such paths are never
// optimal.  (2) Its 16-bit kernel leaves the lazy-F loop as soon as no lane has F - gap_ext > H - gap_open (ssw.c:496-507); with gap_open <= gap_ext
// that is already true one cell behind a stripe boundary, so a longer gap across a boundary is lost whenever the score needs the 16-bit kernel
// (>= 255 - bias: a 150-nt read's good alignments).  Measured against ssw.c itself on seeded pairs: 2 / -3 / 3 / 3 and 2 / -3 / 2 / 3 differ in
// 1 - 5 % of the high-scoring pairs, every scheme with gap_open > gap_ext in none (tests/test_oracle_golden.py).
const char* scheme_unsupported(int mismatch, int score_N, int gap_open, int gap_ext) {
  const int mm = std::max(-mismatch, -std::min(score_N, 0));
  if (2 * gap_open < mm || 2 * gap_ext < mm) return "scoring scheme outside the supported range (2*gap_open and 2*gap_ext must be >= |mismatch|)";
  // (3) rows and columns beyond the end of a sequence are scored as N by the packed kernels, which is harmless as long as N never adds to a score
  // (found by tools/fuzz_emu.py: with score_N = +1 and ambiguous letters in the reference a reverse pass ran one row past the start of a read)
  if (score_N > 0) return "scoring scheme outside the supported range (score_N must be <= 0: the kernels pad sequences with N)";
  if (gap_open <= gap_ext) return "scoring scheme outside the supported range (gap_open must be greater than gap_ext: the reference's 16-bit kernel ends its lazy-F loop early otherwise, ssw.c:496-507)";
  return nullptr;
}

// sw: the call scores with the Smith-Waterman kernels (the banded traceback alone is exact under every scheme: smr_cigar_batch)
int check_params(smr_ctx* c, const smr_params* p, bool sw = true) {
  if (!p) { set_err(c, "null params"); return SMR_ERR_ARG; }
  if (p->index_num >= 64) { set_err(c, "index_num must be < 64"); return SMR_ERR_ARG; }
  if ((uint64_t)p->minoccur >= 0x3FFFFFFFull) { set_err(c, "minoccur must be < 2^30 - 1"); return SMR_ERR_ARG; }
  if (p->num_seeds < 1 || p->gap_open < 0 || p->gap_ext < 0 || p->match <= 0 || p->mismatch > 0) { set_err(c, "bad scoring/seed options"); return SMR_ERR_ARG; }
  // the reference's scoring matrix is int8_t (ssw_init, ssw.h:88); the SW kernel keeps a row's scores as 4 signed bytes
  if (p->match > 127 || p->mismatch < -127 || p->score_N > 127 || p->score_N < -127 || p->gap_open > 255 || p->gap_ext > 255) { set_err(c, "scores must fit int8 / gaps uint8 like the reference's"); return SMR_ERR_ARG; }
  // --edges: the reference's parser takes 1..10, nucleotides or percent (options.cpp:676); its geometry is not defined outside (0: `tail > edges - 1` is an
  // unsigned compare, alignment.cpp:320,345 -- the whole rest of the reference becomes the window; larger values: more reads whose window length wraps)
  if (sw && (p->edges < 1 || p->edges > 10)) { set_err(c, "edges must be 1..10 (nucleotides or percent), like the reference's --edges"); return SMR_ERR_ARG; }
  if (p->num_alignments > 0 && p->num_alignments > c->b->slots) { set_err(c, "num_alignments exceeds max_alignments_per_read given to smr_reads_upload"); return SMR_ERR_ARG; }
  return SMR_OK;
}

DIndex dindex(const DevIndex& d) {
  DIndex x; x.lookup = d.lookup; x.trie = d.trie; x.pg = d.pg; x.lkc = d.lkc; x.root3 = reinterpret_cast<const uint2*>(d.root3); x.pos_off = d.pos_off; x.pos_arr = d.pos_arr; x.ref_seq = d.ref_seq; x.ref_off = d.ref_off;
  x.n_refs = d.n_refs; x.n_ids = d.n_ids; x.lnwin = d.lnwin; x.partialwin = d.lnwin / 2; x.ref_any_n = d.ref_any_n;
  return x;
}
DReads dreads(const smr_ctx* c) { DReads r; r.words = c->b->d_words; r.rec_off = c->b->d_rec_off; r.len = c->b->d_len; r.n = c->b->n; r.max_len = c->b->max_len; return r; }

int ensure_chain_scratch(smr_ctx* c, const DevIndex& di) {
  if (c->chain_blocks == 0) c->chain_blocks = (uint32_t)c->n_cu * (getenv("SMR_CHAIN_WPC") ? atoi(getenv("SMR_CHAIN_WPC")) : 12);     // k_chain: 3 waves per SIMD by registers
  (void)di;
  uint32_t need_keys = std::max(std::max(c->chain_scap, 1024u), c->keys_need);
  if (c->keys_cap < need_keys) { int rc = dev_alloc(c, &c->d_keys, (size_t)c->chain_blocks * need_keys); if (rc) return rc; c->keys_cap = need_keys; }
  if (c->pairs_cap == 0) c->pairs_cap = 4096;
  if (!c->d_pairs) {
    int rc = dev_alloc(c, &c->d_pairs, (size_t)c->chain_blocks * c->pairs_cap); if (rc) return rc;
    rc = dev_alloc(c, &c->d_lis, (size_t)c->chain_blocks * 2 * c->pairs_cap); if (rc) return rc;
    rc = dev_alloc(c, &c->d_tuples, (size_t)c->chain_blocks * c->pairs_cap); if (rc) return rc;
  }
  if (c->chain_ext) {
    if (!c->d_stab) { int rc = dev_alloc(c, &c->d_stab, (size_t)c->chain_blocks * 4 * CH_EXT_CAP); if (rc) return rc; }
    if (!c->d_tuples2) { int rc = dev_alloc(c, &c->d_tuples2, (size_t)c->chain_blocks * c->pairs_cap); if (rc) return rc; }
  }
  if (c->hits_cap == 0) c->hits_cap = 4096;
  if (!c->d_hits) { int rc = dev_alloc(c, &c->d_hits, (size_t)c->chain_blocks * c->hits_cap); if (rc) return rc; }
  return SMR_OK;
}

// A mark = one pooled HIP event on the engine's stream.  The time between a mark of kind k >= 0 and the next mark belongs to kernel
// family k; ev_stop ends a run.  (Kernels of one stream run back to back anyway: a mark costs a barrier packet, no bubble.)
void ev_mark(smr_ctx* c, int kind) {
  EvMark m; m.kind = kind;
  if (!c->ev_pool.empty()) { m.e = c->ev_pool.back(); c->ev_pool.pop_back(); }
  else (void)hipEventCreate(&m.e);
  (void)hipEventRecord(m.e, c->stream);
  c->events.push_back(m);
}
void ev_stop(smr_ctx* c) { ev_mark(c, -1); }
void ev_collect(smr_ctx* c) {
  for (size_t i = 0; i + 1 < c->events.size(); i++) {
    const int k = c->events[i].kind;
    float ms = 0;
    if (k >= 0 && hipEventElapsedTime(&ms, c->events[i].e, c->events[i + 1].e) == hipSuccess) { c->kp_ms[k] += ms; c->kp_l[k]++; }
  }
  for (auto& m : c->events) c->ev_pool.push_back(m.e);
  c->events.clear();
}
// marks a call that failed half-way left behind (an error return between ev_mark and ev_stop): dropped at the start of the next timed call, or
// ev_collect would pair the dangling mark with the first one of an unrelated run and charge the gap to a kernel family
void ev_drop(smr_ctx* c) {
  for (auto& m : c->events) c->ev_pool.push_back(m.e);
  c->events.clear();
}
KpSave kp_save(const smr_ctx* c) { KpSave k; memcpy(k.ms, c->kp_ms, sizeof k.ms); memcpy(k.l, c->kp_l, sizeof k.l); return k; }
void kp_restore(smr_ctx* c, const KpSave& k) { memcpy(c->kp_ms, k.ms, sizeof k.ms); memcpy(c->kp_l, k.l, sizeof k.l); }

uint32_t num_windows(uint32_t max_len, uint32_t L, uint32_t stride) {
  return max_len >= L ? (max_len - L + stride) / stride : 1;
}

#include "smr_engine_seed.hpp"
#include "smr_engine_chain.hpp"
}  // namespace


// =================================================================================================
// Device index build (kernels in smr_ibuild.hpp)
// =================================================================================================
#include "smr_engine_ibuild.hpp"

#include "smr_engine_swseam.hpp"

// =================================================================================================
extern "C" int smr_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

extern "C" int smr_create(int device, smr_ctx** out, char* err, size_t errcap) {
  if (!out) return SMR_ERR_ARG;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    if (err && errcap) snprintf(err, errcap, "no HIP device available (%s): libsmr_hip has no CPU fallback", e == hipSuccess ? "count=0" : hipGetErrorString(e));
    return SMR_ERR_DEVICE;
  }
  if (device < 0 || device >= ndev) { if (err && errcap) snprintf(err, errcap, "device %d out of range (%d devices)", device, ndev); return SMR_ERR_ARG; }
  auto c = new smr_ctx();
  c->shared = new SharedSort();
  c->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess || hipStreamCreate(&c->upload_stream) != hipSuccess ||
      hipMalloc((void**)&c->d_ctr_snap, C_TOTAL * 8) != hipSuccess) {
    if (err && errcap) snprintf(err, errcap, "cannot initialise device %d", device);
    delete c->shared; delete c; return SMR_ERR_DEVICE;
  }
  if (const char* e = getenv("SMR_SEED_EXACT")) c->seed_exact = atoi(e) != 0;
  if (const char* e = getenv("SMR_CAND_BLOOM")) { uint32_t b = 64; while (b < CAND_BLOOM_WORDS && b < (uint32_t)atoi(e)) b <<= 1; c->cand_bloom = b; }      // measurement aid
  if (const char* e = getenv("SMR_PG_CAND_CAP")) c->ccap = std::min<uint32_t>(PG_CAND_CAP_MAX, std::max<uint32_t>(4u, (uint32_t)atoi(e)));      // test aid: a small candidate pool
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
  if (hipMalloc((void**)&c->b->d_ctr, C_TOTAL * 8) != hipSuccess) { if (err && errcap) snprintf(err, errcap, "hipMalloc failed"); delete c; return SMR_ERR_DEVICE; }
  (void)hipMemset(c->b->d_ctr, 0, C_TOTAL * 8);
  c->b->used = true;
  if (c->sw_mode >= 1) {
    // the packed Smith-Waterman kernel must agree with the 32-bit kernel on this device, or it is not used
    uint32_t cases = SMR_SW_SELFCHECK_CASES;
    if (const char* e2 = getenv("SMR_SW_SELFCHECK")) cases = (uint32_t)atoi(e2);
    uint64_t bad = 0;
    if (cases > 0 && (smr_sw_selfcheck(c, cases, 20260926u, 700, &bad) != SMR_OK || bad != 0)) {
      fprintf(stderr, "libsmr_hip: packed Smith-Waterman kernel disagrees with the 32-bit kernel on %llu self-check cases; using the 32-bit kernel\n", (unsigned long long)bad);
      c->sw_mode = 0;
    }
  }
  *out = c;
  return SMR_OK;
}

extern "C" void smr_destroy(smr_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (int s = 0; s < 64; s++) if (c->idx[s].used) smr_index_unload(c, s);
  for (int k = 0; k < SMR_MAX_BATCHES; k++) {
    Batch& B = c->bt[k];
    dev_free(&B.d_words); dev_free(&B.d_rec_off); dev_free(&B.d_len);
    dev_free(&B.d_saved); dev_free(&B.d_work); dev_free(&B.d_rw); dev_free(&B.d_marks); dev_free(&B.d_saved_aln); dev_free(&B.d_work_aln); dev_free(&B.d_ctr);
    dev_free(&B.d_cigar);
  }
  dev_free(&c->d_bound); dev_free(&c->d_rdq); dev_free(&c->d_mrec); dev_free(&c->d_mpool);
  for (int q = 0; q < 2; q++) { dev_free(&c->d_wlist[q]); dev_free(&c->d_wstate[q]); dev_free(&c->d_wtask[q]); dev_free(&c->d_wres[q]); }
  dev_free(&c->d_wstat);
  dev_free(&c->d_wtidx); dev_free(&c->d_wslow); dev_free(&c->d_wctr);
  dev_free(&c->sb.chist); dev_free(&c->sb.cbase); dev_free(&c->sb.rows); dev_free(&c->sb.bcnt); dev_free(&c->sb.tmp); dev_free(&c->sb.mid);
  dev_free(&c->sb.srt); dev_free(&c->sb.hpre); dev_free(&c->sb.hlist); dev_free(&c->sb.hh); dev_free(&c->sb.pieces); dev_free(&c->sb.redo); dev_free(&c->sb.sn); dev_free(&c->sb.wbin); dev_free(&c->sb.emap); dev_free(&c->sb.zbits); dev_free(&c->sb.gflag);
  for (int d = 0; d < 2; d++) { dev_free(&c->sb.wseg[d]); dev_free(&c->sb.fbits[d]); }
  dev_free(&c->d_pool); dev_free(&c->d_tuples); dev_free(&c->d_tuples2); dev_free(&c->d_stab); dev_free(&c->d_keys); dev_free(&c->d_pairs); dev_free(&c->d_lis); dev_free(&c->d_hits);
  dev_free(&c->d_tasks); dev_free(&c->d_trflags); dev_free(&c->d_trrows); dev_free(&c->d_sw_scr);
  if (c->shared) {
    for (int s = 0; s < 2; s++) for (int p = 0; p < 3; p++) { SharedSet& T = c->shared->set[s][p]; dev_free(&T.srt); dev_free(&T.wbin); dev_free(&T.cbase); dev_free(&T.sn); }
    dev_free(&c->shared->abits);
    delete c->shared;
  }
  for (auto& m : c->events) (void)hipEventDestroy(m.e);
  for (auto& e : c->ev_pool) (void)hipEventDestroy(e);
  dev_free(&c->d_ctr_snap); dev_free(&c->d_fidx); dev_free(&c->d_fstate); dev_free(&c->d_faln);
  (void)hipStreamDestroy(c->upload_stream);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" const char* smr_last_error(const smr_ctx* c) {
  if (!c) return "null context";
  static thread_local std::string copy;                  // the caller's own copy: another host thread may be setting the next error
  { std::lock_guard<std::mutex> l(const_cast<smr_ctx*>(c)->err_m); copy = c->err; }
  return copy.c_str();
}

namespace {
// The pigeonhole layout of the part in slot d (its lookup table and reference-shaped arena are on the device already): smr_pgbuild.hpp
// The position lists as the kernels read them (round 6): a hit's id IS the place of its list -- id' = pos_off[id] + id --, where a header word
// {positions, original id} stands in front of the positions {pos, seq}.  k_cand / k_walk / k_chain went hit -> pos_off[id], pos_off[id + 1] ->
// pos_arr[...]: two random 128-byte lines and two dependent round trips per hit; now the header and (nearly always) the positions share one line.
__global__ void __launch_bounds__(256) k_pos2_build(const uint32_t* __restrict__ pos_off, const uint2* __restrict__ pos_arr, uint32_t n_ids, uint2* __restrict__ pos2) {
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n_ids) return;
  const uint32_t lo = pos_off[id], n = pos_off[id + 1] - lo;
  uint2* o = pos2 + (size_t)lo + id;
  o[0] = make_uint2(n, id);
  for (uint32_t q = 0; q < n; q++) o[1 + q] = pos_arr[lo + q];
}

int build_pigeonhole_device(smr_ctx* c, DevIndex& d, uint32_t nk, uint32_t pw) {
  DevPool pool;
  const uint32_t nb = 2 * nk;
  IB_GET(cnt, uint32_t, nb); IB_GET(eoff, uint32_t, nb); IB_GET(words, smr::u64, nb); IB_GET(woff, smr::u64, nb); IB_GET(derr, uint32_t, 1);
  HIPCHK(c, hipMemsetAsync(derr, 0, 4, c->stream));
  hipLaunchKernelGGL(smr::k_pgb_sizes, dim3((nb + 255) / 256), dim3(256), 0, c->stream, (const Lookup*)d.lookup, (const uint32_t*)d.trie, nk, pw, cnt, words);
  smr::u64 W = 0;
  int rc = dev_scan<smr::u64>(c, pool, words, woff, nb, &W); if (rc) return rc;
  if (W / 3 > 0xFFFFFFF0ull || W / 4 > 0xFFFFFFF0ull) { set_err(c, "pigeonhole arena exceeds 2^34 words"); return SMR_ERR_CAPACITY; }
  uint32_t E = 0;
  if ((rc = dev_scan<uint32_t>(c, pool, cnt, eoff, nb, &E))) return rc;
  if ((rc = dev_alloc(c, &d.pg, (size_t)W + 4))) return rc;
  if ((rc = dev_alloc(c, &d.root3, (size_t)2 * nb))) return rc;
  HIPCHK(c, hipMemsetAsync(d.pg + W, 0, 16, c->stream));                 // one block of slack: a 16-byte read at the last word stays inside
  IB_GET(estr, uint32_t, E); IB_GET(eid, uint32_t, E); IB_GET(eblk, uint32_t, E);
  IB_GET(k0, smr::u64, E); IB_GET(k1, smr::u64, E); IB_GET(v0, uint32_t, E); IB_GET(v1, uint32_t, E);
  hipLaunchKernelGGL(smr::k_pgb_collect, dim3((nb + 255) / 256), dim3(256), 0, c->stream, (const Lookup*)d.lookup, (const uint32_t*)d.trie, nk, pw,
                     (const uint32_t*)cnt, (const uint32_t*)eoff, (const smr::u64*)woff, d.root3, d.pg, estr, eid, eblk, derr, (const uint32_t*)d.pos_off);
  uint32_t herr = 0;                                       // (a block k_pgb_collect refused has no entries written: nothing below may run on them)
  HIPCHK(c, hipMemcpyAsync(&herr, derr, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (herr) { set_err(c, (herr & 1u) ? "a mini-trie is too large for the pigeonhole layout" : "pigeonhole arena exceeds 2^34 words"); return SMR_ERR_CAPACITY; }
  int blockbits = 1; while ((1u << blockbits) < nb) blockbits++;
  const uint32_t gE = (uint32_t)(((smr::u64)E + 255) / 256);
  for (int order = 0; order < 2 && E; order++) {
    const uint32_t kbits = order == 0 ? 2 * (pw + 1) : 2 * (pw - pw / 2);
    smr::u64 *ka = k0, *kb = k1; uint32_t *va = v0, *vb = v1;
    if (order == 0) hipLaunchKernelGGL(smr::k_pgb_keys<0>, dim3(gE), dim3(256), 0, c->stream, (const uint32_t*)estr, (const uint32_t*)eblk, (smr::u64)E, pw, kbits, ka, va);
    else hipLaunchKernelGGL(smr::k_pgb_keys<1>, dim3(gE), dim3(256), 0, c->stream, (const uint32_t*)estr, (const uint32_t*)eblk, (smr::u64)E, pw, kbits, ka, va);
    if ((rc = dev_radix_sort(c, pool, ka, kb, va, vb, E, 0, (int)kbits + blockbits))) return rc;
    if (order == 0) hipLaunchKernelGGL(smr::k_pgb_emit<0>, dim3(gE), dim3(256), 0, c->stream, (const smr::u64*)ka, (const uint32_t*)va, (smr::u64)E, pw, kbits,
                                       (const uint32_t*)estr, (const uint32_t*)eid, (const uint32_t*)eoff, (const uint32_t*)d.root3, d.pg);
    else hipLaunchKernelGGL(smr::k_pgb_emit<1>, dim3(gE), dim3(256), 0, c->stream, (const smr::u64*)ka, (const uint32_t*)va, (smr::u64)E, pw, kbits,
                            (const uint32_t*)estr, (const uint32_t*)eid, (const uint32_t*)eoff, (const uint32_t*)d.root3, d.pg);
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipGetLastError());
  d.pg_words = W;
  return SMR_OK;
}
}  // namespace

extern "C" int smr_index_upload(smr_ctx* c, const smr_index* ix, int slot) {
  if (!c || !ix || slot < 0 || slot >= 64) return SMR_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->idx[slot].used) smr_index_unload(c, slot);
  DevIndex& d = c->idx[slot];
  d.lnwin = ix->lnwin; d.n_refs = ix->n_refs(); d.n_ids = ix->n_ids(); d.trie_words = ix->trie.size(); d.n_pos = ix->pos_arr.size() / 2; d.ref_bytes = ix->ref_seq.size();
  int rc;
  smr_build_lkc(*const_cast<smr_index*>(ix));              // (cached in the smr_index under its mutex: concurrent uploads of one host index are safe)
  if ((rc = dev_alloc(c, &d.lkc, ix->lkc.size()))) return rc;
  HIPCHK(c, hipMemcpyAsync(d.lkc, ix->lkc.data(), ix->lkc.size() * 4, hipMemcpyHostToDevice, c->stream));
  if ((rc = dev_alloc(c, &d.lookup, ix->lookup.size()))) return rc;
  if ((rc = dev_alloc(c, &d.trie, ix->trie.size()))) return rc;
  HIPCHK(c, hipMemcpyAsync(d.lookup, ix->lookup.data(), ix->lookup.size() * sizeof(Lookup), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d.trie, ix->trie.data(), ix->trie.size() * 4, hipMemcpyHostToDevice, c->stream));
  // the position lists: pos_off (which the layout build below and the DFS kernel translate ids with) and pos2 = {header, positions} per seed (k_pos2_build)
  if ((uint64_t)ix->pos_arr.size() / 2 + ix->n_ids() >= 0x7FFFFFF0ull) { set_err(c, "index part limit: positions + distinct seeds < 2^31"); return SMR_ERR_CAPACITY; }
  if ((rc = dev_alloc(c, &d.pos_off, ix->pos_off.size()))) return rc;
  HIPCHK(c, hipMemcpyAsync(d.pos_off, ix->pos_off.data(), ix->pos_off.size() * 4, hipMemcpyHostToDevice, c->stream));
  {
    uint2* raw = nullptr;
    if ((rc = dev_alloc(c, &raw, ix->pos_arr.size() / 2))) return rc;
    HIPCHK(c, hipMemcpyAsync(raw, ix->pos_arr.data(), ix->pos_arr.size() * 4, hipMemcpyHostToDevice, c->stream));
    if ((rc = dev_alloc(c, &d.pos_arr, ix->pos_arr.size() / 2 + (size_t)d.n_ids + 1))) { dev_free(&raw); return rc; }
    if (d.n_ids) hipLaunchKernelGGL(k_pos2_build, dim3((d.n_ids + 255u) / 256u), dim3(256), 0, c->stream, (const uint32_t*)d.pos_off, (const uint2*)raw, d.n_ids, d.pos_arr);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    dev_free(&raw);
  }
  // The pigeonhole layout of the tries (what k_seed_pg reads) is built on the device from the arena just uploaded (smr_pgbuild.hpp).  An index
  // that already carries the host-built layout (smr_index_selfcheck, SMR_PG_HOST=1) is uploaded as it is.
  bool host_pg = getenv("SMR_PG_HOST") && atoi(getenv("SMR_PG_HOST"));
  { std::lock_guard<std::mutex> l_(const_cast<smr_index*>(ix)->pg_mutex); host_pg = host_pg || !ix->root3.empty(); }     // (another context may be inside smr_build_pigeonhole on the same host index)
  if (host_pg) {
    std::string why;
    if (!smr_build_pigeonhole(*const_cast<smr_index*>(ix), 0, why)) { set_err(c, why); return SMR_ERR_CAPACITY; }
    if ((rc = dev_alloc(c, &d.pg, ix->pg.size()))) return rc;
    if ((rc = dev_alloc(c, &d.root3, ix->root3.size()))) return rc;
    HIPCHK(c, hipMemcpyAsync(d.pg, ix->pg.data(), ix->pg.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d.root3, ix->root3.data(), ix->root3.size() * 4, hipMemcpyHostToDevice, c->stream));
    d.pg_words = ix->pg.size() >= 4 ? ix->pg.size() - 4 : 0;
  } else if ((rc = build_pigeonhole_device(c, d, (uint32_t)ix->lookup.size(), ix->lnwin / 2))) return rc;
  if (getenv("SMR_VERBOSE")) fprintf(stderr, "libsmr_hip: index part: tries %.2f GB, pigeonhole arena %.2f GB (%s), positions %.2f GB\n", ix->trie.size() * 4e-9, d.pg_words * 4e-9,
                                     host_pg ? "host-built" : "built on the device", ix->pos_arr.size() * 4e-9);
  if ((rc = dev_alloc(c, &d.ref_seq, ix->ref_seq.size() + 64))) return rc;
  if ((rc = dev_alloc(c, &d.ref_off, ix->ref_off.size()))) return rc;
  HIPCHK(c, hipMemcpyAsync(d.ref_seq, ix->ref_seq.data(), ix->ref_seq.size(), hipMemcpyHostToDevice, c->stream));
  d.ref_any_n = (!ix->ref_seq.empty() && memchr(ix->ref_seq.data(), 4, ix->ref_seq.size())) ? 1u : 0u;      // (k_sw16 asks each window for its ambiguous letters only then)
  HIPCHK(c, hipMemcpyAsync(d.ref_off, ix->ref_off.data(), ix->ref_off.size() * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  d.used = true;
  return SMR_OK;
}

// the device-built pigeonhole layout of slot `slot` against the host transform of the same index (smr_build_pigeonhole): word for word
extern "C" int smr_index_check_device(smr_ctx* c, int slot, smr_index* ix) {
  if (!c || !ix || slot < 0 || slot >= 64 || !c->idx[slot].used) return SMR_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const DevIndex& d = c->idx[slot];
  std::string why;
  if (!smr_build_pigeonhole(*ix, 0, why)) { set_err(c, why); return SMR_ERR_CAPACITY; }
  if (ix->pg.size() != d.pg_words + 4) { set_err(c, "pigeonhole layout: the device arena has " + std::to_string(d.pg_words) + " words, the host's " + std::to_string(ix->pg.size() - 4)); return SMR_ERR_STATE; }
  std::vector<uint32_t> r3(ix->root3.size()), pg(ix->pg.size());
  HIPCHK(c, hipMemcpy(r3.data(), d.root3, r3.size() * 4, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(pg.data(), d.pg, pg.size() * 4, hipMemcpyDeviceToHost));
  for (size_t q = 0; q < r3.size(); q++) if (r3[q] != ix->root3[q]) { set_err(c, "pigeonhole layout: block table differs at word " + std::to_string(q)); return SMR_ERR_STATE; }
  for (size_t q = 0; q < pg.size(); q++) if (pg[q] != ix->pg.data()[q]) { set_err(c, "pigeonhole layout: arena differs at word " + std::to_string(q)); return SMR_ERR_STATE; }
  return SMR_OK;
}

extern "C" int smr_index_unload(smr_ctx* c, int slot) {
  if (!c || slot < 0 || slot >= 64) return SMR_ERR_ARG;
  DevIndex& d = c->idx[slot];
  dev_free(&d.lookup); dev_free(&d.trie); dev_free(&d.pg); dev_free(&d.root3); dev_free(&d.lkc); dev_free(&d.pos_off); dev_free(&d.pos_arr); dev_free(&d.ref_seq); dev_free(&d.ref_off);
  d = DevIndex();
  return SMR_OK;
}

extern "C" int smr_batch_select(smr_ctx* c, int batch) {
  if (!c || batch < 0 || batch >= SMR_MAX_BATCHES) return SMR_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  Batch& B = c->bt[batch];
  if (!B.d_ctr) {
    HIPCHK(c, hipMalloc((void**)&B.d_ctr, C_TOTAL * 8));
    HIPCHK(c, hipMemset(B.d_ctr, 0, C_TOTAL * 8));
  }
  B.used = true;
  { std::lock_guard<std::mutex> l(c->sel_m); c->b = &B; }
  return SMR_OK;
}

extern "C" int smr_set_seed_mode(smr_ctx* c, int exact_counters) {
  if (!c) return SMR_ERR_ARG;
  c->seed_exact = exact_counters ? 1 : 0;
  return SMR_OK;
}

namespace {
int reset_batch(smr_ctx* c, Batch& B, hipStream_t st) {
  B.gen++;
  HIPCHK(c, hipMemsetAsync(B.d_saved, 0, (size_t)B.n * sizeof(RState), st));
  HIPCHK(c, hipMemsetAsync(B.d_saved_aln, 0, (size_t)B.n * B.slots * sizeof(AlignRec), st));
  // the Readstats counters, error flags and cursors start over; the profiling work counters (windows .. SW cells, scored-ahead counts and
  // their shards) keep accumulating until smr_prof_reset, like the kernel times do
  HIPCHK(c, hipMemsetAsync(B.d_ctr, 0, (size_t)C_WINDOWS * 8, st));
  HIPCHK(c, hipMemsetAsync(B.d_ctr + C_ERR_HITCAP, 0, (size_t)(C_SW_SPEC - C_ERR_HITCAP) * 8, st));
  HIPCHK(c, hipMemsetAsync(B.d_ctr + C_PCUR, 0, (size_t)C_NSHARD * C_PCUR_STRIDE * 8, st));
  HIPCHK(c, hipStreamSynchronize(st));
  B.fetched = false;
  return SMR_OK;
}
template <class T> int grow(smr_ctx* c, T** p, size_t& cap, size_t need) {        // grow-only device array
  if (*p && cap >= need) return SMR_OK;
  int rc = dev_alloc(c, p, need); if (rc) return rc;
  cap = need;
  return SMR_OK;
}
// the packed reads of r into batch B (+ its per-read state, reset), all work on stream st; touches nothing but B
int upload_into(smr_ctx* c, Batch& B, const smr_reads* r, uint32_t max_aln, hipStream_t st) {
  if (max_aln == 0) max_aln = 1;
  int rc;
  if (!B.d_ctr) { HIPCHK(c, hipMalloc((void**)&B.d_ctr, C_TOTAL * 8)); HIPCHK(c, hipMemsetAsync(B.d_ctr, 0, C_TOTAL * 8, st)); }
  const size_t nw = r->words.size() + 4, nr = (size_t)r->n + 1, na = std::max<size_t>((size_t)r->n * max_aln, 1);     // + slack: window extraction reads 2 words ahead
  if (B.cap_words < nw) { if ((rc = dev_alloc(c, &B.d_words, nw))) return rc; B.cap_words = nw; }
  if (B.cap_reads < nr) {
    if ((rc = dev_alloc(c, &B.d_rec_off, nr))) return rc;
    if ((rc = dev_alloc(c, &B.d_len, nr))) return rc;
    if ((rc = dev_alloc(c, &B.d_saved, nr))) return rc;
    if ((rc = dev_alloc(c, &B.d_work, nr))) return rc;
    if ((rc = dev_alloc(c, &B.d_rw, nr))) return rc;
    if ((rc = dev_alloc(c, &B.d_marks, nr))) return rc;
    B.cap_reads = nr;
  }
  if (B.cap_aln < na) {
    if ((rc = dev_alloc(c, &B.d_saved_aln, na))) return rc;
    if ((rc = dev_alloc(c, &B.d_work_aln, na))) return rc;
    B.cap_aln = na;
  }
  B.n = r->n; B.max_len = r->max_len; B.slots = max_aln; B.used = true;
  for (int k = 0; k < 7; k++) B.min_ge[k] = ~0u;
  B.min_ge_known = false;
  if (r->n && r->min_len >= 100) { for (int k = 0; k < 7; k++) B.min_ge[k] = r->min_len; B.min_ge_known = true; }      // (1 % of 100 letters is a margin already: nothing to look for)
  HIPCHK(c, hipMemcpyAsync(B.d_words, r->words.data(), r->words.size() * 4, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(B.d_rec_off, r->rec_off.data(), r->rec_off.size() * 8, hipMemcpyHostToDevice, st));
  if (r->n) HIPCHK(c, hipMemcpyAsync(B.d_len, r->len.data(), r->len.size() * 4, hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemsetAsync(B.d_rw, 0, (size_t)B.n * sizeof(RWork), st));
  HIPCHK(c, hipMemsetAsync(B.d_work, 0, (size_t)B.n * sizeof(RState), st));
  B.cigar_words = 0; dev_free(&B.d_cigar);
  return reset_batch(c, B, st);
}
}  // namespace

extern "C" const char* smr_params_refused(const smr_params* p) {
  if (!p) return "null params";
  if (p->edges < 1 || p->edges > 10) return "edges must be 1..10 (nucleotides or percent), like the reference's --edges";
  return nullptr;
}

extern "C" int smr_state_reset(smr_ctx* c) {
  if (!c || !c->b->d_saved) return SMR_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  return reset_batch(c, *c->b, c->stream);
}

extern "C" int smr_reads_upload(smr_ctx* c, const smr_reads* r, uint32_t max_aln) {
  if (!c || !r) return SMR_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  return upload_into(c, *c->b, r, max_aln, c->stream);
}

// The same into batch `batch`, without selecting it, on the context's UPLOAD stream: may be called from a second host thread while the
// first one is inside smr_align_part / smr_traceback / smr_results_fetch of ANOTHER batch (the two calls share no device buffer: each
// batch owns its reads, per-read state, counters and CIGAR pool; the scratch of the kernels is sized inside smr_align_part).
extern "C" int smr_reads_upload_batch(smr_ctx* c, int batch, const smr_reads* r, uint32_t max_aln) {
  if (!c || !r || batch < 0 || batch >= SMR_MAX_BATCHES) return SMR_ERR_ARG;
  {
    std::lock_guard<std::mutex> l(c->sel_m);
    if (&c->bt[batch] == c->b) { std::lock_guard<std::mutex> l2(c->err_m); c->err = "smr_reads_upload_batch: the batch is the selected one (use smr_reads_upload)"; return SMR_ERR_STATE; }
  }
  HIPCHK(c, hipSetDevice(c->device));
  return upload_into(c, c->bt[batch], r, max_aln, c->upload_stream);
}

namespace {
int ensure_pool(smr_ctx* c) {          // seed-hit pool: scratch shared by all batches, sized for the selected one before its kernels run
  const uint64_t want_pool = std::max<uint64_t>((uint64_t)c->b->n * 64 + (1u << 20), 1u << 22);
  if (c->pool_words < want_pool) { int rc = dev_alloc(c, &c->d_pool, want_pool); if (rc) return rc; c->pool_words = want_pool; }
  return SMR_OK;
}
}  // namespace

__global__ void k_ctr_begin(unsigned long long* __restrict__ ctr, const unsigned long long* __restrict__ snap) {
  for (int k = threadIdx.x; k < C_TOTAL; k += blockDim.x) {
    const bool zero = k == C_NUM_SHORT || (k >= C_ERR_HITCAP && k <= C_ERR_TRACE) || k == C_ERR_SCAP || k == C_ERR_REDO || k == C_POOL_CURSOR || k == C_WORK_NEXT || k >= C_PCUR;
    ctr[k] = zero ? 0ull : snap[k];
  }
}

extern "C" int smr_align_part(smr_ctx* c, int slot, const smr_params* p) {
  if (!c || slot < 0 || slot >= 64) return SMR_ERR_ARG;
  if (!c->idx[slot].used || !c->b->d_saved) { set_err(c, "index slot empty or no reads uploaded"); return SMR_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  int rc = check_params(c, p); if (rc) return rc;
  const DevIndex& di = c->idx[slot];
  if (di.lnwin < 8 || di.lnwin > 20) { set_err(c, "unsupported seed length"); return SMR_ERR_ARG; }
  DParams P = make_dparams(c, di, p);
  // --edges N% of a read of fewer than 100 / N letters is 0, and 0 is not "no margin" in the reference: `tail > edges - 1` is an unsigned compare
  // (alignment.cpp:320,345), so such a read is aligned against the whole rest of its reference sequence.  Not built here (the windows of the
  // Smith-Waterman kernels are sized read + 2 x edges): said before anything runs, not as a capacity error of some kernel.
  if (p->is_as_percent) {
    if (!c->b->min_ge_known) {                                // the shortest searchable read of the batch, per seed length: from the lengths on the device, once per batch
      std::vector<uint32_t> len(c->b->n);
      if (c->b->n) { HIPCHK(c, hipMemcpyAsync(len.data(), c->b->d_len, (size_t)c->b->n * 4, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
      for (uint32_t l : len) for (int k = 0; k < 7; k++) if (l >= 8u + 2u * (uint32_t)k && l < c->b->min_ge[k]) c->b->min_ge[k] = l;
      c->b->min_ge_known = true;
    }
    const uint32_t lmin = c->b->min_ge[(std::min<uint32_t>(std::max<uint32_t>(di.lnwin, 8u), 20u) - 8u) / 2u];
    if (lmin != ~0u && (uint32_t)((p->edges / 100.0) * (double)lmin) == 0) {
      set_err(c, "edges as a percentage: the batch has a searchable read so short that the percentage rounds to 0 letters; the reference then aligns it against the whole rest of the reference sequence (alignment.cpp:320,345), which is not supported -- use an absolute --edges");
      return SMR_ERR_ARG;
    }
  }
  uint32_t ml, rf; size_t lds; chain_lds(c, P, ml, rf, lds);
  if (lds > 150 * 1024) { set_err(c, "reads too long for this build of the SW kernel (LDS)"); return SMR_ERR_CAPACITY; }
  c->b->last_num_alignments = p->num_alignments;
  c->b->fetched = false;
  ev_drop(c);
  if (c->b->n == 0) return SMR_OK;
  if ((rc = ensure_pool(c))) return rc;
  if (P.sw_mode < 0) {
    // the striped slow path: per block of k_chain / k_begins five arrays of 16 x ceil(len / 8) uint16 (smr_sw_striped.hpp)
    if (c->chain_blocks == 0) c->chain_blocks = (uint32_t)c->n_cu * (getenv("SMR_CHAIN_WPC") ? atoi(getenv("SMR_CHAIN_WPC")) : 12);
    const uint32_t stride = 5u * 16u * ((c->b->max_len + 7u) / 8u + 1u);
    const size_t need = (size_t)std::max<uint32_t>(c->chain_blocks, (uint32_t)c->n_cu * 8u) * stride;
    if (c->sw_scr_words < need || c->sw_scr_stride < stride) { if ((rc = dev_alloc(c, &c->d_sw_scr, need))) return rc; c->sw_scr_words = need; c->sw_scr_stride = stride; }
    P.sw_scratch = c->d_sw_scr; P.sw_scratch_stride = c->sw_scr_stride;
    if (getenv("SMR_VERBOSE")) fprintf(stderr, "libsmr_hip: scoring scheme %d/%d/%d (N %d): %s -- Smith-Waterman through the slow path that reproduces ssw.c's stripe geometry\n",
                                       p->match, p->mismatch, p->gap_open, p->score_N, scheme_unsupported(p->mismatch, p->score_N, p->gap_open, p->gap_ext));
  }
  std::vector<unsigned long long> h;
  // the counters as they stand now stay on the device; an attempt that has to be redone starts from them again (one read-back per attempt, none before)
  HIPCHK(c, hipMemcpyAsync(c->d_ctr_snap, c->b->d_ctr, C_TOTAL * 8, hipMemcpyDeviceToDevice, c->stream));
  const uint32_t tb = 256, nb = (c->b->n + tb - 1) / tb;
  const int single = (p->is_forward != 0) ^ (p->is_reverse != 0);
  const int num_strands = single ? 1 : 2;
  for (int attempt = 0; attempt < 16; attempt++) {            // (the hit-list ladder of the DFS kernel alone has seven steps: grow_hcap)
    if ((rc = ensure_chain_scratch(c, di))) return rc;
    const KpSave kp0 = kp_save(c);
    c->wstat_n = 0;
    // restore counters (retry) and clear the per-part ones (processor.cpp:230 resets num_short per part)
    hipLaunchKernelGGL(k_ctr_begin, dim3(1), dim3(256), 0, c->stream, c->b->d_ctr, (const unsigned long long*)c->d_ctr_snap);
    hipLaunchKernelGGL(k_begin_part, dim3(nb), dim3(tb), 0, c->stream, dreads(c), P, c->b->d_saved, c->b->d_saved_aln, c->b->d_work, c->b->d_work_aln, c->b->d_rw, c->b->d_ctr);
    for (int count = 0; count < num_strands; count++) {
      hipLaunchKernelGGL(k_begin_strand, dim3(nb), dim3(tb), 0, c->stream, c->b->n, P, count, c->b->d_work, c->b->d_rw);
      HIPCHK(c, hipMemsetAsync(&c->b->d_ctr[C_PCUR], 0, C_NSHARD * C_PCUR_STRIDE * 8, c->stream));
      for (int pass = 0; pass < 3; pass++) {
        if (pass > 0 && P.skip[pass] == P.skip[pass - 1]) continue;     // equal strides are skipped (paralleltraversal.cpp:269-272)
        if ((rc = launch_seed(c, di, P, pass, !p->is_last_index_part, ((single && p->is_reverse) || count == 1) ? 1 : 0))) return rc;
        if ((rc = launch_chain(c, di, P, pass, single || count == 1))) return rc;
      }
    }
    HIPCHK(c, hipGetLastError());
    if ((rc = read_ctr(c, h))) return rc;
    ev_collect(c);
    bool retry = false;
    if (h[C_ERR_HITCAP]) { if (!grow_hcap(c, P.partialwin)) return SMR_ERR_CAPACITY; retry = true; }
    if (h[C_ERR_POOL]) { uint64_t w = c->pool_words * 2; if (w > 0x7FFFFFF0ull) { set_err(c, "seed-hit pool exceeds 8 GiB"); return SMR_ERR_CAPACITY; }
      if ((rc = dev_alloc(c, &c->d_pool, w))) return rc; c->pool_words = w; retry = true; }
    if (h[C_ERR_PAIRS]) {
      c->pairs_cap *= 4; c->hits_cap *= 4; dev_free(&c->d_pairs); dev_free(&c->d_lis); dev_free(&c->d_hits); dev_free(&c->d_tuples); dev_free(&c->d_tuples2);
      if (c->pairs_cap > (1u << 22)) { set_err(c, "per-read candidate scratch exceeds capacity"); return SMR_ERR_CAPACITY; }
      retry = true;
    }
    if (h[C_ERR_REDO]) { c->seed_exact = 1; retry = true; }     // too many overflowing waves for the redo list: use the DFS kernel throughout
    {
      // k_seed_pg's candidate pool: when more than 1/64 of this part's waves overflowed it (they were searched again by the DFS kernel:
      // right, but slow), the next launches get twice the pool
      if (h[C_SEED_REDO] < c->b->redo_seen || h[C_WINDOWS] < c->b->win_seen) c->b->redo_seen = c->b->win_seen = 0;       // counters were reset
      const unsigned long long redo = h[C_SEED_REDO] - c->b->redo_seen, waves = (h[C_WINDOWS] - c->b->win_seen) / 32;   // forward + reverse search per window
      c->b->redo_seen = h[C_SEED_REDO]; c->b->win_seen = h[C_WINDOWS];
      if (!retry && redo * 64 > waves && c->ccap < PG_CAND_CAP_MAX) {
        c->ccap *= 2;
        if (getenv("SMR_VERBOSE")) fprintf(stderr, "libsmr_hip: %llu of ~%llu seed-search waves overflowed their candidate pool: %u records per wave from now on\n", redo, waves, c->ccap);
      }
    }
    if (h[C_ERR_SCAP]) {
      // a read shares seeds with more references than the LDS table of its wave holds (384): from now on such reads build their set in a
      // per-block table in global memory; the candidate keys need room for as many members
      if (c->chain_ext) { set_err(c, "more than 49152 references share seeds with one read (candidate set capacity)"); return SMR_ERR_CAPACITY; }
      c->chain_ext = true; retry = true;
      if (getenv("SMR_VERBOSE")) fprintf(stderr, "libsmr_hip: a read shares seeds with more references than its wave's LDS table holds: per-block global candidate tables enabled (%.1f GB)\n",
                                         (double)c->chain_blocks * (4.0 * CH_EXT_CAP * 4 + (double)c->pairs_cap * 8 + (double)CH_EXT_CAP * 8) / 1e9);
      if (c->keys_cap < CH_EXT_CAP) { dev_free(&c->d_keys); c->keys_cap = 0; c->keys_need = CH_EXT_CAP; }
    }
    if (h[C_ERR_SLOTS]) { set_err(c, "a read produced more alignments than max_alignments_per_read (smr_reads_upload)"); return SMR_ERR_CAPACITY; }
    if (retry) kp_restore(c, kp0);   // timings of a discarded attempt
    if (!retry) {
      if ((rc = adapt_walk_rounds(c))) return rc;
      // the begin cells of the alignments that are still stored (k_chain records the accepted ones "begin pending"): four reverse passes per wave
      {
        const uint64_t ntot = (uint64_t)c->b->n * c->b->slots;
        if (c->tasks_cap < ntot) { if ((rc = dev_alloc(c, &c->d_tasks, 2 * ntot))) return rc; c->tasks_cap = ntot; }
        uint32_t ml, rf; size_t chain_bytes;
        chain_lds(c, P, ml, rf, chain_bytes);
        const int x4 = (P.sw_mode >= 1 && c->b->max_len <= SW_X4_MAX_ROWS &&
                        (long long)c->b->max_len * P.match + 255 < 32768 && rf + 128 <= 8191 && P.gap_open + P.mismatch >= 0 && P.gap_open + P.score_N >= 0 &&
                        P.match + P.gap_open <= 255 && P.score_N + P.gap_open <= 255) ? 1 : 0;
        const size_t lds_b = x4 ? (size_t)4 * (ml + rf) : (c->b->max_len > SW_X4_MAX_ROWS ? 0 : (size_t)ml + rf);
        const uint32_t bg_blocks = (uint32_t)c->n_cu * 8u;
        if ((rc = ensure_bound(c, std::max(bg_blocks, c->chain_blocks), rf))) return rc;
        if (lds_b > 64 * 1024 && lds_b > c->begins_lds_attr) {
          HIPCHK(c, hipFuncSetAttribute((const void*)k_begins<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
          HIPCHK(c, hipFuncSetAttribute((const void*)k_begins<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
          HIPCHK(c, hipFuncSetAttribute((const void*)k_begins<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
          HIPCHK(c, hipFuncSetAttribute((const void*)k_begins<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
          c->begins_lds_attr = lds_b;
        }
        HIPCHK(c, hipMemsetAsync(&c->b->d_ctr[C_BEGIN_N], 0, 16, c->stream));       // C_BEGIN_N, C_BEGIN_NEXT
        ev_mark(c, KP_BEGINS);
        hipLaunchKernelGGL(k_begins_collect, dim3((uint32_t)((ntot + 1023) / 1024)), dim3(1024), 0, c->stream, c->b->n, c->b->slots, (const RState*)c->b->d_work, (const RWork*)c->b->d_rw,
                           (const AlignRec*)c->b->d_work_aln, c->d_tasks, c->b->d_ctr);
        const size_t task_cap = (size_t)c->walk_cap * c->walk_kcap;
        if (x4 && c->walk_split && c->d_wtask[0] && c->d_wctr && ntot <= task_cap && !getenv("SMR_BEGINS_X4")) {
          // sixteen per wave through k_sw16 (smr_walk.hpp): stage 0 = the end cells of the alignments stored end-pending, stage 1 = the begin cells of all
          const uint32_t wmq = std::min<uint32_t>(c->b->max_len, WK_MAX_ROWS);
          const int swr = wmq <= 104 ? 13 : wmq <= 152 ? 19 : wmq <= 208 ? 26 : 32;
          const uint32_t sw_blocks = (uint32_t)c->n_cu * 4u * (uint32_t)SW16_WAVES(swr);
          for (int stage = 0; stage < 2; stage++) {
            HIPCHK(c, hipMemsetAsync(c->d_wctr, 0, (size_t)WC_STRIDE * 8, c->stream));
            hipLaunchKernelGGL(k_begins_prep, dim3((uint32_t)c->n_cu * 2u), dim3(1024), 0, c->stream, dindex(di), c->b->slots, (const uint32_t*)c->d_tasks, (const unsigned long long*)&c->b->d_ctr[C_BEGIN_N],
                               (const AlignRec*)c->b->d_work_aln, stage, c->d_wtask[0], c->d_wtidx, c->d_wctr);
#define SW16_ARGS dreads(c), dindex(di), P, (const WTask*)c->d_wtask[0], (const uint32_t*)c->d_wtidx, (const uint32_t*)(c->d_wtidx + task_cap), (const unsigned long long*)c->d_wctr, c->d_wres[0]
            if (swr == 13) hipLaunchKernelGGL(k_sw16<13>, dim3(sw_blocks), dim3(64), 0, c->stream, SW16_ARGS);
            else if (swr == 19) hipLaunchKernelGGL(k_sw16<19>, dim3(sw_blocks), dim3(64), 0, c->stream, SW16_ARGS);
            else if (swr == 26) hipLaunchKernelGGL(k_sw16<26>, dim3(sw_blocks), dim3(64), 0, c->stream, SW16_ARGS);
            else hipLaunchKernelGGL(k_sw16<32>, dim3(sw_blocks), dim3(64), 0, c->stream, SW16_ARGS);
#undef SW16_ARGS
            hipLaunchKernelGGL(k_begins_apply, dim3((uint32_t)c->n_cu * 4u), dim3(256), 0, c->stream, (const uint32_t*)c->d_tasks, (const unsigned long long*)&c->b->d_ctr[C_BEGIN_N], c->b->d_work_aln, stage,
                               (const WTask*)c->d_wtask[0], (const uint2*)c->d_wres[0], c->b->d_ctr);
          }
        } else if (P.sw_mode < 0) {
          if (c->b->max_len > SW_X4_MAX_ROWS) hipLaunchKernelGGL((k_begins<true, true>), dim3(bg_blocks), dim3(64), lds_b, c->stream, dreads(c), dindex(di), P, (const uint32_t*)c->d_tasks, c->b->d_work_aln, c->b->d_ctr, ml, rf, x4, c->d_bound, c->d_rdq);
          else hipLaunchKernelGGL((k_begins<false, true>), dim3(bg_blocks), dim3(64), lds_b, c->stream, dreads(c), dindex(di), P, (const uint32_t*)c->d_tasks, c->b->d_work_aln, c->b->d_ctr, ml, rf, x4, (int*)nullptr, (uint8_t*)nullptr);
        } else if (c->b->max_len > SW_X4_MAX_ROWS)
          hipLaunchKernelGGL(k_begins<true>, dim3(bg_blocks), dim3(64), lds_b, c->stream, dreads(c), dindex(di), P, (const uint32_t*)c->d_tasks, c->b->d_work_aln, c->b->d_ctr, ml, rf, x4, c->d_bound, c->d_rdq);
        else
          hipLaunchKernelGGL(k_begins<false>, dim3(bg_blocks), dim3(64), lds_b, c->stream, dreads(c), dindex(di), P, (const uint32_t*)c->d_tasks, c->b->d_work_aln, c->b->d_ctr, ml, rf, x4, (int*)nullptr, (uint8_t*)nullptr);
        ev_stop(c);
      }
      // only a clean attempt is committed to the persistent per-read state (kvdb.put, processor.cpp:150-155)
      hipLaunchKernelGGL(k_commit_part, dim3(nb), dim3(tb), 0, c->stream, c->b->n, P, c->b->d_saved, c->b->d_saved_aln, c->b->d_work, c->b->d_work_aln, c->b->d_rw);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipStreamSynchronize(c->stream));
      ev_collect(c);
      return SMR_OK;
    }
  }
  set_err(c, "capacity retries exhausted");
  return SMR_ERR_CAPACITY;
}

#include "smr_engine_trace.hpp"
extern "C" int smr_counters(smr_ctx* c, uint64_t* out, uint32_t n_db) {
  if (!c || !out) return SMR_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<unsigned long long> h;
  int rc = read_ctr(c, h); if (rc) return rc;
  out[0] = h[C_NUM_ALIGNED]; out[1] = h[C_NUM_SHORT];
  for (uint32_t i = 0; i < n_db && i < 64; i++) out[2 + i] = h[C_PER_DB + i];
  return SMR_OK;
}
extern "C" int smr_counters_device(smr_ctx* c, void** dptr, uint32_t* n_u64) {
  if (!c || !dptr || !n_u64) return SMR_ERR_ARG;
  *dptr = c->b->d_ctr; *n_u64 = C_PER_DB + 64;
  return SMR_OK;
}

__global__ void k_ctr_accumulate(const unsigned long long* __restrict__ ctr, unsigned long long* __restrict__ acc, uint32_t n) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) acc[k] += ctr[k];
}
// acc[k] += the selected batch's counter k, k < n_u64 <= the block smr_counters_device hands out, on the device (a host that aligns its
// shard in chunks keeps ONE device block of sums and all-reduces that over the ranks)
extern "C" int smr_counters_accumulate(smr_ctx* c, void* d_acc, uint32_t n_u64) {
  if (!c || !d_acc || n_u64 > C_PER_DB + 64 || !c->b->d_ctr) return SMR_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(k_ctr_accumulate, dim3((n_u64 + 127) / 128), dim3(128), 0, c->stream, (const unsigned long long*)c->b->d_ctr, (unsigned long long*)d_acc, n_u64);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SMR_OK;
}

// the reads that have alignments, packed: read number, state, alignment slots (order = whatever the atomics hand out)
__global__ void k_results_compact(uint32_t n, uint32_t slots, const RState* __restrict__ saved, const AlignRec* __restrict__ saved_aln,
                                  uint32_t* __restrict__ out_idx, RState* __restrict__ out_state, AlignRec* __restrict__ out_aln, unsigned long long* __restrict__ ctr) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  RState s; s.n_align = 0;
  if (r < n) s = saved[r];
  const uint32_t p = block_append(&ctr[C_FETCH_N], s.n_align != 0);
  if (s.n_align == 0) return;
  out_idx[p] = r; out_state[p] = s;
  for (uint32_t k = 0; k < s.n_align && k < slots; k++) out_aln[(size_t)p * slots + k] = saved_aln[(size_t)r * slots + k];
}

// Only the reads that have alignments cross the bus (a tenth of the batch on the bench workload): compacted on the device, then
// read number -> packed position on the host.
extern "C" int smr_results_fetch(smr_ctx* c) {
  if (!c || !c->b->d_saved) return SMR_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  Batch& B = *c->b;
  int rc;
  const size_t need_r = std::max<size_t>(B.n, 1), need_a = std::max<size_t>((size_t)B.n * B.slots, 1);
  if (c->fetch_cap_r < need_r) {
    if ((rc = dev_alloc(c, &c->d_fidx, need_r))) return rc;
    if ((rc = dev_alloc(c, &c->d_fstate, need_r))) return rc;
    c->fetch_cap_r = need_r;
  }
  if (c->fetch_cap_a < need_a) { if ((rc = dev_alloc(c, &c->d_faln, need_a))) return rc; c->fetch_cap_a = need_a; }
  HIPCHK(c, hipMemsetAsync(&B.d_ctr[C_FETCH_N], 0, 8, c->stream));
  if (B.n) hipLaunchKernelGGL(k_results_compact, dim3((B.n + 1023) / 1024), dim3(1024), 0, c->stream, B.n, B.slots, (const RState*)B.d_saved, (const AlignRec*)B.d_saved_aln,
                              c->d_fidx, c->d_fstate, c->d_faln, B.d_ctr);
  std::vector<unsigned long long> h;
  rc = read_ctr(c, h); if (rc) return rc;
  const uint64_t cw = std::min<uint64_t>(h[C_CIGAR_CURSOR], B.cigar_words);
  const size_t nhit = (size_t)std::min<unsigned long long>(h[C_FETCH_N], B.n);
  B.h_cigar.resize(cw);
  B.h_idx.resize(nhit); B.h_state.resize(nhit); B.h_aln.resize(nhit * B.slots);
  if (nhit) {
    HIPCHK(c, hipMemcpyAsync(B.h_idx.data(), c->d_fidx, nhit * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(B.h_state.data(), c->d_fstate, nhit * sizeof(RState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(B.h_aln.data(), c->d_faln, nhit * B.slots * sizeof(AlignRec), hipMemcpyDeviceToHost, c->stream));
  }
  if (cw) HIPCHK(c, hipMemcpyAsync(B.h_cigar.data(), B.d_cigar, cw * 4, hipMemcpyDeviceToHost, c->stream));
  B.h_map.assign(B.n, 0xFFFFFFFFu);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (size_t j = 0; j < nhit; j++) B.h_map[B.h_idx[j]] = (uint32_t)j;
  B.fetched = true;
  return SMR_OK;
}

// Read::toBinString read.cpp:429-462 (+ alignment_struct2::toString, s_align2::toString ssw.hpp:106-140)
static size_t record_of(const Batch& B, uint32_t i, uint8_t* buf, size_t cap);
extern "C" size_t smr_result_record(const smr_ctx* c, uint32_t i, uint8_t* buf, size_t cap) { return c ? record_of(*c->b, i, buf, cap) : 0; }
// the same for batch `batch` whichever batch is selected: reads only the host copy smr_results_fetch made of THAT batch, so a second host
// thread can serialise the records of batch k while the first one aligns batch k+1
extern "C" size_t smr_result_record_batch(const smr_ctx* c, int batch, uint32_t i, uint8_t* buf, size_t cap) {
  return (c && batch >= 0 && batch < SMR_MAX_BATCHES) ? record_of(c->bt[batch], i, buf, cap) : 0;
}
static size_t record_of(const Batch& B, uint32_t i, uint8_t* buf, size_t cap) {
  const Batch* const b_ = &B;
  struct { const Batch* b; } cc{b_}; const auto* c = &cc;          // (the body below reads c->b)
  if (!c->b->fetched || i >= c->b->n) return 0;
  const uint32_t j = c->b->h_map[i];
  if (j == 0xFFFFFFFFu) return 0;
  const RState& s = c->b->h_state[j];
  if (s.n_align == 0) return 0;
  const AlignRec* al = c->b->h_aln.data() + (size_t)j * c->b->slots;
  size_t need = 24 + 3 + 2 + 4 + 4 + 8 + 4 + 4 + 8;
  for (uint32_t k = 0; k < s.n_align; k++) need += 8 + 8 + (size_t)(al[k].has_cigar ? al[k].cigar_len : 0) * 4 + 24 + 6 + 1;
  if (!buf || cap < need) return need;
  uint8_t* p = buf;
  auto put = [&](const void* v, size_t n) { memcpy(p, v, n); p += n; };
  uint32_t z32 = 0; uint8_t z8 = 0;
  put(&s.lastIndex, 4); put(&s.lastPart, 4); put(&z32, 4); put(&z32, 4); put(&z32, 4); put(&z32, 4);
  put(&s.is_done, 1); put(&s.is_hit, 1); put(&z8, 1);
  put(&s.max_SW_count, 2);
  int32_t na = (int32_t)c->b->last_num_alignments; put(&na, 4);      // Read::init: num_alignments = opts.num_alignments (if > 0)
  put(&s.hit_seeds, 4);
  uint64_t asz = 16;
  for (uint32_t k = 0; k < s.n_align; k++) asz += 8 + 8 + (uint64_t)(al[k].has_cigar ? al[k].cigar_len : 0) * 4 + 24 + 6 + 1;
  put(&asz, 8); put(&s.min_index, 4); put(&s.max_index, 4);
  uint64_t nal = s.n_align; put(&nal, 8);
  for (uint32_t k = 0; k < s.n_align; k++) {
    const AlignRec& a = al[k];
    uint64_t cl = a.has_cigar ? a.cigar_len : 0;
    uint64_t rl = 8 + cl * 4 + 24 + 6 + 1; put(&rl, 8); put(&cl, 8);
    if (cl) put(c->b->h_cigar.data() + a.cigar_off, cl * 4);
    put(&a.ref_num, 4); put(&a.ref_begin1, 4); put(&a.ref_end1, 4); put(&a.read_begin1, 4); put(&a.read_end1, 4); put(&a.readlen, 4);
    put(&a.score1, 2); put(&a.part, 2); put(&a.index_num, 2); put(&a.strand, 1);
  }
  return (size_t)(p - buf);
}
extern "C" int smr_result_is_hit(const smr_ctx* c, uint32_t i) {
  if (!c || !c->b->fetched || i >= c->b->n || c->b->h_map[i] == 0xFFFFFFFFu) return 0;
  return c->b->h_state[c->b->h_map[i]].is_hit;
}

// ---- standalone seed scan (kernel-level parity and the seed-scan roofline bench) ------------------
__global__ void k_force_pass(uint32_t n, DParams P, int pass, const uint32_t* __restrict__ len, RWork* __restrict__ rw) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  RWork w = rw[i];
  w.strand_active = len[i] >= P.lnwin ? 1 : 0; w.search = 1; w.pass_n = (uint8_t)pass; w.win_shift = P.skip[pass];
  for (int q = 0; q < 3; q++) w.blk_cnt[q] = 0;
  w.hit_total = 0; w.valid = w.strand_active;
  rw[i] = w;
}

extern "C" int smr_seed_scan(smr_ctx* c, int slot, const smr_params* p, int strand, int pass, uint64_t* n_hits_out) {
  if (!c || slot < 0 || slot >= 64 || pass < 0 || pass > 2) return SMR_ERR_ARG;
  if (!c->idx[slot].used || !c->b->d_saved) { set_err(c, "index slot empty or no reads uploaded"); return SMR_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  int rc = check_params(c, p); if (rc) return rc;
  const DevIndex& di = c->idx[slot];
  DParams P = make_dparams(c, di, p);
  c->last_seed_slot = slot;
  if ((rc = ensure_pool(c))) return rc;
  ev_drop(c);
  const uint32_t tb = 256, nb = (c->b->n + tb - 1) / tb;
  std::vector<unsigned long long> h;
  for (int attempt = 0; attempt < 16; attempt++) {
    HIPCHK(c, hipMemsetAsync(&c->b->d_ctr[C_ERR_HITCAP], 0, 16, c->stream));          // HITCAP, POOL
    HIPCHK(c, hipMemsetAsync(&c->b->d_ctr[C_PCUR], 0, C_NSHARD * C_PCUR_STRIDE * 8, c->stream));
    HIPCHK(c, hipMemsetAsync(&c->b->d_ctr[C_HIT], 0, 8, c->stream));
    HIPCHK(c, hipMemsetAsync(&c->b->d_ctr[C_SHARDS], 0, C_SHARD_W * C_NSHARD * 8, c->stream));
    // fresh per-part/strand state: forward, or reverse-complement with ambiguous letters complemented (aval 0 -> 3)
    hipLaunchKernelGGL(k_begin_part, dim3(nb), dim3(tb), 0, c->stream, dreads(c), P, c->b->d_saved, c->b->d_saved_aln, c->b->d_work, c->b->d_work_aln, c->b->d_rw, c->b->d_ctr);
    DParams Q = P; Q.is_forward = 1; Q.is_reverse = 1;
    hipLaunchKernelGGL(k_begin_strand, dim3(nb), dim3(tb), 0, c->stream, c->b->n, Q, strand ? 1 : 0, c->b->d_work, c->b->d_rw);
    hipLaunchKernelGGL(k_force_pass, dim3(nb), dim3(tb), 0, c->stream, c->b->n, P, pass, c->b->d_len, c->b->d_rw);
    if ((rc = launch_seed(c, di, P, pass))) return rc;
    if ((rc = read_ctr(c, h))) return rc;
    ev_collect(c);
    bool retry = false;
    if (h[C_ERR_HITCAP]) { if (!grow_hcap(c, P.partialwin)) return SMR_ERR_CAPACITY; retry = true; }
    if (h[C_ERR_POOL]) { uint64_t w = c->pool_words * 2; if ((rc = dev_alloc(c, &c->d_pool, w))) return rc; c->pool_words = w; retry = true; }
    if (!retry) { if (n_hits_out) *n_hits_out = h[C_HIT]; return SMR_OK; }
  }
  set_err(c, "capacity retries exhausted");
  return SMR_ERR_CAPACITY;
}

// Test seam (the roofline numerator's audit, tests/test_gpu_parity.py): the sorted tuples of the LAST seed-stage launch of this context -- what
// k_seed_pg searched -- and what it takes to decode them: meta = {tuples, forward tuples, coarse bins nc, fine bits fb, char bits cb, nkh,
// candidate records per wave}, cbase[nc + 1].
extern "C" int smr_seed_tuples_fetch(smr_ctx* c, uint64_t* tuples, uint64_t cap_tuples, uint32_t* cbase, uint32_t cap_cbase, uint32_t meta[8]) {
  if (!c || !meta) return SMR_ERR_ARG;
  if (!c->sb.srt || !c->sb.sn) { set_err(c, "no seed stage has run"); return SMR_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t sn[SN_COUNT];
  HIPCHK(c, hipMemcpy(sn, c->sb.sn, sizeof sn, hipMemcpyDeviceToHost));
  const uint32_t nt = (uint32_t)std::min<uint64_t>(sn[SN_TUPLES], 2 * c->sb_slots);      // (cap_tuples is a per-launch field of launch_seed's copy; the arrays hold 2 x sb_slots)
  meta[0] = nt; meta[1] = std::min(sn[SN_FWD], nt); meta[2] = c->sb.nc; meta[3] = c->sb.fb; meta[4] = c->sb.cb; meta[5] = c->sb.nkh; meta[6] = c->ccap; meta[7] = sn[SN_REDO];
  if (tuples) { if (cap_tuples < nt) return SMR_ERR_CAPACITY; if (nt) HIPCHK(c, hipMemcpy(tuples, c->sb.srt, (size_t)nt * sizeof(SeedTup), hipMemcpyDeviceToHost)); }
  if (cbase) { if (cap_cbase < c->sb.nc + 1u) return SMR_ERR_CAPACITY; HIPCHK(c, hipMemcpy(cbase, c->sb.cbase, (size_t)(c->sb.nc + 1u) * 4, hipMemcpyDeviceToHost)); }
  return SMR_OK;
}

extern "C" int smr_seed_hits_fetch(smr_ctx* c, uint32_t* triples, uint64_t cap_triples, uint64_t* n_out) {
  if (!c || !n_out) return SMR_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<unsigned long long> h;
  int rc = read_ctr(c, h); if (rc) return rc;
  uint64_t words = h[C_POOL_CURSOR] ? std::min<uint64_t>(c->pool_words, 0x7FFFFFF0ull) : 0;   // sharded pool: fetch all regions
  std::vector<uint32_t> pool(words);
  std::vector<RWork> rw(c->b->n);
  if (words) HIPCHK(c, hipMemcpy(pool.data(), c->d_pool, words * 4, hipMemcpyDeviceToHost));
  if (c->b->n) HIPCHK(c, hipMemcpy(rw.data(), c->b->d_rw, (size_t)c->b->n * sizeof(RWork), hipMemcpyDeviceToHost));
  // (the device's ids are places in its position array, id' = pos_off[id] + id: back to the index's ids for the caller)
  std::vector<uint32_t> po;
  if (c->last_seed_slot >= 0 && c->idx[c->last_seed_slot].used) {
    const DevIndex& di = c->idx[c->last_seed_slot];
    po.resize((size_t)di.n_ids + 1);
    HIPCHK(c, hipMemcpy(po.data(), di.pos_off, po.size() * 4, hipMemcpyDeviceToHost));
  }
  auto orig = [&](uint32_t idp) -> uint32_t {
    if (po.size() < 2) return idp;
    size_t lo = 0, hi = po.size() - 1;                      // the last id with pos_off[id] + id <= idp
    while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if ((uint64_t)po[mid] + mid <= idp) lo = mid; else hi = mid; }
    return (uint32_t)lo;
  };
  uint64_t o = 0;
  for (uint32_t r = 0; r < c->b->n; r++) {
    for (int pp = 0; pp < 3; pp++) {
      const uint32_t cnt = rw[r].blk_cnt[pp], bo = rw[r].blk_off[pp];
      for (uint32_t q = 0; q < cnt && (uint64_t)bo + 2 * q + 1 < words; q++) {
        if (triples && o < cap_triples) { triples[3 * o] = r; triples[3 * o + 1] = orig(pool[bo + 2 * q]); triples[3 * o + 2] = pool[bo + 2 * q + 1]; }
        o++;
      }
    }
  }
  *n_out = o;
  return SMR_OK;
}

extern "C" int smr_prof_reset(smr_ctx* c) {
  if (!c) return SMR_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  for (int k = 0; k < KP_COUNT; k++) { c->kp_ms[k] = 0; c->kp_l[k] = 0; }
  c->n_seed_shared = c->n_seed_shared_builds = 0;
  for (int k = 0; k < SMR_MAX_BATCHES; k++)
    if (c->bt[k].d_ctr) {
      HIPCHK(c, hipMemsetAsync(&c->bt[k].d_ctr[C_WINDOWS], 0, 9 * 8, c->stream));
      HIPCHK(c, hipMemsetAsync(&c->bt[k].d_ctr[C_SW_SPEC], 0, 3 * 8, c->stream));
      HIPCHK(c, hipMemsetAsync(&c->bt[k].d_ctr[C_TUP_F], 0, C_SHARD_NX * 8, c->stream));
      HIPCHK(c, hipMemsetAsync(&c->bt[k].d_ctr[C_SHARDS], 0, C_SHARD_W * C_NSHARD * 8, c->stream));
    }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SMR_OK;
}
extern "C" int smr_prof_get(smr_ctx* c, smr_prof* o) {
  if (!c || !o) return SMR_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<unsigned long long> h(C_TOTAL, 0), t(C_TOTAL);
  for (int k = 0; k < SMR_MAX_BATCHES; k++) {
    if (!c->bt[k].d_ctr) continue;
    HIPCHK(c, hipMemcpyAsync(t.data(), c->bt[k].d_ctr, C_TOTAL * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (getenv("SMR_DEBUG_PHASES")) {
      unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0};
      for (int s2 = 0; s2 < C_NSHARD; s2++) for (int q = 0; q < 7; q++) ph[q] += t[C_SHARDS + C_SHARD_W * s2 + C_SHARD_PH + q];
      fprintf(stderr, "[smr] phase cycles (batch %d): %llu %llu %llu %llu %llu %llu %llu  (-DSMR_CHAIN_PHASES: claim, gather+prefix, walk1, walk2+cands, pairs+sort, "
              "window/lis/book, sw; -DSMR_SEED_PHASES (k_seed_pg): setup, directory ranges, entries, -, candidate selection, output, -)\n", k, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6]);
    }
    fold_shards(t);
    for (int q = 0; q < C_COUNT; q++) h[q] += t[q];
  }
  o->seed_ms = c->kp_ms[KP_KEYS] + c->kp_ms[KP_SPLIT] + c->kp_ms[KP_BINS] + c->kp_ms[KP_PG0] + c->kp_ms[KP_PG1] + c->kp_ms[KP_FINISH]; o->seed_launches = c->kp_l[KP_KEYS];
  o->chain_ms = c->kp_ms[KP_CAND] + c->kp_ms[KP_CHAIN] + c->kp_ms[KP_BEGINS] + c->kp_ms[KP_WALK] + c->kp_ms[KP_SW16] + c->kp_ms[KP_WNEXT]; o->chain_launches = c->kp_l[KP_CAND] + c->kp_l[KP_BEGINS];
  o->trace_ms = c->kp_ms[KP_TRACE]; o->trace_launches = c->kp_l[KP_TRACE];
  o->n_windows = h[C_WINDOWS]; o->n_lookup = h[C_LOOKUP]; o->n_node = h[C_NODE]; o->n_entry = h[C_ENTRY]; o->n_hit = h[C_HIT]; o->n_read_bytes = h[C_READ_BYTES];
  o->n_sw_fwd = h[C_SW_FWD]; o->n_sw_rev = h[C_SW_REV]; o->n_sw_cells = h[C_SW_CELLS];
  o->n_sw_spec = h[C_SW_SPEC]; o->n_sw_spec_used = h[C_SW_SPEC_USED]; o->n_seed_redo = h[C_SEED_REDO]; o->hit_list_cap = c->hcap;
  o->n_seed_shared = c->n_seed_shared; o->n_seed_shared_builds = c->n_seed_shared_builds;
  return SMR_OK;
}

// Per kernel family: HIP-event time, launches and the algorithmic HBM bytes the kernels counted for themselves (0: not counted)
extern "C" int smr_prof_kernels(smr_ctx* c, smr_kprof* out, uint32_t cap, uint32_t* n_out) {
  if (!c || !out || !n_out) return SMR_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<unsigned long long> h(C_TOTAL, 0), t(C_TOTAL);
  for (int k = 0; k < SMR_MAX_BATCHES; k++) {
    if (!c->bt[k].d_ctr) continue;
    HIPCHK(c, hipMemcpyAsync(t.data(), c->bt[k].d_ctr, C_TOTAL * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    fold_shards(t);
    for (int q = 0; q < C_COUNT; q++) h[q] += t[q];
  }
  const unsigned long long T = h[C_TUP_ALL];
  unsigned long long bytes[KP_COUNT] = {};
  bytes[KP_KEYS] = h[C_B_KEYS] + sizeof(SeedTup) * T;                 // its inputs (counted by the kernel) + every tuple written once
  bytes[KP_SPLIT] = bytes[KP_BINS] = 2 * sizeof(SeedTup) * T;         // each of the two sort passes reads and writes every tuple once
  bytes[KP_PG0] = h[C_B_PG0]; bytes[KP_PG1] = h[C_B_PG1]; bytes[KP_FINISH] = h[C_B_FIN];
  uint32_t n = 0;
  for (int k = 0; k < KP_COUNT && n < cap; k++, n++) {
    memset(&out[n], 0, sizeof out[n]);
    snprintf(out[n].name, sizeof out[n].name, "%s", KP_NAME[k]);
    out[n].ms = c->kp_ms[k]; out[n].launches = c->kp_l[k]; out[n].bytes = bytes[k];
  }
  *n_out = n;
  return SMR_OK;
}
