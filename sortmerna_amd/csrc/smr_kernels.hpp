// smr_kernels.hpp -- hand-written HIP kernels for gfx950 (CDNA4, wave64) of SortMeRNA's per-read hot path.
//
//   k_begin_part / k_begin_strand / k_commit_part : per-read state bookkeeping (processor.cpp:104-161)
//   k_seed   : window scan (9-mer hash + lookup probe) + burst-trie descent with the LEV(1) automaton
//              (paralleltraversal.cpp:124-249, traverse_bursttrie.cpp:100-298, bitvector.cpp:57-132)
//   k_cand   : per read "is there any candidate reference at all?" (Bloom bitmap over the references of all seed positions); the
//              reads without one end their pass here (alignment.cpp:117-148, paralleltraversal.cpp:253-297)
//   k_chain  : exact candidate set, candidate walk as a task generator (LIS, SW window geometry), Smith-Waterman four problems per
//              wave, accept/best-N bookkeeping, pass control (alignment.cpp:100-509, ssw.c:834-899, paralleltraversal.cpp:253-297)
//   k_begins : begin cells of the alignments that are still stored (reverse SW passes, ssw.c:900-918), four per wave
//   k_trace_band / k_trace_wide : banded DP with lanes across the band (F by prefix scan) + walk back -> CIGAR (what ssw.c:577-773 yields)
//
// Design notes (DESIGN.md has the long form):
//  * integer work, VALU-issue / latency bound: no MFMA anywhere.
//  * seed stage: the windows of the whole batch are binned by 9-mer key (counting sort) and searched in key order, one lane per
//    search over the exact-key directories of the pigeonhole layout (k_seed_pg); lane-local hit lists in LDS.
//  * k_cand: 16 lanes per read, four reads per wave; the positions of a read it marks are handed to k_chain as a record.  k_chain: one wave
//    per marked read, persistent blocks claiming up to 64 reads at a time from an atomic counter (per-read work varies by orders of
//    magnitude); the data-parallel pieces (position walks, bitonic sorts, LIS as patience piles across lanes, the SW systolic arrays) use
//    all 64 lanes, control flow is wave-uniform.
//  * k_trace_*: lanes across the band diagonals, one DP row per step; 4-bit direction flags in LDS (short reads) or a global tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smr_host.hpp"

#include <smr_device_ops.hpp>      // SMR_DYN_LDS, packed 16-bit ops, v_perm_b32, v_rcp_f32 (the test suite's kernel emulator supplies its own)

namespace smr {

// ------------------------------------------------------------------------------------------------
// device-side views
// ------------------------------------------------------------------------------------------------
struct DIndex {
  const Lookup* lookup;
  const uint32_t* lkc;         // min(count, 2^30 - 1) | forward mini-trie present << 30 | reverse present << 31
  const uint32_t* trie;
  const uint32_t* pg;          // pigeonhole arena (smr_host.hpp) and its block table: forward / reverse mini-trie of key k at [2k], [2k+1] = {offset / 4, n | cA << 24 | cB << 28}
  const uint2* root3;
  const uint32_t* pos_off;     // the index's CSR offsets (id -> list start): what a search's id is made of, id' = pos_off[id] + id
  const uint2* pos_arr;        // per seed, at [id']: a header {positions, the index's id}, then its positions {pos, seq} (k_pos2_build)
  const uint8_t* ref_seq;
  const uint64_t* ref_off;
  uint32_t n_refs, n_ids, lnwin, partialwin;
  uint32_t ref_any_n;          // 0: no reference of this index part holds an ambiguous letter (4)
};

struct DReads {
  const uint32_t* words;
  const uint64_t* rec_off;
  const uint32_t* len;
  uint32_t n, max_len;
};

struct RState {                 // what round-trips through the reference's KVDB (read.cpp:429-539)
  uint32_t lastIndex, lastPart, hit_seeds, min_index, max_index, n_align;
  uint16_t max_SW_count;
  uint8_t is_done, is_hit;
};

struct RWork {                  // per (read, index part) transient state
  int32_t best;                 // Read::best (not stored in the KVDB, read.hpp:122)
  uint32_t blk_off[3];          // per pass: pool offset of this strand's contiguous (id, win_pos) pairs
  uint32_t blk_cnt[3];          // ... and their number
  uint32_t hit_total;           // number of (id,win) hits of this strand so far
  uint32_t win_shift;
  uint8_t is_new_hit;
  uint8_t is04;                 // Read::is04 (only ever toggles when the read has ambiguous letters)
  uint8_t aval;                 // current value stored at ambiguous positions: 0, 3 or 4
  uint8_t reversed;
  uint8_t valid;                // long enough and not done at part start
  uint8_t strand_active;
  uint8_t search;               // `search` of traverse()
  uint8_t pass_n;               // index into skiplengths of the pass to run next
  uint8_t has_amb;              // read has letters outside ACGTU (Read::ambiguous_nt non-empty)
  uint8_t pad_[3];
};

struct AlignRec {               // s_align2 (ssw.hpp:44-58) + where its CIGAR lives
  uint32_t ref_num;
  int32_t ref_begin1, ref_end1, read_begin1, read_end1;
  uint32_t readlen;
  uint16_t score1, part, index_num;
  uint8_t strand, has_cigar;
  uint32_t cigar_off, cigar_len;
};

struct DParams {
  uint32_t lnwin, partialwin;
  uint32_t skip[3];
  int32_t num_seeds, min_lis, edges, is_as_percent;
  int32_t match, mismatch, score_N, gap_open, gap_ext;
  uint32_t minimal_score, num_alignments;
  int32_t is_best, is_full_search, is_forward, is_reverse;
  uint32_t minoccur, index_num, part;
  int32_t is_last_index_part;
  uint32_t slots;               // alignment slots per read
  int32_t sw_mode;              // 1 / 2: packed 16-bit Smith-Waterman kernels where they apply (smr_sw_pk.hpp), 0: 32-bit kernel only, < 0: ssw.c's stripe geometry (smr_sw_striped.hpp)
  uint16_t* sw_scratch;         // sw_mode < 0: scratch rows of the striped slow path, sw_scratch_stride uint16 per block
  uint32_t sw_scratch_stride;
};
__device__ __forceinline__ uint16_t* sw_scr(const DParams& P) { return P.sw_scratch ? P.sw_scratch + (size_t)blockIdx.x * P.sw_scratch_stride : nullptr; }

// global counters (u64 each)
enum {
  C_NUM_ALIGNED = 0, C_NUM_SHORT = 1, C_PER_DB = 2,           // C_PER_DB .. C_PER_DB+63
  C_WINDOWS = 66, C_LOOKUP, C_NODE, C_ENTRY, C_HIT, C_READ_BYTES, C_SW_FWD, C_SW_REV, C_SW_CELLS,
  C_ERR_HITCAP, C_ERR_POOL, C_ERR_SLOTS, C_ERR_PAIRS, C_ERR_CIGAR, C_ERR_TRACE, C_POOL_CURSOR, C_WORK_NEXT,
  C_CIGAR_CURSOR, C_TRACE_NEXT, C_ERR_SCAP, C_ERR_REDO, C_TRACE_DEFER, C_BEGIN_N, C_BEGIN_NEXT, C_FETCH_N, C_SW_SPEC, C_SW_SPEC_USED, C_SEED_REDO,
  // what the seed-stage kernels THEMSELVES move, for the roofline of each (smr_prof_kernels): tuples of the forward searches / of both
  // (every tuple is written once by k_seed_keys, read and written once by each of the two sort passes, read once by k_seed_pg), and the
  // algorithmic HBM bytes of k_seed_keys' inputs, of the two search launches and of k_seed_finish -- every kernel documents its own sum
  C_TUP_F, C_TUP_ALL, C_B_KEYS, C_B_PG0, C_B_PG1, C_B_FIN, C_COUNT = 112,
  // Work counters and the pool cursor are sharded 64 ways (by block id): one address would serialise ~10 ns per
  // atomic over ~10^6 waves.  Shard s keeps counter C_WINDOWS+k (k < 9) at C_SHARDS + 32*s + k and C_TUP_F+k at C_SHARDS + 32*s + 9 + k
  // (slots 16.. of a shard: the cycle counters of the -DSMR_*_PHASES debug builds); the host folds them.
  C_NSHARD = 64, C_SHARD_W = 32, C_SHARD_X = 9, C_SHARD_NX = 6, C_SHARD_PH = 16, C_SHARDS = C_COUNT, C_PCUR = C_SHARDS + C_SHARD_W * C_NSHARD, C_PCUR_STRIDE = 16 /* a 128-byte line per cursor: atomics on one line queue up */, C_TOTAL = C_PCUR + C_NSHARD * C_PCUR_STRIDE
};
__device__ __forceinline__ void ctr_add(unsigned long long* ctr, int idx, unsigned long long v) {
  atomicAdd(&ctr[C_SHARDS + (blockIdx.x & (C_NSHARD - 1)) * C_SHARD_W + (idx >= C_TUP_F ? C_SHARD_X + idx - C_TUP_F : idx - C_WINDOWS)], v);
}

// LEV(1) universal automaton (traverse_bursttrie.cpp:68-98), flattened: t0[16][14], t1[8][14], t2[4][14], t3[2][14]
__device__ __constant__ uint8_t c_lev[(16 + 8 + 4 + 2) * 14] = {
    3,14,14,14,14,14,14,14,14,14,14,14,14,14, 3,14,14,14,14,14,14,14,14,14,14,14,14,14,
    7,14,14,14,4,4,4,4,14,14,14,14,14,14, 7,14,14,14,4,4,4,4,14,14,14,14,14,14,
    0,14,2,2,14,14,2,2,14,14,14,14,14,14, 0,14,2,2,14,14,2,2,14,14,14,14,14,14,
    0,14,2,2,4,4,6,6,14,14,14,14,14,14, 0,14,2,2,4,4,6,6,14,14,14,14,14,14,
    3,1,14,1,14,1,14,1,14,14,14,14,14,14, 3,1,14,1,14,1,14,1,14,14,14,14,14,14,
    7,1,14,1,4,5,4,5,14,14,14,14,14,14, 7,1,14,1,4,5,4,5,14,14,14,14,14,14,
    0,1,2,3,14,1,2,3,14,14,14,14,14,14, 0,1,2,3,14,1,2,3,14,14,14,14,14,14,
    0,1,2,3,4,5,6,7,14,14,14,14,14,14, 0,1,2,3,4,5,6,7,14,14,14,14,14,14,
    // t1 (3-bit vectors)
    3,14,14,14,14,14,14,14,14,14,14,14,14,14, 13,14,14,14,10,10,10,10,14,14,14,14,14,14,
    8,14,2,2,14,14,2,2,14,14,14,14,14,14, 8,14,2,2,10,10,12,12,14,14,14,14,14,14,
    3,1,14,1,14,1,14,1,14,14,14,14,14,14, 13,1,14,1,10,11,10,11,14,14,14,14,14,14,
    8,1,2,3,14,1,2,3,14,14,14,14,14,14, 8,1,2,3,10,11,12,13,14,14,14,14,14,14,
    // t2 (2-bit vectors)
    12,14,14,14,14,14,14,14,12,14,14,14,14,14, 9,14,10,10,14,14,10,10,9,14,14,14,10,10,
    12,1,14,1,14,1,14,1,12,14,14,1,14,1, 9,1,10,12,14,1,10,12,9,14,14,1,10,12,
    // t3 (1-bit vectors)
    10,14,14,14,14,14,14,14,14,10,14,14,14,14, 10,10,14,10,14,10,14,10,14,10,14,14,10,14};
#define LEV_T1 (16 * 14)
#define LEV_T2 (LEV_T1 + 8 * 14)
#define LEV_T3 (LEV_T2 + 4 * 14)
#define LEV_SIZE ((16 + 8 + 4 + 2) * 14)

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Append from a whole block with ONE returning atomic (blocks of up to 1024 threads; every thread must call it).  A counter that is hit by a
// returning atomic from every wave of a launch is a queue: 83 per microsecond and 128-byte line on the MI355X whatever the CUs do, returning or not (tools/microbench/atomics.hip) -- 125 000 waves, 1.5 ms
// for a kernel that reads 200 MB (round 4: k_begins_collect, k_trace_collect, k_results_compact).  The compiler already folds a wave's lanes
// into one atomic; this folds the block's waves.
__device__ __forceinline__ uint32_t block_append(unsigned long long* counter, bool take) {
  __shared__ uint32_t s_wcnt[16], s_wbase[16];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, nwv = (blockDim.x + 63u) >> 6;
  const unsigned long long m = __ballot(take);
  if (lane == 0) s_wcnt[wv] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (uint32_t q = 0; q < nwv; q++) { s_wbase[q] = tot; tot += s_wcnt[q]; }
    const uint32_t base = tot ? (uint32_t)atomicAdd(counter, (unsigned long long)tot) : 0u;
    for (uint32_t q = 0; q < nwv; q++) s_wbase[q] += base;
  }
  __syncthreads();
  return s_wbase[wv] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// ------------------------------------------------------------------------------------------------
// read decoding: nt k of read record in the CURRENT strand/encoding state.
// forward:  code or aval at ambiguous positions; reversed: 3-code of position len-1-k (Read::revIntStr,
// read.cpp:350-357) and aval at (mirrored) ambiguous positions (Read::flip34, read.cpp:379-401).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t read_nt(const uint32_t* rec, uint32_t len, uint32_t k, uint32_t reversed, uint32_t aval) {
  uint32_t j = reversed ? (len - 1 - k) : k;
  uint32_t cw = (len + 15) >> 4;
  uint32_t code = (rec[j >> 4] >> ((j & 15) * 2)) & 3u;
  uint32_t amb = (rec[cw + (j >> 5)] >> (j & 31)) & 1u;
  if (reversed) code = 3u - code;
  return amb ? aval : code;
}

// ------------------------------------------------------------------------------------------------
// state bookkeeping kernels
// ------------------------------------------------------------------------------------------------
__global__ void k_begin_part(DReads rd, DParams P, const RState* __restrict__ saved, const AlignRec* __restrict__ saved_aln,
                             RState* __restrict__ work, AlignRec* __restrict__ work_aln, RWork* __restrict__ rw,
                             unsigned long long* __restrict__ ctr) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t is_short = 0;
  if (i < rd.n) {
    RState s = saved[i];
    work[i] = s;
    for (uint32_t k = 0; k < s.n_align; k++) work_aln[(size_t)i * P.slots + k] = saved_aln[(size_t)i * P.slots + k];
    RWork w;
    w.best = P.min_lis > 0 ? P.min_lis : 0;          // Read::init read.cpp:264-271
    for (int q = 0; q < 3; q++) { w.blk_off[q] = 0; w.blk_cnt[q] = 0; }
    w.hit_total = 0; w.win_shift = P.skip[0];
    w.is_new_hit = 0; w.is04 = 0; w.aval = 0; w.reversed = 0;
    is_short = rd.len[i] < P.lnwin;                   // processor.cpp:109-114
    w.valid = (!is_short && !s.is_done) ? 1 : 0;      // :120-126
    w.strand_active = 0; w.search = 0; w.pass_n = 0;
    {
      const uint32_t ln = rd.len[i]; const uint32_t* rec = rd.words + rd.rec_off[i];
      uint32_t cw = (ln + 15) >> 4, mw = (ln + 31) >> 5, any = 0;
      for (uint32_t q = 0; q < mw; q++) any |= rec[cw + q];
      w.has_amb = any ? 1 : 0; w.pad_[0] = w.pad_[1] = w.pad_[2] = 0;
    }
    rw[i] = w;
  }
  unsigned long long m = __ballot(is_short);
  if (lane_id() == 0 && m) atomicAdd(&ctr[C_NUM_SHORT], (unsigned long long)__popcll(m));
}

// strand loop head: processor.cpp:137-147
__global__ void k_begin_strand(uint32_t n, DParams P, int count, const RState* __restrict__ work, RWork* __restrict__ rw) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  RWork w = rw[i];
  w.strand_active = 0;
  if (w.valid && !work[i].is_done) {
    int single = (P.is_forward != 0) ^ (P.is_reverse != 0);
    if ((single && P.is_reverse) || count == 1) {
      if (!w.reversed) {                                 // Read::revIntStr: complement {3,2,1,0,4}
        w.reversed = 1;
        w.aval = (w.aval == 0) ? 3 : (w.aval == 3 ? 0 : 4);
      }
    }
    w.strand_active = 1; w.search = 1; w.pass_n = 0; w.win_shift = P.skip[0];
    for (int q = 0; q < 3; q++) w.blk_cnt[q] = 0;       // read.id_win_hits.clear()
    w.hit_total = 0;
  }
  rw[i] = w;
}

// kvdb.put only when is_new_hit: processor.cpp:150-155
__global__ void k_commit_part(uint32_t n, DParams P, RState* __restrict__ saved, AlignRec* __restrict__ saved_aln,
                              const RState* __restrict__ work, const AlignRec* __restrict__ work_aln, const RWork* __restrict__ rw) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (rw[i].valid && rw[i].is_new_hit) {
    RState s = work[i];
    saved[i] = s;
    for (uint32_t k = 0; k < s.n_align; k++) saved_aln[(size_t)i * P.slots + k] = work_aln[(size_t)i * P.slots + k];
  }
}

}  // namespace smr

#include "smr_seed.hpp"
#include "smr_seed_pg.hpp"
#include "smr_chain.hpp"
#include "smr_walk.hpp"
#include "smr_trace.hpp"
