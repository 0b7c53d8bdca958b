// smr_kernels.hpp -- hand-written HIP kernels for gfx950 (CDNA4, wave64) of SortMeRNA's per-read hot path.
//
//   k_begin_part / k_begin_strand / k_commit_part : per-read state bookkeeping (processor.cpp:104-161)
//   k_seed   : window scan (9-mer hash + lookup probe) + burst-trie descent with the LEV(1) automaton
//              (paralleltraversal.cpp:124-249, traverse_bursttrie.cpp:100-298, bitvector.cpp:57-132)
//   k_chain  : candidate histogram, LIS chaining, SW window geometry, Smith-Waterman score + begin position,
//              accept/best-N bookkeeping, pass control (alignment.cpp:100-509, ssw.c:834-918,
//              paralleltraversal.cpp:253-297)
//   k_trace  : banded traceback -> CIGAR (ssw.c:577-773)
//
// Design notes (DESIGN.md has the long form):
//  * integer work, HBM/latency bound: no MFMA anywhere.
//  * k_seed: a wave is split into groups of `gw` lanes (gw = pow2 >= windows per read in this pass); each
//    group owns one read, each lane one window.  Lane-local hit lists live in LDS (k*64+lane layout, conflict
//    free); per-group compaction uses a segmented wave prefix sum and ONE atomic on the global hit pool.
//  * k_chain: one wave per read (persistent blocks pull reads from an atomic counter because per-read work
//    varies by orders of magnitude).  Control flow is wave-uniform and follows the reference statement by
//    statement; the data-parallel pieces (position-list walks, bitonic sorts, the SW anti-diagonal systolic
//    array with lane = read row) use all 64 lanes.
//  * k_trace: one thread per alignment, direction bytes interleaved by lane so a wave's stores coalesce.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smr_host.hpp"

namespace smr {

// ------------------------------------------------------------------------------------------------
// device-side views
// ------------------------------------------------------------------------------------------------
struct DIndex {
  const Lookup* lookup;
  const uint32_t* trie;
  const uint32_t* pos_off;
  const uint2* pos_arr;        // {pos, seq}
  const uint8_t* ref_seq;
  const uint64_t* ref_off;
  uint32_t n_refs, n_ids, lnwin, partialwin;
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
  uint32_t hit_head;            // head of this strand's hit-segment list in the pool (NONE = empty)
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
};

// global counters (u64 each)
enum {
  C_NUM_ALIGNED = 0, C_NUM_SHORT = 1, C_PER_DB = 2,           // C_PER_DB .. C_PER_DB+63
  C_WINDOWS = 66, C_LOOKUP, C_NODE, C_ENTRY, C_HIT, C_READ_BYTES, C_SW_FWD, C_SW_REV, C_SW_CELLS,
  C_ERR_HITCAP, C_ERR_POOL, C_ERR_SLOTS, C_ERR_PAIRS, C_ERR_CIGAR, C_ERR_TRACE, C_POOL_CURSOR, C_WORK_NEXT,
  C_CIGAR_CURSOR, C_TRACE_NEXT, C_COUNT = 96
};

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
    w.hit_head = NONE; w.hit_total = 0; w.win_shift = P.skip[0];
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
    w.hit_head = NONE; w.hit_total = 0;                 // read.id_win_hits.clear()
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

// ------------------------------------------------------------------------------------------------
// k_seed
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lev_step(const uint8_t* lev, uint32_t depth, uint32_t partialwin, uint32_t bv_lo_hi_sel_nibble,
                                             uint32_t last_row_nibble, uint32_t state) {
  // depth < partialwin-2: 4-bit vector into t0; else the last row masked to (partialwin-depth+1) bits
  if (depth < partialwin - 2) return lev[bv_lo_hi_sel_nibble * 14 + state];
  uint32_t t = 3 - partialwin + depth;                         // 1,2,3
  uint32_t v = last_row_nibble & ((2u << (partialwin - depth)) - 1u);
  uint32_t base = t == 1 ? LEV_T1 : (t == 2 ? LEV_T2 : LEV_T3);
  return lev[base + v * 14 + state];
}

// characteristic bit-vectors of a 9-mer (bitvector.cpp:57-132): nibble (d, nt), bit k set iff c[d+2-k] == nt
struct BitVec {
  unsigned long long lo, hi;      // nibble index e = d*4+nt ; e < 16 -> lo, else hi
  __device__ __forceinline__ uint32_t get(uint32_t d, uint32_t nt) const {
    uint32_t e = d * 4 + nt;
    return (uint32_t)((e < 16 ? (lo >> (e * 4)) : (hi >> ((e - 16) * 4))) & 15ull);
  }
};
__device__ __forceinline__ BitVec make_bitvec(uint32_t chars /*2 bits per char, char i at bits 2i*/, uint32_t partialwin) {
  BitVec b; b.lo = 0; b.hi = 0;
  uint32_t rows = partialwin - 2;
  for (uint32_t d = 0; d < rows; d++) {
    for (uint32_t k = 0; k < 4; k++) {
      int ci = (int)d + 2 - (int)k;
      if (ci < 0) continue;
      uint32_t nt = (chars >> (2 * ci)) & 3u;
      uint32_t e = d * 4 + nt;
      if (e < 16) b.lo |= (unsigned long long)(1u << k) << (e * 4);
      else b.hi |= (unsigned long long)(1u << k) << ((e - 16) * 4);
    }
  }
  return b;
}

struct SeedCounters { uint32_t lookup, node, entry; };

// traversetrie_align (traverse_bursttrie.cpp:100-298), iterative.  Lane-local hit list in LDS: hl[k*64+lane].
// returns true when a 0-error match was found (accept_zero_kmer).
__device__ bool trie_search(const uint32_t* __restrict__ trie, uint32_t root, const BitVec& bv, uint32_t partialwin,
                            int is_full_search, const uint8_t* lev, uint32_t* hl, uint32_t hcap, uint32_t& nh, bool& overflow,
                            SeedCounters& sc) {
  const int lane = lane_id();
  uint32_t st_node[12]; uint8_t st_ne[12]; uint8_t st_pivot[12];
  int sp = 0;
  st_node[0] = 0; st_ne[0] = 0; st_pivot[0] = 0;
  const uint32_t last_row = partialwin - 3;
  sc.node++;
  while (sp >= 0) {
    if (st_ne[sp] == 4) { sp--; continue; }
    uint32_t ne = st_ne[sp]++;
    uint32_t e = trie[root + st_node[sp] + ne];
    uint32_t flag = e >> ELEM_FLAG_SHIFT;
    if (flag == 0) continue;
    uint32_t depth = (uint32_t)sp;
    uint32_t lev_t = lev_step(lev, depth, partialwin, depth < partialwin - 2 ? bv.get(depth, ne) : 0, bv.get(last_row, ne), st_pivot[sp]);
    if (lev_t == 14) continue;
    if (flag == 1) {
      sp++;
      st_node[sp] = e & ELEM_OFF_MASK; st_ne[sp] = 0; st_pivot[sp] = (uint8_t)lev_t;
      sc.node++;
      continue;
    }
    // bucket
    uint32_t nent = (e >> ELEM_NENT_SHIFT) & 0xFFu;
    const uint32_t* bp = trie + root + (e & ELEM_OFF_MASK);
    uint32_t s = partialwin - depth;
    for (uint32_t q = 0; q < nent; q++) {
      uint32_t entry_str = bp[2 * q];
      uint32_t id = bp[2 * q + 1];
      uint32_t depth_b = depth;
      uint32_t lv = lev_t;
      bool local_accept = false, zero = false;
      sc.entry++;
      for (uint32_t j = 0; j < s; j++) {
        uint32_t nt = entry_str & 3u;
        depth_b++;
        lv = lev_step(lev, depth_b, partialwin, depth_b < partialwin - 2 ? bv.get(depth_b, nt) : 0, bv.get(last_row, nt), lv);
        if (lv == 14) break;
        if (depth_b >= partialwin - 2) {
          if (lv >= 8) local_accept = true;
          if (depth_b == partialwin - 1 && lv == 9) zero = !is_full_search;
        }
        if (local_accept) {
          if (zero) { nh = 0; hl[0 * 64 + lane] = id; nh = 1; return true; }
          bool found = false;
          for (uint32_t f = 0; f < nh; f++) if (hl[f * 64 + lane] == id) { found = true; break; }
          if (found) break;
          if (nh < hcap) hl[nh * 64 + lane] = id; else overflow = true;
          if (nh < hcap) nh++;
        }
        entry_str >>= 2;
      }
    }
  }
  return false;
}

// One wave = 64/gw reads; lane (g*gw + w) searches window w of the group's read in pass `pass`.
// dynamic LDS: per wave 64*hcap u32 hit ids.
__global__ void __launch_bounds__(256) k_seed(DReads rd, DIndex ix, DParams P, int pass, uint32_t gw, uint32_t hcap,
                                              RState* __restrict__ work, RWork* __restrict__ rw, uint32_t* __restrict__ pool,
                                              uint32_t pool_words, unsigned long long* __restrict__ ctr) {
  extern __shared__ uint32_t lds_dyn[];
  __shared__ uint8_t s_lev[LEV_SIZE];
  for (uint32_t i = threadIdx.x; i < LEV_SIZE; i += blockDim.x) s_lev[i] = c_lev[i];
  __syncthreads();
  const int lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  uint32_t* hl = lds_dyn + (size_t)wave * 64 * hcap;
  const uint32_t gpw = 64 / gw;                                   // groups (reads) per wave
  const uint32_t waves_per_block = blockDim.x >> 6;
  const uint32_t g = lane / gw, wl = lane % gw;
  const uint32_t r = (blockIdx.x * waves_per_block + wave) * gpw + g;
  const uint32_t pw = P.partialwin, L = P.lnwin;

  bool active = false;
  RWork w;
  uint32_t len = 0;
  const uint32_t* rec = nullptr;
  if (r < rd.n) {
    w = rw[r];
    active = (w.strand_active && w.search && w.pass_n == (uint32_t)pass);
    len = rd.len[r];
    rec = rd.words + rd.rec_off[r];
  }
  uint32_t aval = 0, reversed = 0;
  if (active) {
    // traverse(): `if (read.is04) read.flip34()` before every window (paralleltraversal.cpp:126)
    aval = w.is04 ? 0 : w.aval;
    reversed = w.reversed;
  }
  const uint32_t stride = P.skip[pass];
  const uint32_t numwin = active ? (len - L + stride) / stride : 0;       // :118-120

  SeedCounters sc; sc.lookup = 0; sc.node = 0; sc.entry = 0;
  uint32_t n_win_searched = 0, grp_hits_total = 0, grp_seeds = 0;
  bool overflow = false;
  uint32_t seg_head = active ? w.hit_head : NONE;

  for (uint32_t wbase = 0; __any(wbase < numwin); wbase += gw) {
    uint32_t k = wbase + wl;
    uint32_t nh = 0;
    bool mine = active && k < numwin;
    uint32_t win_pos = k * stride;
    if (mine) {                                        // read_pos_searched (paralleltraversal.cpp:128-131)
      for (int q = 0; q < pass; q++) if (win_pos % P.skip[q] == 0) mine = false;
    }
    if (mine) {
      n_win_searched++;
      // window content, char i at bits 2i
      unsigned long long wchars = 0;
      for (uint32_t i = 0; i < L; i++) wchars |= (unsigned long long)read_nt(rec, len, win_pos + i, reversed, aval) << (2 * i);
      // forward half: key = first partialwin chars (MSB first, Read::hashKmer read.cpp:601-611),
      // bit-vectors from chars [pw .. 2pw) (init_win_f)
      uint32_t keyf = 0, keyr = 0, fchars = 0, rchars = 0;
      for (uint32_t i = 0; i < pw; i++) {
        keyf = (keyf << 2) | (uint32_t)((wchars >> (2 * i)) & 3);
        keyr = (keyr << 2) | (uint32_t)((wchars >> (2 * (pw + i))) & 3);
        fchars |= (uint32_t)((wchars >> (2 * (pw + i))) & 3) << (2 * i);
        rchars |= (uint32_t)((wchars >> (2 * (pw - 1 - i))) & 3) << (2 * i);      // init_win_r walks backwards from pw-1
      }
      bool zero = false;
      Lookup lk = ix.lookup[keyf]; sc.lookup++;
      if (lk.count > P.minoccur && lk.rootF != NONE) {
        BitVec bv = make_bitvec(fchars, pw);
        zero = trie_search(ix.trie, lk.rootF, bv, pw, P.is_full_search, s_lev, hl, hcap, nh, overflow, sc);
      }
      if (!zero) {
        Lookup lr = ix.lookup[keyr]; sc.lookup++;
        if (lr.count > P.minoccur && lr.rootR != NONE) {
          BitVec bv = make_bitvec(rchars, pw);
          trie_search(ix.trie, lr.rootR, bv, pw, P.is_full_search, s_lev, hl, hcap, nh, overflow, sc);
        }
      }
    }
    // segmented (width gw) inclusive scan of nh
    uint32_t incl = nh;
    for (uint32_t d = 1; d < gw; d <<= 1) { uint32_t t = __shfl_up(incl, d, gw); if (wl >= d) incl += t; }
    uint32_t total = __shfl(incl, gw - 1, gw);
    uint32_t seeds = __popcll(__ballot(nh > 0) & (gw == 64 ? ~0ull : (((1ull << gw) - 1) << (g * gw))));
    uint32_t base = 0;
    if (wl == 0 && total > 0) {
      unsigned long long old = atomicAdd(&ctr[C_POOL_CURSOR], (unsigned long long)(2 + 2 * total));
      if (old + 2 + 2 * (unsigned long long)total > pool_words) { atomicAdd(&ctr[C_ERR_POOL], 1ull); base = NONE; }
      else { base = (uint32_t)old; pool[base] = seg_head; pool[base + 1] = total; seg_head = base; }
    }
    base = __shfl(base, 0, gw);
    if (total > 0 && base != NONE) {
      uint32_t o = base + 2 + 2 * (incl - nh);
      for (uint32_t q = 0; q < nh; q++) { pool[o + 2 * q] = hl[q * 64 + lane]; pool[o + 2 * q + 1] = win_pos; }
    }
    grp_hits_total += total; grp_seeds += seeds;
  }
  if (overflow) atomicAdd(&ctr[C_ERR_HITCAP], 1ull);
  if (active && wl == 0) {
    w.is04 = 0; w.aval = (uint8_t)aval;                     // the flip back to 0..3 is persistent
    w.hit_head = seg_head; w.hit_total += grp_hits_total;
    rw[r] = w;
    work[r].hit_seeds += grp_seeds;                          // ++read.hit_seeds per window with hits (:242-249)
  }
  // work counters
  unsigned long long v[6];
  v[0] = n_win_searched; v[1] = sc.lookup; v[2] = sc.node; v[3] = sc.entry; v[4] = (wl == 0) ? grp_hits_total : 0;
  v[5] = (active && wl == 0) ? ((len + 3) / 4) : 0;
  for (int c = 0; c < 6; c++) {
    unsigned long long x = v[c];
    for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
    if (lane == 0 && x) atomicAdd(&ctr[C_WINDOWS + c], x);
  }
}

// ------------------------------------------------------------------------------------------------
// k_chain helpers (block = one wave)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  for (int d = 32; d > 0; d >>= 1) { unsigned long long o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
  return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t v, uint32_t& total) {
  uint32_t incl = v; const int lane = lane_id();
  for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
  total = __shfl(incl, 63, 64);
  return incl - v;
}

// in-wave bitonic sort of n u64 keys (ascending); buffer must hold npow2 >= n entries
__device__ void wave_sort_u64(unsigned long long* keys, uint32_t n) {
  const int lane = lane_id();
  uint32_t np = 1; while (np < n) np <<= 1;
  for (uint32_t i = n + lane; i < np; i += 64) keys[i] = ~0ull;
  __syncthreads();
  for (uint32_t k = 2; k <= np; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = lane; i < np; i += 64) {
        uint32_t p = i ^ j;
        if (p > i) {
          unsigned long long a = keys[i], b = keys[p];
          bool up = (i & k) == 0;
          if ((a > b) == up) { keys[i] = b; keys[p] = a; }
        }
      }
      __syncthreads();
    }
  }
}

struct SwRes { int score, end_ref, end_read; };

// Smith-Waterman score + end cell as an anti-diagonal systolic array: lane = read row within a strip of 64 rows,
// step t computes column t-lane.  Same H as the reference's striped SSE2 kernels (ssw.c:150-575) under the
// affine model H = max(0, diag+s, E, F), first gap base costs gap_open, further bases gap_ext; end cell =
// first column (in processing order) where the maximum is first reached, smallest row in that column
// (ssw.c:305-336).  dir = +1 forward, -1 reverse (the reverse pass ssw.c:900-918 runs the same recurrence on the
// reversed prefixes; since its maximum equals the forward score, stopping at `terminate` selects the same cell).
// rdq: read in 0..4 alphabet (LDS), rfq: reference window (LDS).  bound: 2*n ints of LDS.
__device__ SwRes sw_wave(const uint8_t* rdq, int m, int rd0, int rdstep, const uint8_t* rfq, int n, int rf0, int rfstep,
                         int* bound, int match, int mismatch, int scoreN, int go, int ge) {
  const int lane = lane_id();
  int bestH = 0, bestcol = 0x7fffffff, bestrow = 0x7fffffff;
  const int nstrips = (m + 63) >> 6;
  for (int s = 0; s < nstrips; s++) {
    const int row = s * 64 + lane;
    const bool vrow = row < m;
    const int rnt = vrow ? rdq[rd0 + rdstep * row] : 4;
    int Hcur = 0, Fcur = 0, Ecur = 0, Hdiag = 0;
    const int steps = n + 63;
    for (int t = 0; t < steps; t++) {
      int upH = __shfl_up(Hcur, 1, 64);
      int upF = __shfl_up(Fcur, 1, 64);
      if (lane == 0) {
        if (s == 0 || t >= n) { upH = 0; upF = 0; }
        else { upH = bound[2 * t]; upF = bound[2 * t + 1]; }
      }
      const int col = t - lane;
      const bool act = vrow && col >= 0 && col < n;
      int Hn = Hcur, Fn = Fcur, En = Ecur;
      if (act) {
        const int fnt = rfq[rf0 + rfstep * col];
        const int sc = (fnt == 4 || rnt == 4) ? scoreN : (fnt == rnt ? match : mismatch);
        const int Hleft = (col == 0) ? 0 : Hcur;
        const int Eleft = (col == 0) ? 0 : Ecur;
        int e = max(Eleft - ge, Hleft - go);
        int f = max(upF - ge, upH - go);
        int h = Hdiag + sc;
        h = max(h, e); h = max(h, f); h = max(h, 0);
        e = max(e, 0); f = max(f, 0);
        Hn = h; Fn = f; En = e;
        if (h > bestH || (h == bestH && (col < bestcol || (col == bestcol && row < bestrow)))) {
          if (h > 0) { bestH = h; bestcol = col; bestrow = row; }
        }
      }
      // diag for the next step is the up value of this step (H(row-1, col))
      Hdiag = (col >= 0) ? upH : 0;
      Hcur = Hn; Fcur = Fn; Ecur = En;
      if (lane == 63 && act && s + 1 < nstrips) { bound[2 * col] = Hn; bound[2 * col + 1] = Fn; }
    }
    __syncthreads();
  }
  // lexicographic reduce: max H, then min col, then min row
  unsigned long long key = bestH > 0 ? (((unsigned long long)bestH << 42) | ((unsigned long long)(0x1FFFFF - bestcol) << 21) |
                                        (unsigned long long)(0x1FFFFF - bestrow)) : 0ull;
  key = wave_max_u64(key);
  SwRes r;
  if (key == 0) { r.score = 0; r.end_ref = -1; r.end_read = m - 1; return r; }
  r.score = (int)(key >> 42);
  r.end_ref = 0x1FFFFF - (int)((key >> 21) & 0x1FFFFF);
  r.end_read = 0x1FFFFF - (int)(key & 0x1FFFFF);
  return r;
}

// per-block (= per persistent wave slot) scratch in global memory
struct ChainScratch {
  uint32_t* cnt;                    // n_refs counters, all zero between reads
  unsigned long long* keys;         // candidate keys, capacity keys_cap (>= pow2(n_refs))
  unsigned long long* pairs;        // hits on one reference, capacity pairs_cap (pow2)
  uint32_t* lis;                    // 2 * pairs_cap (b and p arrays of find_lis)
  uint2* hits;                      // gathered (id,win) of the read, capacity hits_cap
  uint32_t keys_cap, pairs_cap, hits_cap;
};

#define CH_KEYS_LDS 512
#define CH_PAIRS_LDS 256
#define CH_HITS_LDS 256

// find_lis (alignment.cpp:58-98) over a[0..n): keys = ref_pos<<32 | read_pos ; compares read_pos (.second).
// executed redundantly by every lane (uniform control flow); b,p hold indices.
__device__ uint32_t find_lis_dev(const unsigned long long* a, uint32_t n, uint32_t* b, uint32_t* p) {
  if (n == 0) return 0;
  uint32_t nb = 0;
  for (uint32_t i = 0; i < n; i++) p[i] = 0;
  b[nb++] = 0;
  for (uint32_t i = 1; i < n; i++) {
    uint32_t ai = (uint32_t)a[i];
    if ((uint32_t)a[b[nb - 1]] < ai) { p[i] = b[nb - 1]; b[nb++] = i; continue; }
    uint32_t u = 0, v = nb - 1;
    while (u < v) { uint32_t c = (u + v) / 2; if ((uint32_t)a[b[c]] < ai) u = c + 1; else v = c; }
    if (ai < (uint32_t)a[b[u]]) { if (u > 0) p[i] = b[u - 1]; b[u] = i; }
  }
  for (uint32_t u = nb, v = b[nb - 1]; u--; v = p[v]) b[u] = v;
  return nb;
}

// One block (64 threads = one wave) per read, persistent.  Dynamic LDS layout (bytes), ML = max_len rounded:
//   rdq[ML] | rfq[ML+2*edges_max+64] | bound[2*(ML+...)] ints | keys[CH_KEYS_LDS] u64 | pairs[CH_PAIRS_LDS] u64 |
//   lis[2*CH_PAIRS_LDS] u32 | hits[CH_HITS_LDS] uint2
__global__ void __launch_bounds__(64) k_chain(DReads rd, DIndex ix, DParams P, int pass, int is_last_strand,
                                              RState* __restrict__ work, AlignRec* __restrict__ work_aln, RWork* __restrict__ rw,
                                              const uint32_t* __restrict__ pool, unsigned long long* __restrict__ ctr,
                                              uint32_t* g_cnt, unsigned long long* g_keys, unsigned long long* g_pairs, uint32_t* g_lis,
                                              uint2* g_hits, uint32_t keys_cap, uint32_t pairs_cap, uint32_t hits_cap,
                                              uint32_t lds_ml, uint32_t lds_rf) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  __shared__ uint32_t s_next;
  __shared__ uint32_t s_ncand;
  const int lane = lane_id();
  uint8_t* rdq = lds_raw;
  uint8_t* rfq = rdq + lds_ml;
  int* bound = (int*)(rfq + lds_rf);
  unsigned long long* l_keys = (unsigned long long*)(bound + 2 * lds_rf);
  unsigned long long* l_pairs = l_keys + CH_KEYS_LDS;
  uint32_t* l_lis = (uint32_t*)(l_pairs + CH_PAIRS_LDS);
  uint2* l_hits = (uint2*)(l_lis + 2 * CH_PAIRS_LDS);

  uint32_t* cnt = g_cnt + (size_t)blockIdx.x * ix.n_refs;
  unsigned long long* gk = g_keys + (size_t)blockIdx.x * keys_cap;
  unsigned long long* gp = g_pairs + (size_t)blockIdx.x * pairs_cap;
  uint32_t* gl = g_lis + (size_t)blockIdx.x * 2 * pairs_cap;
  uint2* gh = g_hits + (size_t)blockIdx.x * hits_cap;

  for (;;) {
    __syncthreads();
    if (lane == 0) s_next = (uint32_t)atomicAdd(&ctr[C_WORK_NEXT], 1ull);
    __syncthreads();
    const uint32_t r = s_next;
    if (r >= rd.n) break;
    RWork w = rw[r];
    if (!(w.strand_active && w.search && w.pass_n == (uint32_t)pass)) continue;
    RState st = work[r];
    const uint32_t len = rd.len[r];
    const uint32_t* rec = rd.words + rd.rec_off[r];
    int search = 1;
    const uint32_t max_SW_score = len * (uint32_t)P.match;

    if (st.hit_seeds >= (uint32_t)P.num_seeds && w.hit_total > 0) {
      // ---------------- compute_lis_alignment (alignment.cpp:100-509) ----------------
      // gather this strand's hits (all passes so far) into a flat array
      const uint32_t nh = w.hit_total;
      uint2* hits = nh <= CH_HITS_LDS ? l_hits : gh;
      bool cap_err = nh > hits_cap && nh > CH_HITS_LDS;
      if (cap_err) { if (lane == 0) atomicAdd(&ctr[C_ERR_PAIRS], 1ull); }
      else {
        uint32_t o = 0;
        for (uint32_t seg = w.hit_head; seg != NONE;) {
          uint32_t nxt = pool[seg], c = pool[seg + 1];
          for (uint32_t q = lane; q < c; q += 64) hits[o + q] = make_uint2(pool[seg + 2 + 2 * q], pool[seg + 3 + 2 * q]);
          o += c; seg = nxt;
        }
        __syncthreads();
        // 1. per-reference histogram of seed hits (:117-130) with device atomics on this slot's private counters
        if (lane == 0) s_ncand = 0;
        __syncthreads();
        for (uint32_t hb = 0; hb < nh; hb += 64) {
          uint32_t h = hb + lane;
          uint32_t lo = 0, hi = 0;
          if (h < nh) { uint32_t id = hits[h].x; lo = ix.pos_off[id]; hi = ix.pos_off[id + 1]; }
          // short lists: lane-serial; long lists: the whole wave walks them together
          bool lng = (hi - lo) > 32;
          if (!lng) {
            for (uint32_t k = lo; k < hi; k++) {
              uint32_t seq = ix.pos_arr[k].y;
              uint32_t old = atomicAdd(&cnt[seq], 1u);
              if (old + 1 == (uint32_t)P.num_seeds) { uint32_t sl = atomicAdd(&s_ncand, 1u); if (sl < keys_cap) gk[sl] = seq; }
            }
          }
          unsigned long long lm = __ballot(lng);
          while (lm) {
            int src = __ffsll((long long)lm) - 1; lm &= lm - 1;
            uint32_t llo = __shfl(lo, src, 64), lhi = __shfl(hi, src, 64);
            for (uint32_t k = llo + lane; k < lhi; k += 64) {
              uint32_t seq = ix.pos_arr[k].y;
              uint32_t old = atomicAdd(&cnt[seq], 1u);
              if (old + 1 == (uint32_t)P.num_seeds) { uint32_t sl = atomicAdd(&s_ncand, 1u); if (sl < keys_cap) gk[sl] = seq; }
            }
          }
        }
        __threadfence_block();
        __syncthreads();
        uint32_t ncand = s_ncand;
        if (ncand > keys_cap) { if (lane == 0) atomicAdd(&ctr[C_ERR_PAIRS], 1ull); ncand = 0; cap_err = true; }
        unsigned long long* keys = ncand <= CH_KEYS_LDS ? l_keys : gk;
        // key = (~count, ref): ascending order == count desc, ref asc (:134-148)
        for (uint32_t c = lane; c < ncand; c += 64) {
          uint32_t ref = (uint32_t)gk[c];
          uint32_t count = __hip_atomic_load(&cnt[ref], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          keys[c] = ((unsigned long long)(0xFFFFFFFFu - count) << 32) | ref;
        }
        __syncthreads();
        // clear the counters (walk again)
        for (uint32_t hb = 0; hb < nh; hb += 64) {
          uint32_t h = hb + lane;
          uint32_t lo = 0, hi = 0;
          if (h < nh) { uint32_t id = hits[h].x; lo = ix.pos_off[id]; hi = ix.pos_off[id + 1]; }
          bool lng = (hi - lo) > 32;
          if (!lng) for (uint32_t k = lo; k < hi; k++) cnt[ix.pos_arr[k].y] = 0;
          unsigned long long lm = __ballot(lng);
          while (lm) {
            int src = __ffsll((long long)lm) - 1; lm &= lm - 1;
            uint32_t llo = __shfl(lo, src, 64), lhi = __shfl(hi, src, 64);
            for (uint32_t k = llo + lane; k < lhi; k += 64) cnt[ix.pos_arr[k].y] = 0;
          }
        }
        __syncthreads();
        if (ncand > 1) wave_sort_u64(keys, ncand);
        __syncthreads();

        // 2. candidate loop (:150-508)
        int is_aligned = 0;
        int is_search_candidates = 1;
        for (uint32_t k = 0; k < ncand && is_search_candidates && !cap_err; k++) {
          const unsigned long long ck = keys[k];
          const uint32_t max_ref = (uint32_t)ck;
          const uint32_t max_occur = 0xFFFFFFFFu - (uint32_t)(ck >> 32);
          if (max_occur < (uint32_t)P.num_seeds) break;
          if (is_aligned && P.min_lis > 0 && k > 0 && max_occur < (0xFFFFFFFFu - (uint32_t)(keys[k - 1] >> 32))) {   // :165-169
            --w.best;
            if (w.best < 1) break;
          }
          // 3. hits on this reference (:181-201): each lane binary-searches one hit's (seq-sorted) position list
          uint32_t np = 0;
          for (int phase = 0; phase < 2; phase++) {
            // phase 0 counts, phase 1 writes at deterministic offsets
            uint32_t run = 0;
            unsigned long long* pairs = np <= CH_PAIRS_LDS ? l_pairs : gp;
            for (uint32_t hb = 0; hb < nh; hb += 64) {
              uint32_t h = hb + lane;
              uint32_t first = 0, cntm = 0, win = 0;
              if (h < nh) {
                uint32_t id = hits[h].x; win = hits[h].y;
                uint32_t lo = ix.pos_off[id], hi = ix.pos_off[id + 1];
                uint32_t a = lo, b = hi;
                while (a < b) { uint32_t mid = (a + b) >> 1; if (ix.pos_arr[mid].y < max_ref) a = mid + 1; else b = mid; }
                first = a;
                uint32_t c2 = a, d2 = hi;
                while (c2 < d2) { uint32_t mid = (c2 + d2) >> 1; if (ix.pos_arr[mid].y <= max_ref) c2 = mid + 1; else d2 = mid; }
                cntm = c2 - a;
              }
              uint32_t tot; uint32_t ex = wave_excl_scan_u32(cntm, tot);
              if (phase == 1) for (uint32_t q = 0; q < cntm; q++) pairs[run + ex + q] = ((unsigned long long)ix.pos_arr[first + q].x << 32) | win;
              run += tot;
            }
            if (phase == 0) { np = run; if (np > pairs_cap && np > CH_PAIRS_LDS) { cap_err = true; break; } }
          }
          if (cap_err) { if (lane == 0) atomicAdd(&ctr[C_ERR_PAIRS], 1ull); break; }
          unsigned long long* pairs = np <= CH_PAIRS_LDS ? l_pairs : gp;
          uint32_t* lisb = np <= CH_PAIRS_LDS ? l_lis : gl;
          uint32_t* lisp = lisb + (np <= CH_PAIRS_LDS ? CH_PAIRS_LDS : pairs_cap);
          __syncthreads();
          if (np > 1) wave_sort_u64(pairs, np);
          __syncthreads();
          // 4. sliding window of read length along the reference (:203-506)
          uint32_t it = 0, ms_lo = 0, ms_hi = 0;
          uint32_t begin_ref = (uint32_t)(pairs[0] >> 32), begin_read = (uint32_t)pairs[0];
          const uint64_t reflen = ix.ref_off[max_ref + 1] - ix.ref_off[max_ref];
          while (it != np && is_search_candidates) {
            const uint64_t end_ref_max = (uint64_t)begin_ref + len - begin_read - P.lnwin + 1;
            int push = 0;
            while (it != np && (uint64_t)(uint32_t)(pairs[it] >> 32) <= end_ref_max) { ms_hi = ++it; push = 1; }
            int skip_to_pop = 0;
            if (!push && is_aligned) skip_to_pop = 1;        // heuristic 1 (:243-246)
            else is_aligned = 0;
            if (!skip_to_pop && (ms_hi - ms_lo) >= (uint32_t)P.num_seeds) {
              uint32_t nl = find_lis_dev(pairs + ms_lo, ms_hi - ms_lo, lisb, lisp);
              if (nl >= (uint32_t)P.min_lis) {
                const uint32_t lcs_ref_start = (uint32_t)(pairs[ms_lo + lisb[0]] >> 32);
                const uint32_t lcs_que_start = (uint32_t)pairs[ms_lo + lisb[0]];
                uint64_t head = 0, tail = 0, align_ref_start = 0, align_que_start = 0, align_length = 0;
                const uint64_t rlen = len;
                uint32_t edges;
                if (P.is_as_percent) edges = (uint32_t)((P.edges / 100.0) * (double)rlen);
                else edges = (uint32_t)P.edges;
                if (lcs_ref_start < lcs_que_start) {                         // :287-325
                  align_ref_start = 0; align_que_start = lcs_que_start - lcs_ref_start; head = 0;
                  if (reflen < rlen) {
                    tail = 0;
                    if (align_que_start > (rlen - reflen)) align_length = reflen - (align_que_start - (rlen - reflen));
                    else align_length = reflen;
                  } else {
                    tail = reflen - align_ref_start - rlen;
                    if (tail > (uint64_t)(uint32_t)(edges - 1)) tail = edges;
                    align_length = rlen + head + tail - align_que_start;
                  }
                } else {                                                     // :326-357
                  align_ref_start = lcs_ref_start - lcs_que_start; align_que_start = 0;
                  if (align_ref_start > (uint64_t)(uint32_t)(edges - 1)) head = edges;
                  if (align_ref_start + rlen > reflen) { tail = 0; align_length = reflen - align_ref_start - head; }
                  else {
                    tail = reflen - align_ref_start - rlen;
                    if (tail > (uint64_t)(uint32_t)(edges - 1)) tail = edges;
                    align_length = rlen + head + tail;
                  }
                }
                // read.flip34() to the 0..4 alphabet before SSW (:360-361)
                // (is03/is04 only toggle when the read has ambiguous letters; aval tracks the stored value)
                if (w.has_amb && !w.is04) { w.is04 = 1; w.aval = 4; }
                const int m = (int)(align_length - head - tail);
                const int nref = (int)align_length;
                const uint64_t rf_start = ix.ref_off[max_ref] + align_ref_start - head;
                SwRes fw; fw.score = 0; fw.end_ref = -1; fw.end_read = m - 1;
                bool sw_ok = (m > 0 && nref > 0 && (uint32_t)m <= lds_ml && (uint32_t)nref <= lds_rf);
                if (!sw_ok && (m > 0 && nref > 0)) { if (lane == 0) atomicAdd(&ctr[C_ERR_PAIRS], 1ull); cap_err = true; }
                if (sw_ok) {
                  for (int q = lane; q < m; q += 64) rdq[q] = (uint8_t)read_nt(rec, len, (uint32_t)align_que_start + q, w.reversed, w.aval);
                  for (int q = lane; q < nref; q += 64) rfq[q] = ix.ref_seq[rf_start + q];
                  __syncthreads();
                  fw = sw_wave(rdq, m, 0, 1, rfq, nref, 0, 1, bound, P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext);
                  if (lane == 0) { atomicAdd(&ctr[C_SW_FWD], 1ull); atomicAdd(&ctr[C_SW_CELLS], (unsigned long long)m * nref); }
                }
                int score1 = fw.score > 65535 ? 65535 : fw.score;
                int ref_begin1 = -1, read_begin1 = -1;
                const int ref_end1 = fw.end_ref, read_end1 = fw.end_read;
                if (sw_ok && (uint32_t)score1 >= (P.minimal_score & 0xFFFFu)) {   // ssw_align: flag==2 && score1 < filters -> no begin
                  // reverse pass (ssw.c:900-918) on the prefixes ending at (read_end1, ref_end1)
                  SwRes bw = sw_wave(rdq, read_end1 + 1, read_end1, -1, rfq, ref_end1 + 1, ref_end1, -1, bound,
                                     P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext);
                  ref_begin1 = ref_end1 - bw.end_ref;
                  read_begin1 = read_end1 - bw.end_read;
                  if (lane == 0) { atomicAdd(&ctr[C_SW_REV], 1ull); atomicAdd(&ctr[C_SW_CELLS], (unsigned long long)(read_end1 + 1) * (ref_end1 + 1)); }
                }
                is_aligned = (sw_ok && (uint32_t)score1 > P.minimal_score);     // strict (:388)
                if (is_aligned) {
                  if ((uint32_t)score1 == max_SW_score) ++st.max_SW_count;
                  AlignRec al;
                  al.ref_begin1 = ref_begin1 + (int32_t)(align_ref_start - head);
                  al.ref_end1 = ref_end1 + (int32_t)(align_ref_start - head);
                  al.read_begin1 = read_begin1 + (int32_t)align_que_start;
                  al.read_end1 = read_end1 + (int32_t)align_que_start;
                  al.readlen = len; al.ref_num = max_ref;
                  al.index_num = (uint16_t)P.index_num; al.part = (uint16_t)P.part;
                  al.strand = (uint8_t)!w.reversed; al.score1 = (uint16_t)score1;
                  al.has_cigar = 0; al.cigar_off = 0; al.cigar_len = 0;
                  AlignRec* slots = work_aln + (size_t)r * P.slots;
                  if (!st.is_hit) {                                              // :411-416
                    st.is_hit = 1;
                    if (lane == 0) { atomicAdd(&ctr[C_NUM_ALIGNED], 1ull); atomicAdd(&ctr[C_PER_DB + P.index_num], 1ull); }
                  }
                  if (P.num_alignments == 0 || !P.is_best || (P.is_best && st.n_align < P.num_alignments)) {
                    if (st.n_align < P.slots) { if (lane == 0) slots[st.n_align] = al; st.n_align++; w.is_new_hit = 1; }
                    else { if (lane == 0) atomicAdd(&ctr[C_ERR_SLOTS], 1ull); }
                  } else if (P.is_best && st.n_align == P.num_alignments) {
                    __syncthreads();
                    if (slots[st.min_index].score1 < (uint16_t)score1) {         // :425-459
                      if (P.num_alignments > 1 && st.max_index == 0 && st.min_index == 0) {
                        uint32_t mn = 0, mx = 0;
                        for (uint32_t q = 1; q < st.n_align; q++) { if (slots[q].score1 < slots[mn].score1) mn = q; if (slots[q].score1 > slots[mx].score1) mx = q; }
                        st.min_index = mn; st.max_index = mx;
                      }
                      const uint32_t mn = st.min_index, mx = st.max_index;
                      const uint16_t mx_score = slots[mx].score1;
                      __syncthreads();
                      if (lane == 0) slots[mn] = al;
                      __threadfence_block();
                      __syncthreads();
                      w.is_new_hit = 1;
                      if ((uint16_t)score1 > (mn == mx ? (uint16_t)score1 : mx_score) && st.n_align > 1) {
                        st.max_index = mn;
                        uint32_t m2 = 0;
                        for (uint32_t q = 1; q < st.n_align; q++) if (slots[q].score1 < slots[m2].score1) m2 = q;
                        st.min_index = m2;
                      }
                      // :454-457 decrement/increment of reads_matched_per_db cancel (both use the NEW alignment's index)
                    }
                  }
                  __syncthreads();
                  if (P.num_alignments > 0) {                                    // :462-469
                    if (P.is_best) { if (P.num_alignments == st.max_SW_count) is_search_candidates = 0; }
                    else if (P.num_alignments == st.n_align) is_search_candidates = 0;
                  }
                  search = 0;
                }
              }
            }
            // pop (:486-506)
            if (ms_hi > ms_lo) ms_lo++;
            if (ms_hi == ms_lo) {
              if (it != np) { begin_ref = (uint32_t)(pairs[it] >> 32); begin_read = (uint32_t)pairs[it]; }
              else break;
            } else { begin_ref = (uint32_t)(pairs[ms_lo] >> 32); begin_read = (uint32_t)pairs[ms_lo]; }
          }
          __syncthreads();
        }
      }
    }
    // ---------------- pass control (paralleltraversal.cpp:253-277) ----------------
    uint32_t pass_n = w.pass_n;
    if (search) {
      if (pass_n == 2) search = 0;
      else {
        while (pass_n < 2 && P.skip[pass_n] == P.skip[pass_n + 1]) ++pass_n;
        if (++pass_n > 2) search = 0;
        else w.win_shift = P.skip[pass_n];
      }
    }
    w.pass_n = (uint8_t)pass_n; w.search = (uint8_t)search;
    if (!search) {
      // end of traverse() (:279-297)
      st.lastIndex = P.index_num; st.lastPart = P.part;
      if (P.num_alignments > 0) {
        if ((P.is_best && P.num_alignments == st.max_SW_count) || (!P.is_best && st.n_align == P.num_alignments)) st.is_done = 1;
      } else if (P.is_last_index_part && is_last_strand && st.n_align > 0) st.is_done = 1;
      w.strand_active = 0;
    }
    if (lane == 0) { work[r] = st; rw[r] = w; }
  }
}

// ------------------------------------------------------------------------------------------------
// k_trace: banded_sw (ssw.c:577-773), one thread per stored alignment.  Scratch per wave slot:
//   dir  : bytes, element e of lane l at dir[e*64 + l]
//   hbuf : ints,  3 arrays (h_b, e_b, h_c) of `wcap` ints, element e of lane l at (a*wcap + e)*64 + l
// ------------------------------------------------------------------------------------------------
#define TR_SET_U(u, w, i, j) { int x_ = (i) - (w); x_ = x_ > 0 ? x_ : 0; (u) = (j) - x_ + 1; }
#define TR_SET_D(u, w, i, j, p) { int x_ = (i) - (w); x_ = x_ > 0 ? x_ : 0; x_ = (j) - x_; (u) = x_ * 3 + (p); }

__global__ void __launch_bounds__(64) k_trace(DReads rd, DIndex ix, DParams P, const uint32_t* __restrict__ tasks, uint32_t n_tasks,
                                              AlignRec* __restrict__ aln, uint32_t* __restrict__ cigar_pool, uint32_t cigar_words,
                                              unsigned long long* __restrict__ ctr, int8_t* g_dir, int* g_hbuf, uint32_t* g_cig,
                                              uint64_t dir_cap, uint32_t wcap, uint32_t cig_cap) {
  const int lane = lane_id();
  int8_t* dir = g_dir + (size_t)blockIdx.x * dir_cap * 64;
  int* hb = g_hbuf + (size_t)blockIdx.x * 3 * wcap * 64;
  uint32_t* cg = g_cig + (size_t)blockIdx.x * cig_cap * 64;
#define DIRX(e) dir[(size_t)(e) * 64 + lane]
#define HB(a, e) hb[((size_t)(a) * wcap + (e)) * 64 + lane]
#define CG(e) cg[(size_t)(e) * 64 + lane]
  for (uint32_t tb = blockIdx.x * 64; tb < n_tasks; tb += gridDim.x * 64) {
    uint32_t t = tb + lane;
    if (t >= n_tasks) continue;
    const uint32_t slot = tasks[t];
    AlignRec al = aln[slot];
    const uint32_t r = slot / P.slots;
    const uint32_t len = rd.len[r];
    const uint32_t* rec = rd.words + rd.rec_off[r];
    const uint32_t reversed = al.strand ? 0u : 1u;
    const uint8_t* ref = ix.ref_seq + ix.ref_off[al.ref_num] + al.ref_begin1;
    const int refLen = al.ref_end1 - al.ref_begin1 + 1;
    const int readLen = al.read_end1 - al.read_begin1 + 1;
    const int score = al.score1;
    const int gapO = P.gap_open, gapE = P.gap_ext;
    int band_width = abs(refLen - readLen) + 1;
    int i, j, e, f, temp1, temp2, l, mx = 0, width, width_d;
    bool fail = false;
    size_t dl = 0;                                         // direction_line offset (elements)
    // h_b / e_b / h_c start zeroed (calloc-like); stale values across band doublings are kept like the reference
    for (uint32_t q = 0; q < wcap; q++) { HB(0, q) = 0; HB(1, q) = 0; HB(2, q) = 0; }
    do {
      width = band_width * 2 + 3; width_d = band_width * 2 + 1;
      if ((uint32_t)width + 2 > wcap || (uint64_t)width_d * readLen * 3 + 8 > dir_cap) { fail = true; break; }
      for (j = 1; j < width - 1; j++) HB(0, j) = 0;
      for (i = 0; i < readLen; i++) {
        int beg = 0, end = refLen - 1, u = 0, edge;
        j = i - band_width; beg = beg > j ? beg : j;
        j = i + band_width; end = end < j ? end : j;
        edge = end + 1 < width - 1 ? end + 1 : width - 1;
        f = 0; HB(0, 0) = 0; HB(1, 0) = 0; HB(0, edge) = 0; HB(1, edge) = 0; HB(2, 0) = 0;
        dl = (size_t)width_d * i * 3;
        const int rnt = (int)read_nt(rec, len, (uint32_t)(al.read_begin1 + i), reversed, 4u);
        for (j = beg; j <= end; j++) {
          int b, e1, f1, d, de, df, dh;
          TR_SET_U(u, band_width, i, j); TR_SET_U(e, band_width, i - 1, j);
          TR_SET_U(b, band_width, i, j - 1); TR_SET_U(d, band_width, i - 1, j - 1);
          TR_SET_D(de, band_width, i, j, 0);
          TR_SET_D(df, band_width, i, j, 1);
          TR_SET_D(dh, band_width, i, j, 2);
          temp1 = i == 0 ? -gapO : HB(0, e) - gapO;
          temp2 = i == 0 ? -gapE : HB(1, e) - gapE;
          int eb = temp1 > temp2 ? temp1 : temp2;
          HB(1, u) = eb;
          int8_t dde = temp1 > temp2 ? 3 : 2;
          DIRX(dl + de) = dde;
          temp1 = HB(2, b) - gapO;
          temp2 = f - gapE;
          f = temp1 > temp2 ? temp1 : temp2;
          int8_t ddf = temp1 > temp2 ? 5 : 4;
          DIRX(dl + df) = ddf;
          e1 = eb > 0 ? eb : 0;
          f1 = f > 0 ? f : 0;
          temp1 = e1 > f1 ? e1 : f1;
          const int fnt = ref[j];
          const int sc = (fnt == 4 || rnt == 4) ? P.score_N : (fnt == rnt ? P.match : P.mismatch);
          temp2 = HB(0, d) + sc;
          int hc = temp1 > temp2 ? temp1 : temp2;
          HB(2, u) = hc;
          if (hc > mx) mx = hc;
          if (temp1 <= temp2) DIRX(dl + dh) = 1;
          else DIRX(dl + dh) = e1 > f1 ? dde : ddf;
        }
        for (j = 1; j <= u; j++) HB(0, j) = HB(2, j);
      }
      band_width *= 2;
    } while (mx < score);
    uint32_t clen = 0, coff = 0;
    if (!fail) {
      band_width /= 2;
      // trace back (ssw.c:674-747); dl points at the last row
      i = readLen - 1; j = refLen - 1; e = 0; l = 0; f = 0; mx = 0; temp2 = 2;
      while (i > 0) {
        if (j < 0) { fail = true; break; }
        TR_SET_D(temp1, band_width, i, j, temp2);
        if (temp1 < 0 || temp1 >= width_d * 3) { fail = true; break; }
        int dv = DIRX(dl + temp1);
        switch (dv) {
          case 1: --i; --j; temp2 = 2; dl -= (size_t)width_d * 3; f = 0; break;
          case 2: --i; temp2 = 0; dl -= (size_t)width_d * 3; f = 1; break;
          case 3: --i; temp2 = 2; dl -= (size_t)width_d * 3; f = 1; break;
          case 4: --j; temp2 = 1; f = 2; break;
          case 5: --j; temp2 = 2; f = 2; break;
          default: fail = true; i = 0; break;
        }
        if (fail) break;
        if (f == mx) ++e;
        else {
          ++l;
          if ((uint32_t)l + 3 >= cig_cap) { fail = true; break; }
          CG(l - 1) = (uint32_t)e << 4 | (uint32_t)mx;
          mx = f; e = 1;
        }
      }
      if (!fail) {
        if ((uint32_t)l + 3 >= cig_cap) fail = true;
        else if (f == 0) { ++l; CG(l - 1) = (uint32_t)(e + 1) << 4; }
        else { l += 2; CG(l - 2) = (uint32_t)e << 4 | (uint32_t)f; CG(l - 1) = 16; }
      }
      if (!fail) {
        clen = (uint32_t)l;
        unsigned long long old = atomicAdd(&ctr[C_CIGAR_CURSOR], (unsigned long long)clen);
        if (old + clen > cigar_words) { atomicAdd(&ctr[C_ERR_CIGAR], 1ull); fail = true; }
        else { coff = (uint32_t)old; for (uint32_t q = 0; q < clen; q++) cigar_pool[coff + q] = CG(clen - 1 - q); }
      }
    }
    if (fail) { atomicAdd(&ctr[C_ERR_TRACE], 1ull); }
    else { al.has_cigar = 1; al.cigar_off = coff; al.cigar_len = clen; aln[slot] = al; }
  }
#undef DIRX
#undef HB
#undef CG
}

}  // namespace smr
