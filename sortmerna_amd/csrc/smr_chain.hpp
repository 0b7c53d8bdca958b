// smr_chain.hpp -- part of the HIP kernels of libsmr_hip (included by smr_kernels.hpp).
#pragma once

namespace smr {

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

__device__ __forceinline__ uint32_t pow2_floor(uint32_t v) { return v ? 1u << (31 - __clz((int)v)) : 0u; }

struct SwRes { int score, end_ref, end_read, word; };      // word: set by the striped slow path only (smr_sw_striped.hpp: the 16-bit kernel produced the result)

// Smith-Waterman score + end cell as an anti-diagonal systolic array over the wave.  Lane i owns R consecutive read
// rows (R = ceil(m/64) <= 4: a 150-nt read is ONE strip of 64 x 3 rows; longer reads take several strips with an LDS
// boundary row in between); at step t lane i computes its R cells of reference column t - i.  Everything that moves
// between lanes -- the last row's H and F and the column's nucleotide -- moves by one lane per step with a DPP
// wave_shr:1 (a register move, no LDS round trip); lane 0's inputs (boundary H/F of the previous strip, next
// reference nucleotide) are injected by the same instruction and prefetched one step ahead.
// Same H as the reference's striped SSE2 kernels (ssw.c:150-575) under the affine model H = max(0, diag+s, E, F),
// first gap base costs gap_open, further bases gap_ext; end cell = first column (in processing order) where the
// maximum is first reached, smallest row in that column (ssw.c:305-336).  The reverse pass (ssw.c:900-918) runs the
// same recurrence on the reversed prefixes (strides -1); since its maximum equals the forward score, stopping at
// `terminate` selects the same cell.  rdq: read in 0..4 alphabet (LDS), rfq: reference window (LDS), bound: 2*n ints.
__device__ __forceinline__ int dpp_shr1(int inject_lane0, int v) {
  return __builtin_amdgcn_update_dpp(inject_lane0, v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}

// PACKED: scores fit 14 bits and columns 16 bits -> the per-lane running maximum is one v_max_u32 over
// (h << 18 | (0xFFFF - col) << 2 | (3 - r)) per cell (max h, then first column, then smallest row), merged per strip.
template <int R, bool PACKED>
__device__ __forceinline__ SwRes sw_wave_r(const uint8_t* rdq, int m, int rd0, int rdstep, const uint8_t* rfq, int n, int rf0, int rfstep,
                                           int* bound, int match, int mismatch, int scoreN, int go, int ge) {
  const int lane = lane_id();
  int bestH = 0, bestcol = 0x1FFFFF, bestrow = 0x1FFFFF;
  const int rps = 64 * R;
  const int nstrips = (m + rps - 1) / rps;
  for (int s = 0; s < nstrips; s++) {
    const int row0 = s * rps + lane * R;
    // per row: its score against reference nucleotides 0..3 as 4 signed bytes (N in the reference is handled per step)
    int H[R], E[R];
    uint32_t sct[R];
    bool vr[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      vr[r] = row0 + r < m;
      const int c = vr[r] ? rdq[rd0 + rdstep * (row0 + r)] : 4;
      uint32_t t = 0;
      for (int b = 0; b < 4; b++) t |= (uint32_t)((c == 4 ? scoreN : (c == b ? match : mismatch)) & 0xFF) << (8 * b);
      sct[r] = t; H[r] = 0; E[r] = 0;
    }
    uint32_t bkey = 0;
    int lastH = 0, lastF = 0, Hdiag0 = 0, fnt = 4;
    const bool has_prev = s > 0, has_next = s + 1 < nstrips;
    const int steps = n + 63;
    for (int t0 = 0; t0 < steps; t0 += 64) {
      // lane 0's inputs of the next 64 steps (reference letter; boundary H / F of the previous strip), one column per lane, read back with
      // v_readlane: the step loop has no memory access (the boundary rows live in global memory)
      const int cq = t0 + lane;
      int chC = 4, chH = 0, chF = 0;
      if (cq < n) {
        chC = rfq[rf0 + rfstep * cq];
        if (has_prev) { chH = bound[2 * cq]; chF = bound[2 * cq + 1]; }
      }
      const int tend = min(64, steps - t0);
      for (int tt = 0; tt < tend; tt++) {
        const int t = t0 + tt;
        const int cH = __builtin_amdgcn_readlane(chH, tt), cF = __builtin_amdgcn_readlane(chF, tt), cc = __builtin_amdgcn_readlane(chC, tt);
        const int upH = dpp_shr1(cH, lastH), upF = dpp_shr1(cF, lastF);
        fnt = dpp_shr1(cc, fnt);
        const int col = t - lane;
        const bool colok = col >= 0 && col < n;
        const bool refN = fnt == 4;
        const int sh = (fnt & 3) * 8;
        const uint32_t colkey = (uint32_t)(0xFFFF - col) << 2;
        int diag = Hdiag0, uh = upH, uf = upF;
#pragma unroll
        for (int r = 0; r < R; r++) {
          const int sc = refN ? scoreN : (int)__builtin_amdgcn_sbfe(sct[r], sh, 8);
          const int e = max(E[r] - ge, H[r] - go);
          const int f = max(uf - ge, uh - go);
          const int h = max(max(diag + sc, e), max(f, 0));
          diag = H[r];                                         // H(row, col-1) is the diagonal of the next row
          if (colok && vr[r]) {
            H[r] = h; E[r] = e;
            if (PACKED) bkey = max(bkey, ((uint32_t)h << 18) | colkey | (uint32_t)(3 - r));
            else if (h > bestH || (h == bestH && h > 0 && col < bestcol)) { bestH = h; bestcol = col; bestrow = row0 + r; }
          }
          uh = h; uf = f;
        }
        Hdiag0 = col >= 0 ? upH : 0;                           // H(row0-1, col): diagonal of row0 at the next column
        if (colok) { lastH = H[R - 1]; lastF = uf; }
        if (has_next && lane == 63 && colok) { bound[2 * col] = lastH; bound[2 * col + 1] = lastF; }
      }
    }
    if (PACKED) {
      const int h = (int)(bkey >> 18), col = 0xFFFF - (int)((bkey >> 2) & 0xFFFF), row = row0 + 3 - (int)(bkey & 3);
      if (h > bestH || (h == bestH && h > 0 && col < bestcol)) { bestH = h; bestcol = col; bestrow = row; }
    }
    __syncthreads();
  }
  // lexicographic reduce: max H, then min col, then min row
  unsigned long long key = bestH > 0 ? (((unsigned long long)bestH << 42) | ((unsigned long long)(0x1FFFFF - bestcol) << 21) |
                                        (unsigned long long)(0x1FFFFF - bestrow)) : 0ull;
  key = wave_max_u64(key);
  SwRes rr;
  if (key == 0) { rr.score = 0; rr.end_ref = -1; rr.end_read = m - 1; return rr; }
  rr.score = (int)(key >> 42);
  rr.end_ref = 0x1FFFFF - (int)((key >> 21) & 0x1FFFFF);
  rr.end_read = 0x1FFFFF - (int)(key & 0x1FFFFF);
  return rr;
}

}  // namespace smr
#include "smr_sw_pk.hpp"
namespace smr {

// reads of more than 512 letters (mode 2): strips of 128 virtual lanes x R rows.  A step of a strip is 14 R + 35 vector instructions (counted in the
// disassembly: R = 8: 147, R = 16: 259 -- the hand-over between lanes, the inputs of lane 0 and the boundary row for the next strip are paid per
// 128 R cells), and a read of m rows takes ceil(m / 128 R) strips of n + 127 steps: what a column costs is strips x (14 R + 35), least when the strips
// are few AND the last one is full.  Rounds 3 - 5 chose between R = 8 and 16 with a cost model from before the kernel was tuned (9 R + 78), which
// sent 5 kb reads to R = 16 -- three strips, the third 44 % full: 777 per column against 735 with R = 8; with R = 20 it is two strips and 630.  Now the
// cheapest of seven strip heights by the counted cost (mean over N(5000, 500) reads: 771 -> 661 per column).  Functions of their own so that their
// registers (up to 20 rows of state per lane) are not the footprint of every other call of sw_wave.
template <int R>
__device__ __attribute__((noinline)) SwRes sw_wave_long_r(const uint8_t* rdq, int m, int rd0, int rdstep, const uint8_t* rfq, int n, int rf0, int rfstep,
                                                int* bound, int match, int mismatch, int scoreN, int go, int ge, bool hn) {
  if (hn) return sw_wave_pk_r<R, true, true>(rdq, m, rd0, rdstep, rfq, n, rf0, rfstep, bound, match, mismatch, scoreN, go, ge);
  return sw_wave_pk_r<R, false, true>(rdq, m, rd0, rdstep, rfq, n, rf0, rfstep, bound, match, mismatch, scoreN, go, ge);
}
#ifndef SW_LONG_RMAX
#define SW_LONG_RMAX 24
#endif
__host__ __device__ __forceinline__ int sw_long_cost(int m, int R) { return ((m + 128 * R - 1) / (128 * R)) * (14 * R + 35); }
__host__ __device__ __forceinline__ int sw_long_rows(int m) {       // the strip height a read of m rows is scored with
  int best = 8, bc = sw_long_cost(m, 8);
  for (int R = 10; R <= SW_LONG_RMAX; R += 2) { const int c = sw_long_cost(m, R); if (c < bc) { bc = c; best = R; } }
  return best;
}
__device__ __forceinline__ SwRes sw_wave_long(const uint8_t* rdq, int m, int rd0, int rdstep, const uint8_t* rfq, int n, int rf0, int rfstep,
                                              int* bound, int match, int mismatch, int scoreN, int go, int ge, bool hn) {
#define SW_LONG_ARGS rdq, m, rd0, rdstep, rfq, n, rf0, rfstep, bound, match, mismatch, scoreN, go, ge, hn
  switch (sw_long_rows(m)) {
    case 10: return sw_wave_long_r<10>(SW_LONG_ARGS);
    case 12: return sw_wave_long_r<12>(SW_LONG_ARGS);
    case 14: return sw_wave_long_r<14>(SW_LONG_ARGS);
    case 16: return sw_wave_long_r<16>(SW_LONG_ARGS);
    case 18: return sw_wave_long_r<18>(SW_LONG_ARGS);
    case 20: return sw_wave_long_r<20>(SW_LONG_ARGS);
#if SW_LONG_RMAX >= 24
    case 22: return sw_wave_long_r<22>(SW_LONG_ARGS);
    case 24: return sw_wave_long_r<24>(SW_LONG_ARGS);
#endif
    default: return sw_wave_long_r<8>(SW_LONG_ARGS);
  }
#undef SW_LONG_ARGS
}

}  // namespace smr
#include "smr_sw_striped.hpp"
namespace smr {

// mode 1 / 2: the packed 16-bit kernel (smr_sw_pk.hpp; 2 = its wave_ror variant) where its preconditions hold; mode 0: always the 32-bit kernel
// mode < 0: the scheme is one under which ssw.c's striped kernels leave the affine recurrence -- the slow path that reproduces their stripe geometry
// (smr_sw_striped.hpp; terminate / word_in: the forward score and kernel, for the reverse pass; scr: the wave's scratch row)
// (STRIPED is a template parameter of the kernels that can take that path: the function needs 130 vector registers, which the fast instantiations must not carry)
template <bool STRIPED>
__device__ __attribute__((noinline)) SwRes sw_wave_t(const uint8_t* rdq, int m, int rd0, int rdstep, const uint8_t* rfq, int n, int rf0, int rfstep,
                                         int* bound, int match, int mismatch, int scoreN, int go, int ge, int mode, int terminate = 0, int word_in = 0, uint16_t* scr = nullptr) {
#define SW_ARGS rdq, m, rd0, rdstep, rfq, n, rf0, rfstep, bound, match, mismatch, scoreN, go, ge
  if (STRIPED) { int w = word_in; SwRes r = sw_wave_striped(rdq, m, rd0, rdstep, rfq, n, rf0, rfstep, match, mismatch, scoreN, go, ge, terminate, w, scr); r.word = w; return r; }
  if (mode >= 1 && (long long)m * match + 255 < 32768 && n + 128 <= 8191 && go + mismatch >= 0 && go + scoreN >= 0 && match + go <= 255 && scoreN + go <= 255) {
    bool hasn = false;
    for (int q = lane_id(); q < n; q += 64) hasn |= rfq[rf0 + rfstep * q] == 4;
    const bool hn = __any(hasn);
    // rows per virtual lane: 1 / 2 / 4 for reads up to 128 / 256 / 512 letters (one strip); longer reads take strips of 128 x 8 rows -- the
    // per-step overhead (hand-over between lanes, inputs of lane 0, the boundary row for the next strip) is paid per 1024 cells instead of 512
    if (mode == 2) {
      if (hn) { if (m <= 128) return sw_wave_pk_r<1, true, true>(SW_ARGS); if (m <= 256) return sw_wave_pk_r<2, true, true>(SW_ARGS); return sw_wave_pk_r<4, true, true>(SW_ARGS); }
      if (m <= 128) return sw_wave_pk_r<1, false, true>(SW_ARGS);
      if (m <= 256) return sw_wave_pk_r<2, false, true>(SW_ARGS);
      return sw_wave_pk_r<4, false, true>(SW_ARGS);
    }
    if (hn) { if (m <= 128) return sw_wave_pk_r<1, true, false>(SW_ARGS); if (m <= 256) return sw_wave_pk_r<2, true, false>(SW_ARGS); return sw_wave_pk_r<4, true, false>(SW_ARGS); }
    if (m <= 128) return sw_wave_pk_r<1, false, false>(SW_ARGS);
    if (m <= 256) return sw_wave_pk_r<2, false, false>(SW_ARGS);
    return sw_wave_pk_r<4, false, false>(SW_ARGS);
  }
  const bool small = (long long)m * match < 16384 && n < 65535 && match < 128 && mismatch > -128 && scoreN > -128 && scoreN < 128;
  if (small) {
    if (m <= 64) return sw_wave_r<1, true>(SW_ARGS);
    if (m <= 128) return sw_wave_r<2, true>(SW_ARGS);
    if (m <= 192) return sw_wave_r<3, true>(SW_ARGS);
    return sw_wave_r<4, true>(SW_ARGS);
  }
  return sw_wave_r<4, false>(SW_ARGS);
#undef SW_ARGS
}
__device__ __forceinline__ SwRes sw_wave(const uint8_t* rdq, int m, int rd0, int rdstep, const uint8_t* rfq, int n, int rf0, int rfstep,
                                         int* bound, int match, int mismatch, int scoreN, int go, int ge, int mode) {
  return sw_wave_t<false>(rdq, m, rd0, rdstep, rfq, n, rf0, rfstep, bound, match, mismatch, scoreN, go, ge, mode);
}

// a read of more than 512 letters through the 8-row strips where the packed kernel applies (the caller knows that its batch has such reads:
// the instantiations that never see one do not carry the call)
template <bool STRIPED>
__device__ __forceinline__ SwRes sw_wave_any_t(const uint8_t* rdq, int m, int rd0, int rdstep, const uint8_t* rfq, int n, int rf0, int rfstep,
                                             int* bound, int match, int mismatch, int scoreN, int go, int ge, int mode, int terminate = 0, int word_in = 0, uint16_t* scr = nullptr) {
  if (mode == 2 && m > 512 && (long long)m * match + 255 < 32768 && n + 128 <= 8191 && go + mismatch >= 0 && go + scoreN >= 0 && match + go <= 255 && scoreN + go <= 255) {
    bool hasn = false;
    for (int q = lane_id(); q < n; q += 64) hasn |= rfq[rf0 + rfstep * q] == 4;
    return sw_wave_long(rdq, m, rd0, rdstep, rfq, n, rf0, rfstep, bound, match, mismatch, scoreN, go, ge, __any(hasn));
  }
  return sw_wave_t<STRIPED>(rdq, m, rd0, rdstep, rfq, n, rf0, rfstep, bound, match, mismatch, scoreN, go, ge, mode, terminate, word_in, scr);
}

#define CH_EXT_CAP 65536u          // slots of the global candidate-set table of a block (tuples carry the slot in 16 bits)
#define CH_KEYS_LDS 128
#define CH_PAIRS_LDS 256
#define CH_HITS_LDS 128

// Longest strictly increasing subsequence of the read positions (low 32 bits) of a[0..n), n <= 64: its length and the index of the
// FIRST element of the particular subsequence that the reference's find_lis (alignment.cpp:58-98) reconstructs -- the only two things
// compute_lis_alignment uses (alignment.cpp:263-279).  The patience piles live across the lanes: lane c holds the smallest tail of an
// increasing run of length c+1 and the index of the element that run started with.  Placing element i is one ballot (how many tails are
// smaller = its pile) and two lane reads; the predecessor links of find_lis are not needed because the start of a run is inherited from
// the pile to the left at the moment the element is placed, exactly where find_lis records p[i] = b[u-1].
__device__ __forceinline__ uint32_t wave_lis_first(const unsigned long long* a, uint32_t n, uint32_t& first) {
  const int lane = lane_id();
  const uint32_t mine = (uint32_t)lane < n ? (uint32_t)a[lane] : 0u;
  uint32_t tail = 0, root = 0, nb = 0;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t ai = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)i);
    const uint32_t u = (uint32_t)__popcll(__ballot((uint32_t)lane < nb && tail < ai));          // tails are increasing: the piles with a smaller tail
    bool place = u == nb;
    if (!place) place = ai < (uint32_t)__builtin_amdgcn_readlane((int)tail, (int)u);              // equal: nothing changes (find_lis :87)
    if (place) {
      const uint32_t r = u > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)root, (int)(u - 1)) : i;
      if ((uint32_t)lane == u) { tail = ai; root = r; }
      if (u == nb) nb++;
    }
  }
  first = nb ? (uint32_t)__builtin_amdgcn_readlane((int)root, (int)(nb - 1)) : 0u;
  return nb;
}

// The same for n > 64 (more piles than lanes can occur): the textbook O(n log k) form with explicit pile and predecessor arrays
// b, p (2 x capacity n), executed uniformly by the wave; b[0] ends up as the first element of the subsequence.
__device__ uint32_t serial_lis_first(const unsigned long long* a, uint32_t n, uint32_t* b, uint32_t* p, uint32_t& first) {
  first = 0;
  if (n == 0) return 0;
  uint32_t nb = 0;
  for (uint32_t i = 0; i < n; i++) p[i] = 0;
  b[nb++] = 0;
  for (uint32_t i = 1; i < n; i++) {
    const uint32_t ai = (uint32_t)a[i];
    uint32_t lo = 0, hi = nb;                                  // first pile whose tail is >= ai
    while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if ((uint32_t)a[b[mid]] < ai) lo = mid + 1; else hi = mid; }
    if (lo == nb) { p[i] = b[nb - 1]; b[nb++] = i; }
    else if (ai < (uint32_t)a[b[lo]]) { if (lo > 0) p[i] = b[lo - 1]; b[lo] = i; }
  }
  uint32_t v = b[nb - 1];
  for (uint32_t c = nb; c-- > 1;) v = p[v];
  first = v;
  return nb;
}

// end of traverse() for one read: pass control (paralleltraversal.cpp:253-277), state write-back by the lane(s) with writer = true
// (PASS_ONLY: the caller changed nothing of the read's transient state itself -- k_cand --, so only the pass control fields are written back)
template <bool PASS_ONLY = false>
__device__ __forceinline__ void chain_finish_read(const DParams& P, int is_last_strand, uint32_t r, RState& st, RWork& w, int search, bool writer,
                                                  RState* __restrict__ work, RWork* __restrict__ rw) {
  uint32_t pass_n = w.pass_n;
  if (search) {
    // (no P.skip[pass_n]: an index the compiler cannot resolve puts the three strides into private memory)
    const uint32_t k0 = P.skip[0], k1 = P.skip[1], k2 = P.skip[2];
    if (pass_n == 2) search = 0;
    else {
      if (pass_n == 0 && k0 == k1) pass_n = 1;               // equal consecutive strides are skipped (:269-272)
      if (pass_n == 1 && k1 == k2) pass_n = 2;
      if (++pass_n > 2) search = 0;
      else w.win_shift = pass_n == 1 ? k1 : k2;
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
  if (writer) {
    work[r] = st;
    if (PASS_ONLY) { RWork* d = rw + r; d->win_shift = w.win_shift; d->pass_n = w.pass_n; d->search = w.search; d->strand_active = w.strand_active; }
    else rw[r] = w;
  }
}

// ------------------------------------------------------------------------------------------------
// k_cand: the cheap majority first.  Most reads that reach compute_lis_alignment (alignment.cpp:100-148) have no candidate reference
// at all: no reference sequence occurs twice among the positions of their seed hits.  That is decided here for every read of the
// (strand, pass) that has enough seed hits (phase 2; the others end their pass in phase 1, a lane each), 16 lanes per read and four reads per wave: the position lists of the read's hits are walked with one lane per
// POSITION and every reference number sets one bit of a Bloom bitmap in LDS (2 KB per read).  No bit set twice -> no reference seen
// twice -> no candidate (for num_seeds >= 2): the read's pass ends here, exactly as the candidate loop would end it without a single
// ssw_align (pass control + write-back by one lane).  Anything else -- a collision, more hits than the group holds, num_seeds < 2 --
// only marks the read (one byte per read, `marks`) for k_chain, which does the exact work.  k_chain then walks the marked reads only.
// ------------------------------------------------------------------------------------------------
#define CAND_HITS 64u                 // seed hits per read handled here
#define CAND_BLOOM_WORDS 512u         // most Bloom words per read: 16 384 bits
// dynamic LDS bytes of a block (16 reads): Bloom words | prefix of the list lengths | list starts | window positions (16 bits each: reads <= 65 535 letters).
// 18.5 KB with the default 128 Bloom words = 8 blocks per CU, the 8 waves per SIMD the kernel is compiled for (round 5; 20.5 KB = 7 blocks before)
#define CAND_LDS_BYTES(bw, handover) (16u * (((bw) + 2u * CAND_HITS + 1u) * 4u + ((handover) ? 2u * CAND_HITS : 0u)))
#define CAND_REC_MAX 64u              // positions of a marked read that k_cand hands over to k_chain as a record (99 % of the marked reads have no more)
#define CAND_REC_WORDS 32u            // words of mpool per read of the batch (a block's reads share their sum: room for its marked reads)
#define CAND_BLOCK 256u               // reads per block of k_cand (phase 1: a lane each; phase 2: sixteen lanes each, sixteen at a time)
// The hand-over (mrec != nullptr): k_cand has walked hit -> list bounds -> positions of every read it marks; k_chain would repeat those three
// dependent gathers one read per wave.  So a marked read with at most CAND_REC_MAX positions leaves a RECORD in mpool -- npos reference
// numbers | npos reference positions | npos window positions, in the order of the walk -- and {offset, npos} in mrec[r] ({NONE, 0}: none);
// k_chain builds the read's candidate set from it with coalesced loads.
struct __attribute__((aligned(4))) CandPair { uint32_t x, y; };          // two neighbouring words of the pool (4-byte aligned: global_load_dwordx2 takes that)
__global__ void __launch_bounds__(256, 8) k_cand(DReads rd, DIndex ix, DParams P, int pass, int is_last_strand, RState* __restrict__ work, RWork* __restrict__ rw,
                                              const uint32_t* __restrict__ pool, uint8_t* __restrict__ marks, uint32_t bloom_words,
                                              uint2* __restrict__ mrec, uint32_t* __restrict__ mpool, size_t mpool_words) {
  SMR_DYN_LDS(uint32_t, cand_lds);
  const int lane = lane_id(), gl = lane & 15, g = (int)(threadIdx.x >> 4);
  uint32_t* const bloom = cand_lds + (size_t)g * bloom_words;                                  // this read's 32 * bloom_words bits
  uint32_t* const hp_ = cand_lds + 16u * bloom_words + (size_t)g * (CAND_HITS + 1u);
  uint32_t* const lo_ = cand_lds + 16u * (bloom_words + CAND_HITS + 1u) + (size_t)g * CAND_HITS;
  uint16_t* const wn_ = reinterpret_cast<uint16_t*>(cand_lds + 16u * (bloom_words + 2u * CAND_HITS + 1u)) + (size_t)g * CAND_HITS;   // window position of every hit
  __shared__ uint32_t s_rec_cur;                            // words of the block's slice of mpool already given out
  __shared__ uint32_t s_wcnt[4], s_list[CAND_BLOCK];
  const uint32_t bshift = 32u - (5u + (uint32_t)__ffs((int)bloom_words) - 1u);
  // Phase 1, one LANE per read: the reads of the pass without enough seed hits end it here (pass control + write-back), the others are listed.  (Round 6:
  // sixteen lanes and a block slot per read for this -- 500 000 blocks of two dependent round trips each per 8 M-read launch -- was 1 ms of the kernel's
  // 1.7, and all of it on a batch against a reference that few reads hit.)
  uint32_t n_el;
  {
    const uint32_t r1 = blockIdx.x * CAND_BLOCK + threadIdx.x;
    bool el = false;
    if (r1 < rd.n) {
      RWork w1 = rw[r1];
      RState s1 = work[r1];                                 // (asked for with w1, not after it)
      const bool act = w1.strand_active && w1.search && w1.pass_n == (uint32_t)pass;
      el = act && s1.hit_seeds >= (uint32_t)P.num_seeds && w1.hit_total > 0;
      if (act && !el) chain_finish_read<true>(P, is_last_strand, r1, s1, w1, 1, true, work, rw);
      if (!el) { marks[r1] = 0; if (mrec) mrec[r1] = make_uint2(NONE, 0xFFFFFFFFu); }
    }
    const unsigned long long bm = __ballot(el);
    if (lane == 0) s_wcnt[threadIdx.x >> 6] = (uint32_t)__popcll(bm);
    if (threadIdx.x == 0) s_rec_cur = 0;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t q = 0; q < (threadIdx.x >> 6); q++) base += s_wcnt[q];
    if (el) s_list[base + (uint32_t)__popcll(bm & ((1ull << lane) - 1ull))] = r1;       // (in read order: the block's records lie in mpool in read order)
    n_el = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    __syncthreads();
  }
  // Phase 2, sixteen lanes per listed read, sixteen reads at a time
  for (uint32_t e0 = 0; e0 < n_el; e0 += 16u) {
  const bool eligible = e0 + (uint32_t)g < n_el;
  const uint32_t r = eligible ? s_list[e0 + (uint32_t)g] : 0u;
  RWork w;
  if (eligible) w = rw[r];
  const uint32_t nh = eligible ? w.hit_total : 0u;
  bool mark = eligible && (nh > CAND_HITS || P.num_seeds < 2);
  const bool scan = eligible && !mark;
  // the group's hits: list start and length of each, prefix over the lengths (row-wise, 16 hits per round)
  uint32_t npos = 0;
  if (scan) { for (uint32_t q = gl; q < bloom_words; q += 16) bloom[q] = 0; }
  // (all four rows' hit words are asked for, then all list bounds, before anything waits: two round trips per read, not eight)
  uint32_t hid[CAND_HITS / 16], hlo[CAND_HITS / 16], hln[CAND_HITS / 16], hwn[CAND_HITS / 16];
#pragma unroll
  for (uint32_t k = 0; k < CAND_HITS / 16; k++) {
    const uint32_t h = 16u * k + (uint32_t)gl;
    // the hit blocks of the passes run so far on this strand, concatenated (no loop over the three: an index that the compiler cannot
    // resolve sends the read's state to 12 KB of LDS per block).  The load itself is under no branch (a lane without a hit reads pool word 0):
    // its value merging with NONE at the end of an `if` made the compiler wait for each of the four loads in turn.
    const bool ok = scan && h < nh;
    const uint32_t c0 = ok ? w.blk_cnt[0] : 0u, c1 = ok ? w.blk_cnt[1] : 0u;
    const uint32_t at = !ok ? 0u : h < c0 ? w.blk_off[0] + 2 * h : h - c0 < c1 ? w.blk_off[1] + 2 * (h - c0) : w.blk_off[2] + 2 * (h - c0 - c1);
    const CandPair hw = *reinterpret_cast<const CandPair*>(pool + at);        // (id, win_pos) with one 8-byte load
    hid[k] = hw.x; hwn[k] = hw.y;                            // (the window position goes to LDS below: storing it here made every one of the four loads wait for itself)
  }
#pragma unroll
  for (uint32_t k = 0; k < CAND_HITS / 16; k++) if (!(scan && 16u * k + (uint32_t)gl < nh)) { hid[k] = NONE; hwn[k] = 0; }
#pragma unroll
  for (uint32_t k = 0; k < CAND_HITS / 16; k++) {
    hlo[k] = 0; hln[k] = 0;
    if (hid[k] != NONE) { hlo[k] = hid[k] + 1u; hln[k] = ix.pos_arr[hid[k]].x; }      // (a hit's id is the place of its list's header word: {positions, -})
  }
  if (mrec) {
#pragma unroll
    for (uint32_t k = 0; k < CAND_HITS / 16; k++) if (hid[k] != NONE) wn_[16u * k + (uint32_t)gl] = (uint16_t)hwn[k];
  }
#pragma unroll
  for (uint32_t k = 0; k < CAND_HITS / 16; k++) {
    const uint32_t h = 16u * k + (uint32_t)gl;
    const uint32_t lo = hlo[k], ln = hln[k];
    uint32_t inc = ln;                                     // inclusive prefix inside the row of 16 lanes (row_shr:1/2/4/8)
    uint32_t v;
    v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xF, 0xF, false); inc += v;
    v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xF, 0xF, false); inc += v;
    v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xF, 0xF, false); inc += v;
    v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xF, 0xF, false); inc += v;
    if (hid[k] != NONE) { hp_[h] = npos + inc - ln; lo_[h] = lo; }
    npos += (uint32_t)__shfl((int)inc, 15, 16);
  }
  if (scan && gl == 0) hp_[nh] = npos;
  __syncthreads();
  uint32_t rounds = scan ? (npos + 15u) / 16u : 0u;
  for (int d = 32; d > 0; d >>= 1) rounds = max(rounds, (uint32_t)__shfl_xor((int)rounds, d, 64));
  bool hit = false;
  // (the first 64 positions of a read -- four rounds -- stay in registers with their hits: if the read is marked they become its record)
  uint2 kp0 = make_uint2(0, 0), kp1 = kp0, kp2 = kp0, kp3 = kp0;
  uint32_t kh0 = 0, kh1 = 0, kh2 = 0, kh3 = 0;
  // (the same for the positions: the loads of the first four rounds are under way together before the first Bloom bit is set)
  auto hit_of = [&](uint32_t p) -> uint32_t {
    uint32_t h = 0;
    for (uint32_t step = 32; step > 0; step >>= 1) { const uint32_t t = h + step; if (t < nh && hp_[t] <= p) h = t; }
    return h;
  };
  auto bloom_set = [&](uint32_t seq) {
    const uint32_t hb = (seq * 2654435761u) >> bshift;                // 14 bits for 512 words
    const uint32_t old = atomicOr(&bloom[hb >> 5], 1u << (hb & 31u));
    hit |= ((old >> (hb & 31u)) & 1u) != 0;
  };
  if (rounds) {
    const uint32_t p0 = (uint32_t)gl, p1 = p0 + 16u, p2 = p0 + 32u, p3 = p0 + 48u;
    const bool a0 = scan && p0 < npos, a1 = scan && p1 < npos, a2 = scan && p2 < npos, a3 = scan && p3 < npos;
    if (a0) { kh0 = hit_of(p0); kp0 = ix.pos_arr[lo_[kh0] + (p0 - hp_[kh0])]; }
    if (a1) { kh1 = hit_of(p1); kp1 = ix.pos_arr[lo_[kh1] + (p1 - hp_[kh1])]; }
    if (a2) { kh2 = hit_of(p2); kp2 = ix.pos_arr[lo_[kh2] + (p2 - hp_[kh2])]; }
    if (a3) { kh3 = hit_of(p3); kp3 = ix.pos_arr[lo_[kh3] + (p3 - hp_[kh3])]; }
    if (a0) bloom_set(kp0.y);
    if (a1) bloom_set(kp1.y);
    if (a2) bloom_set(kp2.y);
    if (a3) bloom_set(kp3.y);
  }
  for (uint32_t it = 4; it < rounds; it++) {
    const uint32_t p = it * 16u + (uint32_t)gl;
    if (scan && p < npos) {
      const uint32_t h = hit_of(p);
      bloom_set(ix.pos_arr[lo_[h] + (p - hp_[h])].y);
    }
  }
  const unsigned long long hm = __ballot(hit);
  if (scan && ((hm >> (lane & 48)) & 0xFFFFull)) mark = true;
  if (eligible) {
    if (!mark) { RState st = work[r]; chain_finish_read<true>(P, is_last_strand, r, st, w, 1, gl == 0, work, rw); }      // (the read's state record only now: carried from the top it cost the kernel three registers it does not have)
  }
  // one byte per read says whether k_chain has to walk it: its waves claim reads by looking at 64 of these bytes, not at 64 per-read states
  if (eligible && gl == 0) marks[r] = mark ? 1 : 0;
  if (mrec) {
    // a block's records go into ITS slice of mpool (CAND_REC_WORDS per read of the block, placed by an LDS cursor: no atomic leaves the CU);
    // a read that does not fit leaves no record
    const bool want = scan && mark && npos > 0 && npos <= CAND_REC_MAX;
    uint32_t off = NONE;
    if (want && gl == 0) {
      const uint32_t old = atomicAdd(&s_rec_cur, 3u * npos);
      if (old + 3u * npos <= CAND_BLOCK * CAND_REC_WORDS && (size_t)(blockIdx.x + 1u) * CAND_BLOCK * CAND_REC_WORDS <= mpool_words) off = blockIdx.x * CAND_BLOCK * CAND_REC_WORDS + old;
    }
    off = (uint32_t)__shfl((int)off, lane & 48, 64);
    if (want && off != NONE) {
      const uint32_t g0 = (uint32_t)gl;
      if (g0 < npos) { mpool[off + g0] = kp0.y; mpool[off + npos + g0] = kp0.x; mpool[off + 2u * npos + g0] = wn_[kh0]; }
      if (g0 + 16u < npos) { mpool[off + g0 + 16u] = kp1.y; mpool[off + npos + g0 + 16u] = kp1.x; mpool[off + 2u * npos + g0 + 16u] = wn_[kh1]; }
      if (g0 + 32u < npos) { mpool[off + g0 + 32u] = kp2.y; mpool[off + npos + g0 + 32u] = kp2.x; mpool[off + 2u * npos + g0 + 32u] = wn_[kh2]; }
      if (g0 + 48u < npos) { mpool[off + g0 + 48u] = kp3.y; mpool[off + npos + g0 + 48u] = kp3.x; mpool[off + 2u * npos + g0 + 48u] = wn_[kh3]; }
    }
    // (a marked read without a record: .y says why -- its number of positions, ~0 when its hits outgrew the group (k_wlist's census, SMR_WALK_DEBUG))
    if (eligible && gl == 0) mrec[r] = (want && off != NONE) ? make_uint2(off, npos) : make_uint2(NONE, scan ? npos : 0xFFFFFFFFu);
  }
  __syncthreads();                                          // (the groups' LDS rows are free for the next sixteen reads)
  }
}

// ------------------------------------------------------------------------------------------------
// The exact candidate set of one read (alignment.cpp:117-148) in a hash table of `cap` slots -- bl: 32 * cap Bloom bits, sk: keys
// (reference number + 1), sc: exact counts.  Walk 1 over all positions of the read's hits: a reference whose Bloom bit is already set may
// occur twice and becomes a member of S.  Walk 2: exact counts of the members and their (pos, slot, win) tuples.  Candidates = members
// with count >= num_seeds, keyed (~count, ref) and sorted: count descending, reference ascending (:134-148).  false = the table overflowed
// (nothing usable was produced).  k_chain<false> runs it on the wave's LDS table, k_chain<true> on the block's global table.
// ------------------------------------------------------------------------------------------------
struct SetArgs {
  const uint32_t* rec;               // k_cand's record of the read (see k_cand) or nullptr: then the positions are found through hits / hp / pos_arr
  const uint2* pos_arr; const uint2* hits; const uint32_t* hp; uint32_t nh, npos, num_seeds;
  unsigned long long* gt; uint32_t pairs_cap, keys_cap; unsigned long long* l_keys; unsigned long long* gk; unsigned long long* ctr;
  uint32_t* s_ns; uint32_t* s_nt; uint32_t* s_ncand;
};
__device__ __forceinline__ bool chain_build_set(const SetArgs& A, uint32_t* bl, uint32_t* sk, uint32_t* sc, const uint32_t cap,
                                                bool& cap_err, uint32_t& ncand, unsigned long long*& keys) {
  const int lane = lane_id();
  const uint32_t mask = cap - 1, bshift = 32 - (5 + __ffs((int)cap) - 1);     // 32 * cap Bloom bits
  const uint32_t nh = A.nh, npos = A.npos;
  if (lane == 0) { *A.s_ns = 0; *A.s_nt = 0; *A.s_ncand = 0; }
  for (uint32_t q = lane; q < cap; q += 64) { bl[q] = 0; sk[q] = 0; sc[q] = 0; }
  __syncthreads();
  // walk 1: Bloom bitmap -> set S of references that may occur more than once
  bool s_over = false;
  // (four rows of 64 positions per trip, their loads asked for together: a read with hundreds of positions waited for memory once per row)
  for (uint32_t p0 = 0; p0 < npos; p0 += 256) {
    uint32_t sq4[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t p = p0 + 64u * u + lane;
      sq4[u] = 0;
      if (p < npos) {
        if (A.rec) sq4[u] = A.rec[p];
        else {
          uint32_t h = 0;
          for (uint32_t step = pow2_floor(nh); step > 0; step >>= 1) { const uint32_t t = h + step; if (t < nh && A.hp[t] <= p) h = t; }
          sq4[u] = A.pos_arr[A.hits[h].x + (p - A.hp[h])].y;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
    const uint32_t p = p0 + 64u * u + lane;
    if (p < npos) {
      const uint32_t seq = sq4[u];
      const uint32_t hb = (seq * 2654435761u) >> bshift;
      const uint32_t old = atomicOr(&bl[hb >> 5], 1u << (hb & 31u));
      if (((old >> (hb & 31u)) & 1u) || A.num_seeds < 2) {
        uint32_t sl = (seq * 0x9E3779B1u >> 7) & mask;
        for (uint32_t tries = 0; tries < cap; tries++) {
          const uint32_t o = atomicCAS(&sk[sl], 0u, seq + 1);
          if (o == 0) { if (atomicAdd(A.s_ns, 1u) + 1 > (cap * 3) / 4) s_over = true; break; }
          if (o == seq + 1) break;
          sl = (sl + 1) & mask;
        }
      }
    }
    }
  }
  __threadfence_block();
  __syncthreads();
  if (__any(s_over)) return false;
  if (*A.s_ns > 0) {
    // walk 2: exact counts for the members of S, and their (pos, win) tuples
    for (uint32_t p0 = 0; p0 < npos; p0 += 256) {
      uint2 pa4[4]; uint32_t win4[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t p = p0 + 64u * u + lane;
        pa4[u] = make_uint2(0u, 0u); win4[u] = 0;
        if (p < npos) {
          if (A.rec) { pa4[u] = make_uint2(A.rec[npos + p], A.rec[p]); win4[u] = A.rec[2u * npos + p]; }
          else {
            uint32_t h = 0;
            for (uint32_t step = pow2_floor(nh); step > 0; step >>= 1) { const uint32_t t = h + step; if (t < nh && A.hp[t] <= p) h = t; }
            pa4[u] = A.pos_arr[A.hits[h].x + (p - A.hp[h])]; win4[u] = A.hits[h].y;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
      const uint32_t p = p0 + 64u * u + lane;
      if (p < npos) {
        const uint2 pa = pa4[u]; const uint32_t win = win4[u];
        uint32_t sl = (pa.y * 0x9E3779B1u >> 7) & mask;
        for (;;) {
          const uint32_t o = sk[sl];
          if (o == pa.y + 1) {
            atomicAdd(&sc[sl], 1u);
            const uint32_t t = atomicAdd(A.s_nt, 1u);
            if (t < A.pairs_cap) A.gt[t] = ((unsigned long long)pa.x << 32) | ((unsigned long long)sl << 16) | win;
            break;
          }
          if (o == 0) break;
          sl = (sl + 1) & mask;
        }
      }
      }
    }
    __threadfence_block();
    __syncthreads();
    if (*A.s_nt > A.pairs_cap) { if (lane == 0) atomicAdd(&A.ctr[C_ERR_PAIRS], 1ull); cap_err = true; }
    // candidates: members of S with count >= num_seeds
    for (uint32_t q0 = 0; q0 < cap; q0 += 64) {
      const uint32_t q = q0 + lane;
      ncand += (uint32_t)__popcll(__ballot(sk[q] != 0 && sc[q] >= A.num_seeds));
    }
    if (ncand > A.keys_cap) { if (lane == 0) atomicAdd(&A.ctr[C_ERR_PAIRS], 1ull); ncand = 0; cap_err = true; }
    keys = ncand <= CH_KEYS_LDS ? A.l_keys : A.gk;
    if (!cap_err) for (uint32_t q0 = 0; q0 < cap; q0 += 64) {
      const uint32_t q = q0 + lane;
      if (sk[q] != 0 && sc[q] >= A.num_seeds) {
        const uint32_t c = atomicAdd(A.s_ncand, 1u);
        keys[c] = ((unsigned long long)(0xFFFFFFFFu - sc[q]) << 32) | (sk[q] - 1);
      }
    }
    __threadfence_block();
    __syncthreads();
    if (ncand > 1) wave_sort_u64(keys, ncand);
    __syncthreads();
  }
  return true;
}
// group the tuples by member of S: exclusive prefix of the exact counts (so[] = where a member's tuples start), then every tuple to its
// member's next free place; a candidate's (ref_pos, read_pos) pairs are then a slice instead of a filter over all tuples
__device__ __forceinline__ void chain_group_tuples(const SetArgs& A, uint32_t* sc, uint32_t* so, uint32_t cap, unsigned long long* gt2) {
  const int lane = lane_id();
  const uint32_t per = cap / 64;
  uint32_t loc = 0;
  for (uint32_t q = 0; q < per; q++) loc += sc[lane * per + q];
  uint32_t tot; uint32_t run = wave_excl_scan_u32(loc, tot);
  for (uint32_t q = 0; q < per; q++) { const uint32_t cnt = sc[lane * per + q]; so[lane * per + q] = run; sc[lane * per + q] = run; run += cnt; }
  __threadfence_block();
  __syncthreads();
  const uint32_t nt_ = min(*A.s_nt, A.pairs_cap);
  for (uint32_t t0 = 0; t0 < nt_; t0 += 64) {
    const uint32_t t = t0 + lane;
    if (t < nt_) {
      const unsigned long long tv = A.gt[t];
      const uint32_t p = atomicAdd(&sc[(uint32_t)((tv >> 16) & 0xFFFFu)], 1u);
      gt2[p] = (tv & 0xFFFFFFFF00000000ull) | (tv & 0xFFFFull);
    }
  }
  __threadfence_block();
  __syncthreads();
}

// k_chain<EXT, LONG>: compute_lis_alignment (alignment.cpp:100-509) for the reads k_cand marked.  One block = one wave, persistent: chunks of
// 64 reads are claimed with one atomic, their flags fetched by the 64 lanes at once, the marked ones walked one by one.
// Dynamic LDS (bytes), ML = max_len rounded to 16, MQ = min(ML, SW_X4_MAX_ROWS), RF = ML + 2 * edges + 16 rounded, RQ = the same for a read of MQ letters:
//   read slots    5 MQ          the read being walked | four parked reads      (reads of more than MQ letters: global, see below)
//   window slots  9 RQ          0..3: the batch of the read being walked, 4..7: parked tasks
//   keys[CH_KEYS_LDS] u64 | region R = max(4 CH_PAIRS_LDS, 2 s_cap) u32: pairs (two halves) + serial-LIS arrays during the candidate loop,
//   Bloom words + counts while the candidate set is built | hits[CH_HITS_LDS] uint2 | hp[CH_HITS_LDS + 8] u32 | skey[s_cap] u32
// Candidate references (alignment.cpp:117-148) without a per-reference counter array: see chain_build_set.  EXT = true is the second
// instantiation for the reads whose set outgrew the LDS table (global table of the block, tuples grouped by member).
// one Smith-Waterman task of the candidate walk: reference, window geometry (alignment.cpp:271-357)
struct SwTask { uint32_t max_ref; uint64_t rf_start, align_ref_start, head, align_que_start; int m, nref; };
#ifndef SMR_CHAIN_WAVES_PER_SIMD
#define SMR_CHAIN_WAVES_PER_SIMD 3
#endif
template <bool EXT, bool LONG, bool STRIPED = false>
__global__ void __launch_bounds__(64, SMR_CHAIN_WAVES_PER_SIMD) k_chain(DReads rd, DIndex ix, DParams P, int pass, int is_last_strand,
                                              RState* __restrict__ work, AlignRec* __restrict__ work_aln, RWork* __restrict__ rw,
                                              const uint32_t* __restrict__ pool, unsigned long long* __restrict__ ctr,
                                              unsigned long long* g_tuples, unsigned long long* g_keys, unsigned long long* g_pairs, uint32_t* g_lis,
                                              uint2* g_hits, uint32_t keys_cap, uint32_t pairs_cap, uint32_t hits_cap,
                                              uint32_t lds_ml, uint32_t lds_rf, uint32_t s_cap, uint32_t* g_stab, unsigned long long* g_tuples2,
                                              uint32_t lds_rq, int* g_bound, uint8_t* g_rdq, uint8_t* __restrict__ marks,
                                              const uint2* __restrict__ mrec, const uint32_t* __restrict__ mpool,
                                              const uint32_t* __restrict__ rlist, const unsigned long long* __restrict__ rlist_n) {
  SMR_DYN_LDS(unsigned char, lds_raw);
  __shared__ uint32_t s_next;
  __shared__ uint32_t s_ncand;
  const int lane = lane_id();
  // LDS holds the letters and reference windows of reads of ONE strip only (<= SW_X4_MAX_ROWS letters, lds_mq / lds_rq bytes per slot):
  // rdq: read slot 0 (the read being walked) + 4 slots (the parked reads); rfq: 9 reference-window slots (0..3 = the batch of the read being
  // walked, 4..7 = the parked tasks).  A LONGER read never shares a Smith-Waterman pass with others: its letters go to this block's row of
  // g_rdq, its reference window is read where it lies (ix.ref_seq: same alphabet), and the strip-boundary rows (2 ints per reference column)
  // live in g_bound -- so a wave that walks 5 kb reads keeps ~13 KB of LDS instead of ~60 KB: 12 waves per CU (the register budget) instead of 2
  uint8_t* rdq = lds_raw;
  const uint32_t lds_mq = min(lds_ml, (uint32_t)SW_X4_MAX_ROWS);
  uint8_t* rfq = rdq + 5 * (size_t)lds_mq;
  uint8_t* const rfq1 = rfq + lds_rq;
  auto wslot = [&](int e) -> uint8_t* { return e == 0 ? rfq : rfq1 + (size_t)(e - 1) * lds_rq; };
  int* bound = g_bound ? g_bound + (size_t)blockIdx.x * 2 * lds_rf : nullptr;
  unsigned long long* l_keys = (unsigned long long*)(rfq1 + 8 * (size_t)lds_rq);
  // region R: [pairs | lis] during the candidate loop, [bloom | scnt] while the candidate set is built (dead afterwards)
  unsigned long long* l_pairs = l_keys + CH_KEYS_LDS;
  uint32_t* l_lis = (uint32_t*)(l_pairs + CH_PAIRS_LDS);
  uint32_t* bloom = (uint32_t*)l_pairs;
  uint32_t* scnt = bloom + s_cap;
  const uint32_t r_words = max(4u * CH_PAIRS_LDS, 2u * s_cap);
  uint2* l_hits = (uint2*)((uint32_t*)l_pairs + r_words);
  uint32_t* l_hp = (uint32_t*)(l_hits + CH_HITS_LDS);
  uint32_t* skey = l_hp + CH_HITS_LDS + 8;
  __shared__ uint32_t s_ns, s_nt;
  // the batch cache of the walk (tasks scored together, their forward results, the LDS slot of each reference window): indexed with run-time
  // values, so as local arrays they lived in private memory (scratch loads and stores on every look-up) -- a wave's LDS is the place
  __shared__ SwTask s_ctk[4];
  __shared__ SwRes s_cfw[4];
  __shared__ int s_cslot[4];
  const uint32_t s_mask = s_cap - 1;

  unsigned long long* gt = g_tuples + (size_t)blockIdx.x * pairs_cap;
  // a read whose set S outgrows the LDS table (thousands of references sharing seeds with it) builds it again in this block's global table
  // of CH_EXT_CAP slots (Bloom words | keys | counts | tuple offsets); its tuples are then grouped by member in gt2 (g_stab / g_tuples2 are
  // only allocated after a first overflow, see smr_align_part)
  uint32_t* xt = g_stab ? g_stab + (size_t)blockIdx.x * 4 * CH_EXT_CAP : nullptr;
  unsigned long long* gt2 = g_tuples2 ? g_tuples2 + (size_t)blockIdx.x * pairs_cap : nullptr;
  unsigned long long* gk = g_keys + (size_t)blockIdx.x * keys_cap;
  unsigned long long* gp = g_pairs + (size_t)blockIdx.x * pairs_cap;
  uint32_t* gl = g_lis + (size_t)blockIdx.x * 2 * pairs_cap;
  uint2* gh = g_hits + (size_t)blockIdx.x * hits_cap;

  unsigned long long n_fwd = 0, n_cells = 0, n_spec = 0, n_spec_used = 0;   // flushed once per block (lane 0)
#ifdef SMR_CHAIN_PHASES                                   // per-phase cycle accounting (build with -DSMR_CHAIN_PHASES, run with SMR_DEBUG_PHASES=1)
  unsigned long long tph[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#define TPH(i) { const unsigned long long tn_ = clock64(); tph[i] += tn_ - tlast; tlast = tn_; }
#else
#define TPH(i)
#endif
#ifdef SMR_CHAIN_STATS                                    // read-class census in the same seven debug slots (build with -DSMR_CHAIN_STATS instead)
  unsigned long long tph[7] = {0, 0, 0, 0, 0, 0, 0};
#define TST(i) { tph[i]++; }
#else
#define TST(i) {}
#endif
  auto finish_read = [&](uint32_t r, RState& st, RWork& w, int search, bool writer) { chain_finish_read(P, is_last_strand, r, st, w, search, writer, work, rw); };
  // Reads whose walk meets exactly ONE Smith-Waterman task (the usual case for a background read with a spurious candidate) are PARKED:
  // task and sequences go into one of four LDS slots, nothing is written back, the wave moves on to its next read.  Four parked tasks
  // of four different reads are scored by one pass of the four-problem kernel.  A result that is "no alignment" -- what the walk was
  // run ahead under -- completes its read exactly as the sequential walk would have; any other result sends the read through the
  // walk again from the start, in immediate mode, with this result already in its cache.
  __shared__ uint32_t q_r[4], q_max_ref[4], q_aq[4], q_m[4], q_nref[4], q_ars[4], q_head[4];
  __shared__ unsigned long long q_rf[4];
  __shared__ int q_score[4], q_eref[4], q_eread[4];
  uint32_t q_n = 0, n_redo = 0, redo_slots = 0;          // parked tasks; parked reads to walk again (bit mask of their slots)
  bool q_hasn = false, need_flush = false, out_of_reads = false;
  uint32_t chunk_base = 0;                               // reads are claimed 64 at a time (one atomic per chunk: 31 k instead of 2 M same-address atomics per launch);
  unsigned long long chunk_todo = 0;                     // bit i: read chunk_base + i is still to be walked
  uint32_t chunk_ri = 0;                                 // lane i: that read (chunk_base + i, or entry chunk_base + i of rlist)
  for (;;) {
    uint32_t r;
    int mode = 0, seed_slot = -1;                         // mode 1: immediate (sequential walk), possibly seeded with the parked task's result
    if (n_redo > 0) {
      seed_slot = __ffs((int)redo_slots) - 1; redo_slots &= redo_slots - 1; n_redo--;
      r = q_r[seed_slot]; mode = 1;
    } else if (need_flush || (out_of_reads && q_n > 0)) {
      // ---- score the parked tasks together ----
      __syncthreads();
      TPH(5)
      if (q_n >= 2) {
        const int g = lane >> 4;
        const bool mine = (uint32_t)g < q_n;
        const int gm = mine ? (int)q_m[g] : 0, gn = mine ? (int)q_nref[g] : 0, gq = mine ? (int)q_aq[g] : 0;
        int mm = gm;
        for (int d = 32; d > 0; d >>= 1) mm = max(mm, __shfl_xor(mm, d, 64));
        const SwRes r4 = sw_wave_x4(rdq + (size_t)lds_mq + (size_t)g * lds_mq, gm, gq, 1, wslot(4 + g), gn, 0, 1, P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext, mm, q_hasn);
        if ((lane & 15) == 0 && mine) { q_score[g] = r4.score; q_eref[g] = r4.end_ref; q_eread[g] = r4.end_read; }
      } else {
        const SwRes r1 = sw_wave_t<STRIPED>(rdq + lds_mq, (int)q_m[0], (int)q_aq[0], 1, wslot(4), (int)q_nref[0], 0, 1, bound, P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext, P.sw_mode, 0, 0, sw_scr(P));
        if (lane == 0) { q_score[0] = r1.score; q_eref[0] = r1.end_ref; q_eread[0] = r1.end_read; }
      }
      __syncthreads();
      TPH(6)
      for (uint32_t e = 0; e < q_n; e++) {
        const int score1 = q_score[e] > 65535 ? 65535 : q_score[e];
        if ((uint32_t)score1 > P.minimal_score) { redo_slots |= 1u << e; n_redo++; continue; }
        // no alignment: the read ends as the sequential walk ends it -- one ssw_align call, nothing recorded
        const uint32_t rr = q_r[e];
        RWork w2 = rw[rr];
        RState st2 = work[rr];
        if (w2.has_amb && !w2.is04) { w2.is04 = 1; w2.aval = 4; }          // read.flip34() before SSW (:360-361)
        n_fwd++; n_cells += (unsigned long long)q_m[e] * q_nref[e]; n_spec++; n_spec_used++;
        finish_read(rr, st2, w2, 1, lane == 0);
      }
      q_n = 0; q_hasn = false; need_flush = false;
      __syncthreads();
      continue;
    } else {
      if (out_of_reads) break;
      if (chunk_todo == 0) {
        // claim the next reads.  Lane i looks at the mark of read i of the chunk (k_cand: reads outside this (strand, pass) and reads without
        // the seeds for compute_lis_alignment, :103-108, are not marked); the marked ones are walked one by one
        __syncthreads();
        // (a claim is 64 reads when the batch has many per wave -- one atomic per 64 reads instead of one per read --, fewer when it has few: a batch
        // of 50 000 long reads claimed 64 at a time kept 781 of the 3 072 waves busy)
        // (rlist: the marked reads the split path of smr_walk.hpp left to this kernel, as a list -- a launch that follows it claims over that list, not over the batch)
        const uint32_t n_claim = rlist ? (uint32_t)*rlist_n : rd.n;
        const uint32_t claim = max(1u, min(64u, n_claim / (gridDim.x * 4u)));
        if (lane == 0) s_next = (uint32_t)atomicAdd(&ctr[C_WORK_NEXT], (unsigned long long)claim);
        __syncthreads();
        chunk_base = s_next;
        if (chunk_base >= n_claim) { out_of_reads = true; continue; }
        // (k_cand ended the pass of every read it did not mark; 2 = left to the EXT launch by the first one)
        const uint32_t li = chunk_base + (uint32_t)lane;
        const bool in_claim = (uint32_t)lane < claim && li < n_claim;
        chunk_ri = rlist ? (in_claim ? rlist[li] : 0u) : li;
        const uint32_t mk = in_claim ? marks[chunk_ri] : 0u;
        const bool todo = EXT ? mk == 2u : mk == 1u;
        chunk_todo = __ballot(todo);
        if (chunk_todo == 0) continue;
      }
      __syncthreads();
      r = (uint32_t)__builtin_amdgcn_readlane((int)chunk_ri, __ffsll((long long)chunk_todo) - 1);
      chunk_todo &= chunk_todo - 1;
    }
    // (all four loads before the first use of any: one round trip, not two -- a marked read is always active.  Tried and dropped: the
    // 22 state words fetched one per lane and read back with v_readlane, the next marked read's a whole read ahead: k_chain 6.0 -> 6.3 ms)
    RWork w = rw[r];
    RState st = work[r];
    const uint32_t len = rd.len[r];
    const uint32_t* rec = rd.words + rd.rec_off[r];
    const uint2 mr = (!EXT && mrec) ? mrec[r] : make_uint2(NONE, 0u);       // k_cand's record of the read's positions, if it left one
    if (!(w.strand_active && w.search && w.pass_n == (uint32_t)pass)) continue;
    int search = 1;
    const uint32_t max_SW_score = len * (uint32_t)P.match;
    const bool x4_ok = P.sw_mode >= 1 && len <= SW_X4_MAX_ROWS && sw_pk_fits((int)len, (int)lds_rf, P.match, P.mismatch, P.score_N, P.gap_open);
    const bool long_rd = LONG && len > lds_mq;             // more than one strip: letters in g_rdq, reference windows read in place (LONG: the batch has such reads)
    uint8_t* const RD = long_rd ? g_rdq + (size_t)blockIdx.x * lds_ml : rdq;
    bool parked = false;

    if (st.hit_seeds >= (uint32_t)P.num_seeds && w.hit_total > 0) {
      // ---------------- compute_lis_alignment (alignment.cpp:100-509) ----------------
      TPH(0)
      if (mode == 0) TST(0)
      // gather this strand's hits (all passes so far) into a flat array
      const uint32_t nh = w.hit_total;
      uint2* hits = nh <= CH_HITS_LDS ? l_hits : gh;
      bool cap_err = nh > hits_cap && nh > CH_HITS_LDS;
      if (cap_err) { if (lane == 0) atomicAdd(&ctr[C_ERR_PAIRS], 1ull); }
      else {
        const bool from_rec = mr.x != NONE;
        uint32_t o = 0;
        if (!from_rec) for (uint32_t pp = 0; pp < 3; pp++) {              // one contiguous block per pass run so far on this strand
          const uint32_t c = w.blk_cnt[pp], bo = w.blk_off[pp];
          for (uint32_t q = lane; q < c; q += 64) hits[o + q] = make_uint2(pool[bo + 2 * q], pool[bo + 2 * q + 1]);
          o += c;
        }
        __syncthreads();
        // 1. per-reference counts of seed hits (:117-130)
        // prefix over the position-list lengths; hits[h].x becomes the list start
        uint32_t* hp = nh <= CH_HITS_LDS ? l_hp : gl;
        uint32_t npos = from_rec ? mr.y : 0u;
        if (!from_rec) for (uint32_t hb = 0; hb < nh; hb += 64) {
          const uint32_t h = hb + lane;
          uint32_t lo = 0, ln = 0;
          if (h < nh) { const uint32_t id = hits[h].x; lo = id + 1u; ln = ix.pos_arr[id].x; }
          uint32_t tot; const uint32_t ex = wave_excl_scan_u32(ln, tot);
          if (h < nh) { hp[h] = npos + ex; hits[h].x = lo; }
          npos += tot;
        }
        if (lane == 0 && !from_rec) hp[nh] = npos;
        uint32_t ncand = 0;
        unsigned long long* keys = l_keys;
        SetArgs sa;
        sa.rec = from_rec ? mpool + mr.x : nullptr;
        sa.pos_arr = ix.pos_arr; sa.hits = hits; sa.hp = hp; sa.nh = nh; sa.npos = npos; sa.num_seeds = (uint32_t)P.num_seeds;
        sa.gt = gt; sa.pairs_cap = pairs_cap; sa.keys_cap = keys_cap; sa.l_keys = l_keys; sa.gk = gk; sa.ctr = ctr;
        sa.s_ns = &s_ns; sa.s_nt = &s_nt; sa.s_ncand = &s_ncand;
        // EXT = false: the set in the wave's LDS table; a read that overflows it is marked for the EXT = true launch of this kernel, which
        // keeps the set in the block's global table (CH_EXT_CAP slots) and groups the tuples by member
        uint32_t* const t_bloom = EXT ? xt : bloom;
        uint32_t* const t_skey = EXT ? xt + CH_EXT_CAP : skey;
        uint32_t* const t_scnt = EXT ? xt + 2 * CH_EXT_CAP : scnt;
        // (the LDS table is cleared and scanned per read: a read with a few dozen positions -- the usual case -- takes a table of twice that, not all s_cap slots)
        const uint32_t t_cap = EXT ? CH_EXT_CAP : min(s_cap, max(64u, (npos > 1 && npos < (1u << 20)) ? 1u << (32 - __clz((int)(2 * npos - 1))) : (npos > 1 ? s_cap : 64u))), t_mask = t_cap - 1;
        const bool set_ok = chain_build_set(sa, t_bloom, t_skey, t_scnt, t_cap, cap_err, ncand, keys);
        if (!set_ok) {
          if (!EXT && xt) {                                  // leave the read to the EXT launch: nothing of it is written but the mark
            if (lane == 0) marks[r] = 2;
            continue;
          }
          if (lane == 0) atomicAdd(&ctr[C_ERR_SCAP], 1ull);
          cap_err = true; ncand = 0;
        }
        if (EXT && set_ok && ncand > 0 && !cap_err) chain_group_tuples(sa, t_scnt, xt + 3 * CH_EXT_CAP, t_cap, gt2);
        TPH(3)
        if (mode == 0 && ncand > 0) TST(2)
#ifdef SMR_CHAIN_STATS
        if (mode == 0 && nh <= 32 && npos <= 64) TST(1)                       // census: small enough for a 16-lane walk
        if (mode == 0 && nh <= 32 && npos <= 64 && ncand <= 8 && (ncand == 0 || 0xFFFFFFFFu - (uint32_t)(keys[0] >> 32) <= 16u)) TST(6)
#endif
        const uint32_t ntup = min(s_nt, pairs_cap);

        // 2. candidate loop (:150-508), organised as a GENERATOR of Smith-Waterman tasks.
        // The reference walks the candidates and, inside each, a sliding window over its (ref_pos, read_pos) pairs; every window
        // whose LIS is long enough costs one ssw_align, and what happens next depends on its result.  Most results are "no
        // alignment" (spurious candidates of background reads come 40 at a time and never align), so the walk is run AHEAD on a
        // copy of its state under that assumption to collect up to four tasks, which one pass of the four-problem SW kernel
        // (sw_wave_x4) scores together.  The real walk then consumes the results in its own order, looking each task up by its
        // geometry; a result that did align changes the real walk's course (heuristic 1, best, termination), tasks predicted
        // past that point simply go unused, and the walk asks for a new batch when it reaches a task that is not in the cache.
        // The results a read ends up with are those of the sequential walk; only the order of evaluation differs.
        struct Walk {                        // where the walk stands: candidate k, window iterators over its sorted pairs
          uint32_t k, np, it, ms_lo, ms_hi, begin_ref, begin_read;
          uint64_t ref0, reflen;             // where candidate k's reference sequence starts, and its length (fetched while its pairs are being collected)
          int is_aligned, best, go_on;       // go_on = is_search_candidates
          int started, pending_pop, buf;     // buf: which pairs buffer holds candidate k (0/1: halves of l_pairs, 2: all of l_pairs, 3: global)
        };
        uint32_t buf_tag[2] = {0xFFFFFFFFu, 0xFFFFFFFFu}, buf_np[2] = {0, 0};
        int real_buf = 0;                                                     // the buffer the real walk's current candidate lives in
        const uint64_t rlen = len;
        auto pairs_of = [&](int bufid) -> unsigned long long* { return bufid == 0 ? l_pairs : bufid == 1 ? l_pairs + CH_PAIRS_LDS / 2 : bufid == 2 ? l_pairs : gp; };

        // candidate wk.k: termination rules (:156-169), its hits (:181-201) into a pairs buffer, sorted.
        // 1 = loaded, 0 = the candidate loop ends here, 2 = (look-ahead only) this candidate has to be left to the real walk
        auto load_candidate = [&](Walk& wk, bool real) -> int {
          if (wk.k >= ncand || !wk.go_on || cap_err) return 0;
          const unsigned long long ck = keys[wk.k];
          const uint32_t max_ref = (uint32_t)ck;
          const uint32_t max_occur = 0xFFFFFFFFu - (uint32_t)(ck >> 32);
          if (max_occur < (uint32_t)P.num_seeds) return 0;
          const uint64_t r0_ = ix.ref_off[max_ref], r1_ = ix.ref_off[max_ref + 1];          // (used at the end: in flight during the rest)
          if (wk.is_aligned && P.min_lis > 0 && wk.k > 0 && max_occur < (0xFFFFFFFFu - (uint32_t)(keys[wk.k - 1] >> 32))) {   // :165-169
            --wk.best;
            if (wk.best < 1) return 0;
          }
          if (real && buf_tag[0] == wk.k) { wk.buf = 0; wk.np = buf_np[0]; }
          else if (real && buf_tag[1] == wk.k) { wk.buf = 1; wk.np = buf_np[1]; }
          else {
            uint32_t cslot = (max_ref * 0x9E3779B1u >> 7) & t_mask;
            while (t_skey[cslot] != max_ref + 1) cslot = (cslot + 1) & t_mask;
            const uint32_t np = max_occur;                                    // the exact count of the member = the number of its tuples
            int bufid;
            if (!real) {                                                      // the look-ahead never touches the buffer the real walk stands on,
              if (np > CH_PAIRS_LDS / 2 || real_buf == 2) return 2;           // and leaves a large candidate to the real walk
              bufid = real_buf == 0 ? 1 : 0;
            }
            else if (np <= CH_PAIRS_LDS / 2) bufid = 0;
            else if (np <= CH_PAIRS_LDS) bufid = 2;
            else if (np <= pairs_cap) bufid = 3;
            else { if (lane == 0) atomicAdd(&ctr[C_ERR_PAIRS], 1ull); cap_err = true; return 0; }
            unsigned long long* pw_ = pairs_of(bufid);
            __syncthreads();
            if (EXT) {                                                        // grouped tuples: the candidate's pairs are a slice
              const uint32_t start = xt[3 * CH_EXT_CAP + cslot];
              for (uint32_t q = lane; q < np; q += 64) pw_[q] = gt2[start + q];
            } else {
              uint32_t run = 0;
              for (uint32_t t0 = 0; t0 < ntup; t0 += 64) {
                const uint32_t t = t0 + lane;
                unsigned long long tv = 0;
                bool mt = false;
                if (t < ntup) { tv = gt[t]; mt = (uint32_t)((tv >> 16) & 0xFFFFu) == cslot; }
                const unsigned long long mm = __ballot(mt);
                if (mt) pw_[run + (uint32_t)__popcll(mm & ((1ull << lane) - 1))] = (tv & 0xFFFFFFFF00000000ull) | (tv & 0xFFFFull);
                run += (uint32_t)__popcll(mm);
              }
            }
            __syncthreads();
            if (np > 1) wave_sort_u64(pw_, np);
            __syncthreads();
            if (bufid >= 2) { buf_tag[0] = buf_tag[1] = 0xFFFFFFFFu; }
            else { buf_tag[bufid] = wk.k; buf_np[bufid] = np; }
            wk.buf = bufid; wk.np = np;
          }
          if (real) real_buf = wk.buf;
          const unsigned long long* pairs = pairs_of(wk.buf);
          wk.it = 0; wk.ms_lo = 0; wk.ms_hi = 0;
          wk.begin_ref = (uint32_t)(pairs[0] >> 32); wk.begin_read = (uint32_t)pairs[0];
          wk.pending_pop = 0;
          wk.ref0 = r0_; wk.reflen = r1_ - r0_;
          return 1;
        };

        // the sliding window of read length along candidate wk.k (:203-506), up to its next window that calls for ssw_align
        auto next_task = [&](Walk& wk, SwTask& tk) -> bool {
          const unsigned long long* pairs = pairs_of(wk.buf);
          const uint32_t np = wk.np;
          const uint32_t max_ref = (uint32_t)keys[wk.k];
          while (wk.it != np && wk.go_on) {
            if (!wk.pending_pop) {
              wk.pending_pop = 1;
              const uint64_t end_ref_max = (uint64_t)wk.begin_ref + len - wk.begin_read - P.lnwin + 1;
              int push = 0;
              // (the pairs are sorted by reference position: the ones to push are a prefix of what is left -- 64 looked at per trip, one per lane)
              for (;;) {
                const uint32_t pi = wk.it + (uint32_t)lane;
                const bool okp = pi < np && (uint64_t)(uint32_t)(pairs[min(pi, np - 1)] >> 32) <= end_ref_max;
                const unsigned long long pm = __ballot(okp);
                const uint32_t pc = pm == ~0ull ? 64u : (uint32_t)__ffsll((long long)~pm) - 1u;
                if (pc) { wk.it += pc; wk.ms_hi = wk.it; push = 1; }
                if (pc < 64u) break;
              }
              int skip_to_pop = 0;
              if (!push && wk.is_aligned) skip_to_pop = 1;        // heuristic 1 (:243-246)
              else wk.is_aligned = 0;
              if (!skip_to_pop && (wk.ms_hi - wk.ms_lo) >= (uint32_t)P.num_seeds) {
                uint32_t lis0;
                const uint32_t nw = wk.ms_hi - wk.ms_lo;
                uint32_t* lisb = nw <= CH_PAIRS_LDS ? l_lis : gl;
                const uint32_t nl = nw <= 64 ? wave_lis_first(pairs + wk.ms_lo, nw, lis0) : serial_lis_first(pairs + wk.ms_lo, nw, lisb, lisb + (nw <= CH_PAIRS_LDS ? CH_PAIRS_LDS : pairs_cap), lis0);
                if (nl >= (uint32_t)P.min_lis) {
                  const uint32_t lcs_ref_start = (uint32_t)(pairs[wk.ms_lo + lis0] >> 32);
                  const uint32_t lcs_que_start = (uint32_t)pairs[wk.ms_lo + lis0];
                  const uint64_t reflen = wk.reflen;
                  uint64_t head = 0, tail = 0, align_ref_start = 0, align_que_start = 0, align_length = 0;
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
                  tk.max_ref = max_ref; tk.align_ref_start = align_ref_start; tk.head = head; tk.align_que_start = align_que_start;
                  tk.m = (int)(align_length - head - tail); tk.nref = (int)align_length;
                  tk.rf_start = wk.ref0 + align_ref_start - head;
                  return true;
                }
              }
            }
            // pop (:486-506)
            wk.pending_pop = 0;
            if (wk.ms_hi > wk.ms_lo) wk.ms_lo++;
            if (wk.ms_hi == wk.ms_lo) {
              if (wk.it != np) { wk.begin_ref = (uint32_t)(pairs[wk.it] >> 32); wk.begin_read = (uint32_t)pairs[wk.it]; }
              else break;
            } else { wk.begin_ref = (uint32_t)(pairs[wk.ms_lo] >> 32); wk.begin_read = (uint32_t)pairs[wk.ms_lo]; }
          }
          return false;
        };

        // 1 = the walk stands at a task, 0 = the walk is over, 2 = (look-ahead only) cannot look further
        auto advance = [&](Walk& wk, bool real, SwTask& tk) -> int {
          for (;;) {
            if (!wk.started) {
              TPH(5)
              const int lc = load_candidate(wk, real);
              TPH(1)
              if (lc != 1) return lc;
              wk.started = 1;
            }
            const bool got = next_task(wk, tk);
            TPH(2)
            if (got) return 1;
            wk.k++; wk.started = 0;
            __syncthreads();
          }
        };
        // (a task that is parked or scored next to others goes into one of the slots 1..8)
        auto task_fits = [&](const SwTask& t) -> bool { return t.m > 0 && t.nref > 0 && (uint32_t)t.m <= lds_mq && (uint32_t)t.nref <= lds_rq; };

        Walk R;
        R.k = 0; R.np = 0; R.it = 0; R.ms_lo = 0; R.ms_hi = 0; R.begin_ref = 0; R.begin_read = 0;
        R.is_aligned = 0; R.best = w.best; R.go_on = 1; R.started = 0; R.pending_pop = 0; R.buf = 0; R.ref0 = 0; R.reflen = 0;
        SwTask* const ctk = s_ctk; SwRes* const cfw = s_cfw; int* const cslot = s_cslot; int n_cached = 0;     // the batch cache: task, forward result, LDS slot of its reference window
        // (every lane writes the same values; the wave barriers keep a lane that runs ahead from rewriting an entry that another is still reading)
        __builtin_amdgcn_wave_barrier();
        for (int e = 0; e < 4; e++) cslot[e] = e;
        bool rdq_staged = false;
        bool immediate = mode == 1 || !x4_ok;
        if (mode == 1 && seed_slot >= 0) {
          ctk[0].max_ref = q_max_ref[seed_slot]; ctk[0].rf_start = q_rf[seed_slot]; ctk[0].align_ref_start = q_ars[seed_slot]; ctk[0].head = q_head[seed_slot];
          ctk[0].align_que_start = q_aq[seed_slot]; ctk[0].m = (int)q_m[seed_slot]; ctk[0].nref = (int)q_nref[seed_slot];
          cfw[0].score = q_score[seed_slot]; cfw[0].end_ref = q_eref[seed_slot]; cfw[0].end_read = q_eread[seed_slot];
          cslot[0] = 4 + seed_slot; n_cached = 1;
        }
        __builtin_amdgcn_wave_barrier();
        if (!immediate) {
          // run the walk ahead assuming that nothing aligns: no task -> the read is finished; exactly one -> park it; more -> walk it now
          Walk L = R;
          SwTask t1, t2;
          const int c1 = advance(L, false, t1);
          if (c1 == 1) {
            L.is_aligned = 0;
            const int c2 = advance(L, false, t2);
            if (c2 == 0 && task_fits(t1)) {
              const uint32_t e = q_n;
              const uint32_t aval = (w.has_amb && !w.is04) ? 4u : (uint32_t)w.aval;             // read.flip34() before SSW (:360-361)
              TPH(5)
              __syncthreads();
              uint8_t* rq = rdq + (size_t)lds_mq + (size_t)e * lds_mq;
              uint8_t* fq = wslot(4 + (int)e);
              for (uint32_t q = lane; q < len; q += 64) rq[q] = (uint8_t)read_nt(rec, len, q, w.reversed, aval);
              bool hn = false;
              for (int q = lane; q < t1.nref; q += 64) { const uint8_t ch = ix.ref_seq[t1.rf_start + q]; fq[q] = ch; hn |= ch == 4; }
              if (__any(hn)) q_hasn = true;
              if (lane == 0) {
                q_r[e] = r; q_max_ref[e] = t1.max_ref; q_rf[e] = t1.rf_start; q_ars[e] = (uint32_t)t1.align_ref_start; q_head[e] = (uint32_t)t1.head;
                q_aq[e] = (uint32_t)t1.align_que_start; q_m[e] = (uint32_t)t1.m; q_nref[e] = (uint32_t)t1.nref;
              }
              __syncthreads();
              q_n++;
              TPH(4)
              if (q_n == 4) need_flush = true;
              parked = true;
              TST(3)
            } else immediate = true;
          } else if (c1 == 2) immediate = true;
        }
        if (immediate && mode == 0) TST(4)
        if (mode == 1) TST(5)
        if (immediate)
        for (;;) {
          SwTask tk;
          if (advance(R, true, tk) != 1) break;
          // read.flip34() to the 0..4 alphabet before SSW (:360-361)
          // (is03/is04 only toggle when the read has ambiguous letters; aval tracks the stored value)
          if (w.has_amb && !w.is04) { w.is04 = 1; w.aval = 4; rdq_staged = false; }
          const int m = tk.m, nref = tk.nref;
          const bool sw_ok = (m > 0 && nref > 0 && (uint32_t)m <= lds_ml && (uint32_t)nref <= lds_rf);
          if (!sw_ok && (m > 0 && nref > 0)) { if (lane == 0) atomicAdd(&ctr[C_ERR_PAIRS], 1ull); cap_err = true; }
          SwRes fw; fw.score = 0; fw.end_ref = -1; fw.end_read = m - 1;
          int ce = -1;
          if (sw_ok) {
            if (!rdq_staged) { __syncthreads(); for (uint32_t q = lane; q < len; q += 64) RD[q] = (uint8_t)read_nt(rec, len, q, w.reversed, w.aval); rdq_staged = true; __syncthreads(); }
            for (int e = 0; e < n_cached; e++)
              if (ctk[e].max_ref == tk.max_ref && ctk[e].rf_start == tk.rf_start && ctk[e].align_que_start == tk.align_que_start && ctk[e].m == m && ctk[e].nref == nref) { ce = e; break; }
            if (ce < 0) {
              // a new batch: this task, and the tasks the walk would reach next if this one and they do not align
              __builtin_amdgcn_wave_barrier();
              ctk[0] = tk; n_cached = 1;
              for (int e = 0; e < 4; e++) cslot[e] = e;
              if (x4_ok) {
                Walk L = R; L.is_aligned = 0;
                while (n_cached < 4) {
                  SwTask t2;
                  if (advance(L, false, t2) != 1) break;
                  if (!task_fits(t2)) break;
                  ctk[n_cached++] = t2; L.is_aligned = 0;
                }
              }
              __syncthreads();
              bool hasn = false;
              if (!long_rd) for (int e = 0; e < n_cached; e++) {
                uint8_t* dst = wslot(e);
                for (int q = lane; q < ctk[e].nref; q += 64) { const uint8_t ch = ix.ref_seq[ctk[e].rf_start + q]; dst[q] = ch; hasn |= ch == 4; }
              }
              __syncthreads();
              TPH(5)
              if (n_cached >= 2) {
                const int g = lane >> 4;
                const bool mine = g < n_cached;
                const int gm = mine ? (g == 0 ? ctk[0].m : g == 1 ? ctk[1].m : g == 2 ? ctk[2].m : ctk[3].m) : 0;
                const int gn = mine ? (g == 0 ? ctk[0].nref : g == 1 ? ctk[1].nref : g == 2 ? ctk[2].nref : ctk[3].nref) : 0;
                const int gq = mine ? (int)(g == 0 ? ctk[0].align_que_start : g == 1 ? ctk[1].align_que_start : g == 2 ? ctk[2].align_que_start : ctk[3].align_que_start) : 0;
                int mm = gm;
                for (int d = 32; d > 0; d >>= 1) mm = max(mm, __shfl_xor(mm, d, 64));
                const SwRes r4 = sw_wave_x4(rdq, gm, gq, 1, wslot(g), gn, 0, 1, P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext, mm, __any(hasn));
                for (int e = 0; e < n_cached; e++) {
                  cfw[e].score = __builtin_amdgcn_readlane(r4.score, 16 * e); cfw[e].end_ref = __builtin_amdgcn_readlane(r4.end_ref, 16 * e);
                  cfw[e].end_read = __builtin_amdgcn_readlane(r4.end_read, 16 * e);
                }
                if (n_cached > 1) n_spec += (unsigned long long)(n_cached - 1);
              } else {
                if (LONG) cfw[0] = sw_wave_any_t<STRIPED>(RD, m, (int)tk.align_que_start, 1, long_rd ? ix.ref_seq + tk.rf_start : rfq, nref, 0, 1, bound, P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext, P.sw_mode, 0, 0, sw_scr(P));
                else cfw[0] = sw_wave_t<STRIPED>(rdq, m, (int)tk.align_que_start, 1, rfq, nref, 0, 1, bound, P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext, P.sw_mode, 0, 0, sw_scr(P));
              }
              __syncthreads();
              TPH(6)
              ce = 0;
            }
            else n_spec_used++;
            fw = cfw[ce];
            n_fwd++; n_cells += (unsigned long long)m * nref;
          }
          const uint64_t align_ref_start = tk.align_ref_start, head = tk.head, align_que_start = tk.align_que_start;
          const uint32_t max_ref = tk.max_ref;
          int score1 = fw.score > 65535 ? 65535 : fw.score;
          const int ref_end1 = fw.end_ref, read_end1 = fw.end_read;
          // The begin cell (ssw_align's reverse pass, ssw.c:900-918) does not influence the walk -- it is only stored -- so it is not
          // computed here: an accepted alignment is recorded with the START OF ITS WINDOW in ref_begin1 / read_begin1 and has_cigar = 2
          // ("begin pending"), and k_begins computes the begins of the alignments that are still stored when the part is done, four per wave.
          R.is_aligned = (sw_ok && (uint32_t)score1 > P.minimal_score);     // strict (:388)
          if (R.is_aligned) {
            if ((uint32_t)score1 == max_SW_score) ++st.max_SW_count;
            AlignRec al;
            al.ref_begin1 = (int32_t)(align_ref_start - head);
            al.ref_end1 = ref_end1 + (int32_t)(align_ref_start - head);
            al.read_begin1 = (int32_t)align_que_start;
            al.read_end1 = read_end1 + (int32_t)align_que_start;
            al.readlen = len; al.ref_num = max_ref;
            al.index_num = (uint16_t)P.index_num; al.part = (uint16_t)P.part;
            al.strand = (uint8_t)!w.reversed; al.score1 = (uint16_t)score1;
            al.has_cigar = 2; al.cigar_off = 0; al.cigar_len = P.sw_mode < 0 ? (uint32_t)fw.word : 0u;
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
              if (P.is_best) { if (P.num_alignments == st.max_SW_count) R.go_on = 0; }
              else if (P.num_alignments == st.n_align) R.go_on = 0;
            }
            search = 0;
          }
        }
        w.best = R.best;
      }
    }
    TPH(5)
    if (!parked) finish_read(r, st, w, search, lane == 0);
  }
  if (lane == 0) {
    if (n_fwd) ctr_add(ctr, C_SW_FWD, n_fwd);
    if (n_cells) ctr_add(ctr, C_SW_CELLS, n_cells);
    if (n_spec) atomicAdd(&ctr[C_SW_SPEC], n_spec);
    if (n_spec_used) atomicAdd(&ctr[C_SW_SPEC_USED], n_spec_used);
#if defined(SMR_CHAIN_PHASES) || defined(SMR_CHAIN_STATS)
    for (int q = 0; q < 7; q++) if (tph[q]) atomicAdd(&ctr[C_SHARDS + (blockIdx.x & (C_NSHARD - 1)) * C_SHARD_W + C_SHARD_PH + q], tph[q]);
#endif
  }
}


// ------------------------------------------------------------------------------------------------
// k_begins: the begin cells of the stored alignments (ssw_align's reverse pass, ssw.c:900-918: the same recurrence on the reversed
// prefixes that end in the forward end cell; first column reaching the maximum, smallest row).  k_chain records an accepted alignment
// with the start of its SW window in ref_begin1 / read_begin1 and has_cigar = 2; best-N bookkeeping may replace it before the part is
// done, so only the survivors cost a reverse pass (43 % of the accepted ones on the bench workload), and they are independent problems:
// four per wave through the four-problem kernel (reads <= SW_X4_MAX_ROWS), else one per wave through sw_wave.
// k_begins_collect lists the pending slots of the reads with a new hit; k_begins claims them four at a time.
// Dynamic LDS: 4 read windows of lds_m bytes | 4 reference windows of lds_n bytes (x4); 1 + 1 in single-problem mode of one-strip reads; none for LONG.
// ------------------------------------------------------------------------------------------------
__global__ void k_begins_collect(uint32_t n, uint32_t slots, const RState* __restrict__ work, const RWork* __restrict__ rw, const AlignRec* __restrict__ work_aln,
                                 uint32_t* __restrict__ tasks, unsigned long long* __restrict__ ctr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool take = false;
  if (i < n * slots) {
    const uint32_t r = i / slots, k = i % slots;
    take = rw[r].is_new_hit && k < work[r].n_align && work_aln[i].has_cigar >= 2;
  }
  const uint32_t o = block_append(&ctr[C_BEGIN_N], take);
  if (take) tasks[o] = i;
}

template <bool LONG, bool STRIPED = false>
__global__ void __launch_bounds__(64, (LONG || STRIPED) ? 3 : 4) k_begins(DReads rd, DIndex ix, DParams P, const uint32_t* __restrict__ tasks, AlignRec* __restrict__ work_aln,
                                               unsigned long long* __restrict__ ctr, uint32_t lds_m, uint32_t lds_n, int x4, int* g_bound, uint8_t* g_rdq) {
  SMR_DYN_LDS(unsigned char, lds_raw);
  __shared__ uint32_t s_t0;
  const int lane = lane_id();
  const uint32_t n_tasks = (uint32_t)ctr[C_BEGIN_N];
  const int per = x4 ? 4 : 1;
  const int g = x4 ? lane >> 4 : 0;
  // four-problem mode: 4 read + 4 reference windows in LDS; single-problem mode: one of each in LDS, or -- LONG: the batch has reads of more
  // than one strip -- the read's letters in this block's row of g_rdq and the reference window read in place
  const bool in_lds = x4 || !LONG;
  uint8_t* rdq = in_lds ? lds_raw + (size_t)g * lds_m : g_rdq + (size_t)blockIdx.x * lds_m;
  uint8_t* rfq_l = lds_raw + (size_t)per * lds_m + (size_t)g * lds_n;
  int* bound = g_bound ? g_bound + (size_t)blockIdx.x * 2 * lds_n : nullptr;          // strip boundaries (single-problem mode, reads of more than one strip)
  unsigned long long n_rev = 0, n_cells = 0;
  uint32_t c_next = 0, c_end = 0;                            // tasks claimed and not yet taken
  for (;;) {
    __syncthreads();
    if (c_next >= c_end) {                                  // (eight rounds per claim: a returning atomic per round of four tasks made the counter a queue -- 200 000 per bench step at 83 per microsecond)
      if (lane == 0) s_t0 = (uint32_t)atomicAdd(&ctr[C_BEGIN_NEXT], (unsigned long long)(8 * per));
      __syncthreads();
      c_next = s_t0; c_end = c_next + 8u * (uint32_t)per;
    }
    const uint32_t t0 = c_next;
    c_next += (uint32_t)per;
    if (t0 >= n_tasks) break;
    const uint32_t t = t0 + (uint32_t)g;
    const bool have = t < n_tasks;
    uint32_t slot = 0;
    AlignRec al;
    int m = 0, n = 0;
    const uint8_t* rfq = rfq_l;
    if (have) {
      slot = tasks[t]; al = work_aln[slot];
      m = al.read_end1 - al.read_begin1 + 1; n = al.ref_end1 - al.ref_begin1 + 1;       // the window prefixes that end in the forward end cell
      const uint32_t r = slot / P.slots, len = rd.len[r];
      const uint32_t* rec = rd.words + rd.rec_off[r];
      const uint8_t* ref = ix.ref_seq + ix.ref_off[al.ref_num] + al.ref_begin1;
      const int l0 = x4 ? (lane & 15) : lane, ls = x4 ? 16 : 64;
      for (int q = l0; q < m; q += ls) rdq[q] = (uint8_t)read_nt(rec, len, (uint32_t)(al.read_begin1 + q), al.strand ? 0u : 1u, 4u);
      if (in_lds) for (int q = l0; q < n; q += ls) rfq_l[q] = ref[q];
      rfq = in_lds ? rfq_l : ref;
    }
    bool hasn = false;
    __syncthreads();
    if (have && x4) { const int l0 = lane & 15; for (int q = l0; q < n; q += 16) hasn |= rfq[q] == 4; }
    // has_cigar = 3 (smr_walk.hpp: an alignment of which only the score was asked): what is stored are the ENDS OF ITS WINDOW; the forward pass over
    // the window finds the end cell (first column reaching the maximum, smallest row: ssw.c:305-336), then the reverse pass as for the others
    const bool fwd = have && al.has_cigar == 3;
    if (__any(fwd)) {
      SwRes fw;
      if (x4) {
        int mf = fwd ? m : 0;
        int mm = mf;
        for (int d = 32; d > 0; d >>= 1) mm = max(mm, __shfl_xor(mm, d, 64));
        fw = sw_wave_x4(rdq, mf, 0, 1, rfq, fwd ? n : 0, 0, 1, P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext, mm, __any(hasn));
      } else {
        if (LONG) fw = sw_wave_any_t<false>(rdq, m, 0, 1, rfq, n, 0, 1, bound, P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext, P.sw_mode);
        else fw = sw_wave(rdq, m, 0, 1, rfq, n, 0, 1, bound, P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext, P.sw_mode);
      }
      if (fwd) {
        if ((fw.score > 65535 ? 65535 : fw.score) != (int)al.score1 && (x4 ? (lane & 15) == 0 : lane == 0)) atomicAdd(&ctr[C_ERR_TRACE], 1ull);     // (cannot happen: same cells, same recurrence)
        al.ref_end1 = al.ref_begin1 + fw.end_ref; al.read_end1 = al.read_begin1 + fw.end_read;
        m = fw.end_read + 1; n = fw.end_ref + 1;
      }
      __syncthreads();
    }
    SwRes bw;
    if (x4) {
      int mm = m;
      for (int d = 32; d > 0; d >>= 1) mm = max(mm, __shfl_xor(mm, d, 64));
      bw = sw_wave_x4(rdq, m, m - 1, -1, rfq, n, n - 1, -1, P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext, mm, __any(hasn));
    } else {
      // (the striped slow path stops where the forward score is reached, in the kernel the forward pass ended with: kept in cigar_len while the begin is pending)
      if (LONG) bw = sw_wave_any_t<STRIPED>(rdq, m, m - 1, -1, rfq, n, n - 1, -1, bound, P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext, P.sw_mode, (int)al.score1, (int)al.cigar_len, sw_scr(P));
      else bw = sw_wave_t<STRIPED>(rdq, m, m - 1, -1, rfq, n, n - 1, -1, bound, P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext, P.sw_mode, (int)al.score1, (int)al.cigar_len, sw_scr(P));
    }
    if (have && (x4 ? (lane & 15) == 0 : lane == 0)) {
      al.ref_begin1 = al.ref_end1 - bw.end_ref;
      al.read_begin1 = al.read_end1 - bw.end_read;
      al.has_cigar = 0; al.cigar_len = 0;
      work_aln[slot] = al;
    }
    if (have) { n_rev += (x4 ? (lane & 15) == 0 : lane == 0) ? 1 : 0; n_cells += (x4 ? (lane & 15) == 0 : lane == 0) ? (unsigned long long)m * n : 0; }
  }
  for (int d = 32; d > 0; d >>= 1) { n_rev += __shfl_xor(n_rev, d, 64); n_cells += __shfl_xor(n_cells, d, 64); }
  if (lane == 0) { if (n_rev) ctr_add(ctr, C_SW_REV, n_rev); if (n_cells) ctr_add(ctr, C_SW_CELLS, n_cells); }
}

}  // namespace smr
