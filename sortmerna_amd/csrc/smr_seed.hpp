// smr_seed.hpp -- part of the HIP kernels of libsmr_hip (included by smr_kernels.hpp).
#pragma once

namespace smr {

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

// The LEV(1) tables as 64-bit rows: row r holds the next state of every state s in nibble s, so a transition is one
// shift once the row is known -- and the row only depends on (depth, nucleotide, the window's bit-vectors), not on the
// state, so all rows of an entry can be fetched up front and the state chain runs in registers.
// rows 0..15 = t0[bit-vector], 16..23 = t1, 24..27 = t2, 28..29 = t3; nibble 14 (dead) maps to 14.
#define LEV_ROWS 30
__device__ __forceinline__ void build_lev_rows(unsigned long long* s_row) {
  const uint32_t i = threadIdx.x;
  if (i < LEV_ROWS) {
    unsigned long long r = (14ull << 56) | (14ull << 60);
    for (uint32_t st = 0; st < 14; st++) r |= (unsigned long long)c_lev[i * 14 + st] << (4 * st);
    s_row[i] = r;
  }
}
__device__ __forceinline__ uint32_t lev_row_index(const BitVec& bv, uint32_t depth, uint32_t nt, uint32_t pw) {
  if (depth < pw - 2) return bv.get(depth, nt);                 // traverse_bursttrie.cpp:131-135
  const uint32_t t = 3 - pw + depth;                            // 1,2,3 (:136-139)
  const uint32_t v = bv.get(pw - 3, nt) & ((2u << (pw - depth)) - 1u);
  return (t == 1 ? 16u : (t == 2 ? 24u : 28u)) + v;
}
__device__ __forceinline__ uint32_t lev_next(unsigned long long row, uint32_t state) { return (uint32_t)(row >> (4 * state)) & 15u; }

// The LEV(1) automaton over a COMPLETE candidate string in closed form (tests/test_lev_closed_form.py proves it equal to
// the table automaton for every seed length): P = the window's pw automaton chars, T = the pw+1 chars of trie path +
// bucket tail (2 bits per char, char i at bits 2i).  With a = common prefix length and s0/s1/s2 = trailing equal chars
// of P vs T, P vs T>>1 char, P>>1 char vs T:  accepted  <=>  a+s2 >= pw-1 (at depth pw-2)  or  a+s0 >= pw-1 (depth pw-1)
// or  a+s1 >= pw (depth pw);  0-error match (state 9 at depth pw-1)  <=>  a >= pw, and then it was accepted at pw-2.
// returns bit 0 = accepted, bit 1 = 0-error match
__device__ __forceinline__ uint32_t lev1_entry(uint32_t P, uint32_t T, uint32_t pw) {
  const uint32_t m2 = (1u << (2 * pw)) - 1u, m2b = m2 >> 2, ev = 0x55555555u;
  const uint32_t x0 = P ^ T, x1 = P ^ (T >> 2), x2 = (P >> 2) ^ T;
  const uint32_t d0 = (x0 | (x0 >> 1)) & ev & m2;            // bit 2i set iff chars i differ
  const uint32_t d1 = (x1 | (x1 >> 1)) & ev & m2;
  const uint32_t d2 = (x2 | (x2 >> 1)) & ev & m2b;
  const uint32_t a = (uint32_t)__builtin_ctz(d0 | (1u << (2 * pw))) >> 1;
  const uint32_t s0 = min((uint32_t)__clz((int)(d0 << (32 - 2 * pw))) >> 1, pw);
  const uint32_t s1 = min((uint32_t)__clz((int)(d1 << (32 - 2 * pw))) >> 1, pw);
  const uint32_t s2 = min((uint32_t)__clz((int)(d2 << (34 - 2 * pw))) >> 1, pw - 1);
  const bool acc = (a + s2 >= pw - 1) || (a + s0 >= pw - 1) || (a + s1 >= pw);
  return (acc ? 1u : 0u) | (a >= pw ? 2u : 0u);
}

// Is the automaton still alive after the first m chars of T (m = depth + 1 <= pw - 1 at trie nodes)?  Closed form, proven
// equal to "state != 14" in tests/test_lev_closed_form.py: with a = common prefix of (P, T[0..m)), alive <=> a >= m, or the
// rest matches after ONE edit at position a: T[i]==P[i] (substitution), T[i]==P[i-1] (extra char in T) for i in (a, m),
// or T[i]==P[i+1] for i in [a, m) (char of P skipped).
__device__ __forceinline__ bool lev1_alive(uint32_t P, uint32_t T, uint32_t m) {
  const uint32_t mm = (1u << (2 * m)) - 1u, ev = 0x55555555u;
  const uint32_t x0 = P ^ T, x1 = (P << 2) ^ T, x2 = (P >> 2) ^ T;
  const uint32_t d0 = (x0 | (x0 >> 1)) & ev & mm;
  const uint32_t a2 = (uint32_t)__builtin_ctz(d0 | (1u << (2 * m)));        // 2 * common prefix length
  const uint32_t d1 = (x1 | (x1 >> 1)) & ev & mm, d2 = (x2 | (x2 >> 1)) & ev & mm;
  return d0 == 0 || (d0 >> (a2 + 2)) == 0 || (d1 >> (a2 + 2)) == 0 || (d2 >> a2) == 0;
}

// ------------------------------------------------------------------------------------------------
// The seed stage of one (strand, pass): window scan + burst-trie descent, organised as a sort-merge join.
//
// The reference probes, per window, lookup_tbl[9-mer] and walks that mini burst trie (paralleltraversal.cpp:124-249,
// traverse_bursttrie.cpp:100-298): a hash-scatter into a structure of GBs.  Here the windows of the WHOLE batch are
// first sorted by their 9-mer key (two-level counting sort, below), and the searches run in
// key order, 64 consecutive tuples per wave: neighbouring lanes walk the same or adjacent mini-tries, so their node
// and bucket loads hit the same cache lines.  The forward half-seed searches of all windows run first (phase F), then
// the reverse searches of the windows whose forward search did not end with a 0-error match (phase R,
// paralleltraversal.cpp:188), seeded with the forward hit list so that the reference's in-order de-duplication rules
// are applied exactly.
//
//   k_seed_keys         window -> forward and reverse (key, payload)  [9-mer hash, lookup probes, flip34 view]
//   k_seed_cscan/split/bins   tuples to key order
//   k_seed_pg<DIR>      the searches over the pigeonhole layout (smr_seed_pg.hpp) -- the default
//   k_seed_search<DIR>  the searches, per-lane DFS formulation (below): overflow redo + exact work counters
//   k_seed_finish       per read: gather the windows' hits into one block, hit_seeds / hit_total (paralleltraversal.cpp:242-249)
//
// k_seed_search, one wave per 64 tuples: lane = one window's search.  A lane walks its mini-trie exactly in the
// reference's DFS order (A<C<G<T, traverse_bursttrie.cpp:117), but the walk is cut into ROUNDS: in a round every lane
// advances over trie NODES only and collects its next few buckets; then the whole wave scans the entries of all
// collected buckets together, one lane per ENTRY (lev1_entry: the automaton over a complete candidate string in closed
// form), so the dominant work runs with full lanes.  Accepted entries ("candidates", rare) are handed back to the owning
// lane in entry order, which applies the reference's sequential rules to its lane-local hit list in LDS:
//   entry accepted at t_a = first step with depth_b >= pw-2 and state >= 8   (traverse_bursttrie.cpp:229-235)
//   COND    0-error entry (state 9 at depth_b == pw-1): it is accepted one step earlier, at pw-2, and the reference reaches
//           the 0-error step only if the id was NOT already in the list then (otherwise the duplicate check `break`s
//           first, :265-277); when it fires: list = {id}, search over (:256-262)
//   PLAIN   1-error hit: appended unless the id is already present
//   (UNCOND, state 9 in the accepting step itself, cannot occur: tests/test_lev_closed_form.py)
// ------------------------------------------------------------------------------------------------
#define SEED_STK 10                                    // trie depth < partialwin - 1 <= 9
#define SEED_OWN_CAP 2048u                             // entries of one round that get a direct entry -> bucket byte map
#define SEED_MAXPW 10u
#define SEED_K 4                                       // buckets a lane may collect per round
#define SEED_GATHER 32u                                // ... or until it holds this many entries
// dynamic LDS words of k_seed_search: hit lists, node stack, row-index table, pref/pb/pth, patterns, owner map
#define SEED_LDS_WORDS(hcap) (64u * (hcap) + SEED_STK * 64u + (SEED_MAXPW + 1u) * 64u + 3u * 64u * SEED_K + 64u + SEED_OWN_CAP / 4u)
#define SEED_ZERO_BIT 0x80000000u

enum { CK_PLAIN = 0, CK_UNCOND = 1, CK_COND = 2 };
enum { SN_TUPLES = 0, SN_REDO = 1, SN_FWD = 2, SN_COUNT = 4 };      // device counters of the seed stage (u32): tuples, redo waves, forward tuples

struct SeedTmp { uint32_t key, lo, hi; };               // 12 bytes; payload = lo | hi << 32: read | win_pos << 24 | chars << 40
__device__ __forceinline__ unsigned long long seed_payload(const SeedTmp& t) { return (unsigned long long)t.lo | ((unsigned long long)t.hi << 32); }

#define SEED_SPLIT_CHUNK 32768u                           // tuples per block of k_seed_split
struct SeedBufs {
  uint32_t* chist;           // [nc + 1] tuples per COARSE bin (key >> fb)
  uint32_t* cbase;           // [nc + 1] exclusive scan of chist
  uint32_t* ccur;            // [nc] allocation cursors of the coarse bins (k_seed_split)
  SeedTmp* tmp;              // unsorted tuples
  SeedTmp* mid;              // tuples grouped by coarse bin
  SeedTmp* srt;              // tuples in key order
  uint32_t* wseg;            // [maxwin][n] (window-major: k_seed_finish's threads = reads read it coalesced) pool offset of the window's hit segment | SEED_ZERO_BIT ; NONE = no hits
  uint32_t* sn;              // SN_* counters
  uint32_t* redo;            // waves of k_seed_pg to be searched again by k_seed_search
  uint32_t nk, nkh, maxwin, cap_tuples, cap_redo;     // nk = 2 * nkh bins: forward keys [0, nkh), reverse keys [nkh, 2 nkh)
  uint32_t fb, nc;           // fine bits (min(9, L): a coarse bin never mixes forward and reverse keys), nc = nk >> fb coarse bins (<= 4096)
  uint32_t n;                // reads in the batch
};
__device__ __forceinline__ size_t wseg_slot(const SeedBufs& sb, uint32_t r, uint32_t k) { return (size_t)k * sb.n + r; }

// nbits <= 40 bits starting at bit `bit0` of a little-endian word stream (reads up to 2 words past the first)
__device__ __forceinline__ unsigned long long extract_bits(const uint32_t* w, uint32_t bit0, uint32_t nbits) {
  const uint32_t i = bit0 >> 5, s = bit0 & 31u;
  unsigned long long v = ((unsigned long long)w[i] | ((unsigned long long)w[i + 1] << 32)) >> s;
  if (s) v |= (unsigned long long)w[i + 2] << (64 - s);
  return v & ((1ull << nbits) - 1ull);
}

// window content of a read in the CURRENT strand/encoding state: 2 bits per nt, char i at bits 2i.
// Fast path: one 2L-bit extraction from the packed record (reverse strand: 2-bit groups reversed and complemented,
// Read::revIntStr read.cpp:350-357); windows touching an ambiguous letter take the per-letter path.
__device__ __forceinline__ unsigned long long window_chars(const uint32_t* rec, uint32_t len, uint32_t win_pos, uint32_t L,
                                                           uint32_t reversed, uint32_t aval) {
  const uint32_t cw = (len + 15) >> 4;
  const uint32_t j0 = reversed ? (len - win_pos - L) : win_pos;           // first read position covered, ascending
  if (extract_bits(rec + cw, j0, L) == 0ull) {
    unsigned long long x = extract_bits(rec, 2 * j0, 2 * L);
    if (reversed) {
      x = __brevll(x) >> (64 - 2 * L);                                      // reverse the bit order ...
      x = ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);   // ... and restore each 2-bit group
      x = ~x & ((1ull << (2 * L)) - 1ull);                                   // complement: 3 - code
    }
    return x;
  }
  unsigned long long wchars = 0;
  for (uint32_t i = 0; i < L; i++) wchars |= (unsigned long long)read_nt(rec, len, win_pos + i, reversed, aval) << (2 * i);
  return wchars;
}

// Both half-seed searches of a window become tuples in ONE pass (so the counting sort runs once per stage):
//   forward: key = first 9-mer, automaton fed by chars [pw, 2pw)          (init_win_f bitvector.cpp:57-91)  -> bins [0, nkh)
//   reverse: key = second 9-mer, automaton fed by chars pw-1 .. 0         (init_win_r :99-132)              -> bins [nkh, 2 nkh)
// The reverse tuple is speculative: the reverse search kernel drops it when the forward search ended with a 0-error match
// (accept_zero_kmer, paralleltraversal.cpp:188).
__global__ void __launch_bounds__(1024) k_seed_keys(DReads rd, DIndex ix, DParams P, int pass, SeedBufs sb,
                                                   const RWork* __restrict__ rw, unsigned long long* __restrict__ ctr, uint32_t n_tiles) {
  SMR_DYN_LDS(uint32_t, lh);                              // [nc] this block's tuples per coarse bin
  __shared__ uint32_t s_cnt[2][16], s_off[2][16], s_win[16];
  __shared__ unsigned long long s_bytes;                  // algorithmic input bytes of this block (C_B_KEYS)
  for (uint32_t c = threadIdx.x; c < sb.nc; c += blockDim.x) lh[c] = 0;
  if (threadIdx.x == 0) s_bytes = 0;
  __syncthreads();
  const int lane = lane_id();
  const uint32_t pw = P.partialwin, L = P.lnwin;
  const uint32_t wv = threadIdx.x >> 6;
  for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint32_t tid = tile * blockDim.x + threadIdx.x;
    const uint32_t r = tid / sb.maxwin, k = tid % sb.maxwin;
    bool emit[2] = {false, false};
    uint32_t key[2] = {0, 0}, is_win = 0, in_bytes = 0;
    unsigned long long payload[2] = {0, 0};
    if (r < rd.n) {
      const RWork w = rw[r];
      const bool active = (w.strand_active && w.search && w.pass_n == (uint32_t)pass);
      const uint32_t len = rd.len[r];
      // inputs, once per read (its first window's thread): per-read state + length; of an active read also the record offset and the packed record
      if (k == 0) in_bytes = (uint32_t)sizeof(RWork) + 4u + (active ? 8u + 4u * (((len + 15) >> 4) + ((len + 31) >> 5)) : 0u);
      const uint32_t stride = P.skip[pass];
      const uint32_t numwin = active ? (len - L + stride) / stride : 0;       // paralleltraversal.cpp:118-120
      bool mine = k < numwin;
      const uint32_t win_pos = k * stride;
      if (mine) for (int q = 0; q < pass; q++) if (win_pos % P.skip[q] == 0) mine = false;   // read_pos_searched (:128-131)
      // (wseg starts as NONE everywhere: launch_seed fills it)
      if (mine) {
        // traverse(): `if (read.is04) read.flip34()` before every window (:126) -> ambiguous positions read as 0 / 3
        const uint32_t aval = w.is04 ? 0 : w.aval;
        const unsigned long long wc = window_chars(rd.words + rd.rec_off[r], len, win_pos, L, w.reversed, aval);
        const unsigned long long half = (1ull << (2 * pw)) - 1ull;
        // first / second 9-mer with char i at bits 2i; hashKmer is MSB-first (read.cpp:601-611) = the 2-bit groups reversed
        const uint32_t a = (uint32_t)(wc & half), b = (uint32_t)((wc >> (2 * pw)) & half);
        uint32_t ra = __brev(a) >> (32 - 2 * pw), rb = __brev(b) >> (32 - 2 * pw);
        ra = ((ra & 0x55555555u) << 1) | ((ra >> 1) & 0x55555555u);
        rb = ((rb & 0x55555555u) << 1) | ((rb >> 1) & 0x55555555u);
        is_win = 1; in_bytes += 8;                                  // two lookup words
        const uint32_t lf = ix.lkc[ra], lr = ix.lkc[rb];          // lookup_tbl[kmer].count and the presence of trie_F / trie_R (paralleltraversal.cpp:155-160, 186-192)
        emit[0] = (lf & 0x3FFFFFFFu) > P.minoccur && ((lf >> 30) & 1u);
        emit[1] = (lr & 0x3FFFFFFFu) > P.minoccur && (lr >> 31);
        key[0] = ra; key[1] = sb.nkh + rb;
        const unsigned long long rw_ = (unsigned long long)r | ((unsigned long long)win_pos << 24);
        payload[0] = rw_ | ((unsigned long long)b << 40);            // forward: second half in order
        payload[1] = rw_ | ((unsigned long long)ra << 40);           // reverse: first half walked backwards
      }
    }
    // block-aggregated slot allocation in the unsorted tuple array (one atomic per 1024 slots)
    const unsigned long long em0 = __ballot(emit[0]), em1 = __ballot(emit[1]), wm = __ballot(is_win);
    for (int d = 32; d > 0; d >>= 1) in_bytes += __shfl_xor(in_bytes, d, 64);
    if (lane == 0) { s_cnt[0][wv] = (uint32_t)__popcll(em0); s_cnt[1][wv] = (uint32_t)__popcll(em1); s_win[wv] = (uint32_t)__popcll(wm); if (in_bytes) atomicAdd(&s_bytes, (unsigned long long)in_bytes); }
    __syncthreads();
    if (threadIdx.x == 0) {                                // the tile's forward tuples first, wave by wave, then its reverse tuples
      uint32_t tc = 0, tw = 0;
      for (int d = 0; d < 2; d++) for (uint32_t q = 0; q < (blockDim.x >> 6); q++) { s_off[d][q] = tc; tc += s_cnt[d][q]; }
      for (uint32_t q = 0; q < (blockDim.x >> 6); q++) tw += s_win[q];
      const uint32_t base = tc ? atomicAdd(&sb.sn[SN_TUPLES], tc) : 0u;
      for (int d = 0; d < 2; d++) for (uint32_t q = 0; q < (blockDim.x >> 6); q++) s_off[d][q] += base;
      if (tw) { ctr_add(ctr, C_WINDOWS, tw); ctr_add(ctr, C_LOOKUP, tw); }     // the forward lookups; the reverse ones are counted by k_seed_finish
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 2; d++) {
      if (emit[d]) {
        const uint32_t idx = s_off[d][wv] + (uint32_t)__popcll((d ? em1 : em0) & ((1ull << lane) - 1));
        if (idx < sb.cap_tuples) {
          SeedTmp t; t.key = key[d]; t.lo = (uint32_t)payload[d]; t.hi = (uint32_t)(payload[d] >> 32);
          sb.tmp[idx] = t;
          atomicAdd(&lh[key[d] >> sb.fb], 1u);
        }
      }
    }
    __syncthreads();                                       // s_cnt / s_off are rewritten by the next tile
  }
  for (uint32_t c = threadIdx.x; c < sb.nc; c += blockDim.x) if (lh[c]) atomicAdd(&sb.chist[c], lh[c]);
  if (threadIdx.x == 0 && s_bytes) ctr_add(ctr, C_B_KEYS, s_bytes);
}

// The tuples are brought into key order by a two-level counting sort whose per-tuple atomics all stay in LDS:
//   k_seed_keys    counts the tuples per COARSE bin (key >> fb, <= 4096 bins) in LDS while it writes them unsorted
//   k_seed_cscan   exclusive scan of the coarse counts (one block)
//   k_seed_split   a block takes SEED_SPLIT_CHUNK unsorted tuples: LDS histogram by coarse bin, one global atomic per non-empty bin reserves
//                  the block's share of that bin, second pass copies the tuples there (runs of ~32 tuples per bin)
//   k_seed_bins    one block per coarse bin: LDS histogram of the fine key bits (<= 512 bins), scan, second pass writes payload and key to
//                  their final places
// (One counting sort over all 2 * 4^pw keys with a returning global atomic per tuple took 2.4 + 1.9 ms per stage of 60 M tuples on the
// MI355X; this takes 1.3 + 0.9 + 1.0 ms.)
__global__ void __launch_bounds__(1024) k_seed_cscan(SeedBufs sb, unsigned long long* __restrict__ ctr) {
  __shared__ uint32_t s_part[16];
  // <= 4096 bins: 4 consecutive bins per thread
  const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  uint32_t v[4], sum = 0;
  for (int q = 0; q < 4; q++) { const uint32_t c = 4 * t + q; v[q] = c < sb.nc ? sb.chist[c] : 0u; sum += v[q]; }
  uint32_t incl = sum;
  for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d, 64); if ((int)lane >= d) incl += x; }
  if (lane == 63) s_part[wv] = incl;
  __syncthreads();
  uint32_t pre = incl - sum;
  for (uint32_t q = 0; q < wv; q++) pre += s_part[q];
  for (int q = 0; q < 4; q++) {
    const uint32_t c = 4 * t + q;
    if (c < sb.nc) { sb.cbase[c] = pre; sb.ccur[c] = pre; if (c == (sb.nkh >> sb.fb)) sb.sn[SN_FWD] = pre; }
    pre += v[q];
    if (c + 1 == sb.nc) sb.cbase[sb.nc] = pre;
  }
  __syncthreads();
  if (t == 0) {                                             // the stage's tuples: the forward ones lie in front of coarse bin nkh >> fb
    const uint32_t all = min(sb.sn[SN_TUPLES], sb.cap_tuples), fw = min(sb.cbase[sb.nkh >> sb.fb], all);
    ctr_add(ctr, C_TUP_F, fw); ctr_add(ctr, C_TUP_R, all - fw);
  }
}

__global__ void __launch_bounds__(1024) k_seed_split(SeedBufs sb) {
  SMR_DYN_LDS(uint32_t, lds);
  uint32_t* lh = lds;                                     // [nc] count, then running offset inside the block's share
  uint32_t* lb = lds + sb.nc;                             // [nc] start of the block's share of the coarse bin
  const uint32_t n = min(sb.sn[SN_TUPLES], sb.cap_tuples);
  const uint32_t i0 = blockIdx.x * SEED_SPLIT_CHUNK, i1 = min(i0 + SEED_SPLIT_CHUNK, n);
  if (i0 >= n) return;
  for (uint32_t c = threadIdx.x; c < sb.nc; c += blockDim.x) lh[c] = 0;
  __syncthreads();
  for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) atomicAdd(&lh[sb.tmp[i].key >> sb.fb], 1u);
  __syncthreads();
  for (uint32_t c = threadIdx.x; c < sb.nc; c += blockDim.x) { const uint32_t m = lh[c]; if (m) lb[c] = atomicAdd(&sb.ccur[c], m); lh[c] = 0; }
  __syncthreads();
  for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
    const SeedTmp t = sb.tmp[i];
    const uint32_t c = t.key >> sb.fb;
    sb.mid[lb[c] + atomicAdd(&lh[c], 1u)] = t;
  }
}

__global__ void __launch_bounds__(1024) k_seed_bins(SeedBufs sb) {
  __shared__ uint32_t fh[512], s_part[8];
  const uint32_t c = blockIdx.x, lo = sb.cbase[c], hi = sb.cbase[c + 1];
  if (lo == hi) return;
  const uint32_t t = threadIdx.x, fm = (1u << sb.fb) - 1u;
  if (t < 512) fh[t] = 0;
  __syncthreads();
  for (uint32_t i = lo + t; i < hi; i += blockDim.x) atomicAdd(&fh[sb.mid[i].key & fm], 1u);
  __syncthreads();
  // exclusive scan of the 512 fine counts (waves 0..7)
  uint32_t v = 0, incl = 0;
  if (t < 512) {
    v = fh[t]; incl = v;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d, 64); if ((int)(t & 63u) >= d) incl += x; }
    if ((t & 63u) == 63u) s_part[t >> 6] = incl;
  }
  __syncthreads();
  if (t < 512) {
    uint32_t pre = incl - v;
    for (uint32_t q = 0; q < (t >> 6); q++) pre += s_part[q];
    fh[t] = lo + pre;                                     // from now on: the next free place of the fine bin
  }
  __syncthreads();
  for (uint32_t i = lo + t; i < hi; i += blockDim.x) {
    const SeedTmp x = sb.mid[i];
    const uint32_t p = atomicAdd(&fh[x.key & fm], 1u);
    sb.srt[p] = x;
  }
}

struct SeedLane {           // per-lane search result
  uint32_t nh; bool zero, overflow; uint32_t n_node, n_entry;
};

// LDS of one search wave (32-bit words unless noted)
struct SeedLds {
  uint32_t* hl;        // [hcap][64]   lane-local hit lists (ids), element k of lane l at k*64+l
  uint32_t* stk;       // [SEED_STK][64] DFS stack: node offset | pending-element mask << 22 | state the node was entered with << 26
  uint32_t* rt;        // [SEED_MAXPW+1][64] per depth: the 4 LEV row indices (5 bits each) of the lane's window
  uint32_t* pref;      // [64*SEED_K]  first flattened entry of bucket (lane*K+slot) in this round
  uint32_t* pb;        // [64*SEED_K]  absolute arena offset of the bucket
  uint32_t* pth;       // [64*SEED_K]  the bucket's trie path (2 bits per char) | number of path chars << 24
  uint32_t* pat;       // [64]         the lane's automaton chars (pattern P)
  uint8_t* own;        // [SEED_OWN_CAP] flattened entry -> bucket (lane*K+slot)
};

// state of the 4 elements of a node given the state `piv` it was entered with: pending mask | states << 4
__device__ __forceinline__ uint32_t node_states(const uint4 nd, uint32_t rtw, uint32_t piv, const unsigned long long* s_row) {
  const uint32_t l0 = lev_next(s_row[rtw & 31u], piv), l1 = lev_next(s_row[(rtw >> 5) & 31u], piv);
  const uint32_t l2 = lev_next(s_row[(rtw >> 10) & 31u], piv), l3 = lev_next(s_row[(rtw >> 15) & 31u], piv);
  uint32_t m = 0;
  if ((nd.x >> ELEM_FLAG_SHIFT) && l0 != 14) m |= 1u;
  if ((nd.y >> ELEM_FLAG_SHIFT) && l1 != 14) m |= 2u;
  if ((nd.z >> ELEM_FLAG_SHIFT) && l2 != 14) m |= 4u;
  if ((nd.w >> ELEM_FLAG_SHIFT) && l3 != 14) m |= 8u;
  return m | (l0 << 4) | (l1 << 8) | (l2 << 12) | (l3 << 16);
}

// All searches of one wave; lane-varying mini-trie root `trie` (offsets inside are relative to it).  hl holds the
// lane-local hit lists (nh entries already present for DIR 1: the forward search's hits).
//
// Round = (1) every lane walks trie nodes in DFS order (pruned by the table automaton: all four element states of a
// node from a per-window table of LEV row indices) and collects its next <= SEED_K buckets; (2) the entries of all
// collected buckets are flattened lane-major (so one window's entries stay in DFS order) and evaluated one lane per
// entry, inputs fetched one chunk ahead; (3) accepted entries go back to the owning lane in entry order.  Work
// counters follow the reference's sequential scan: nothing after a 0-error match is counted.
#ifdef SMR_SEED_PHASES                                    // per-phase cycle accounting (debug build, SMR_DEBUG_PHASES=1)
#define SPH(i) { const unsigned long long tn_ = clock64(); sph[i] += tn_ - slast; slast = tn_; }
#else
#define SPH(i)
#endif
__device__ __forceinline__ void seed_search_wave(const uint32_t* __restrict__ arena, uint32_t root, bool mine, uint32_t chars, uint32_t pw, bool full,
                                                 const unsigned long long* s_row, const SeedLds L, uint32_t hcap, SeedLane& out
#ifdef SMR_SEED_PHASES
                                                 , unsigned long long* sph, unsigned long long& slast
#endif
                                                 ) {
  const int lane = lane_id();
  uint32_t nh = out.nh;
  bool zero = false, overflow = false;
  uint32_t n_node = 0, n_entry = 0;
  int sp = -1;
  uint32_t st = 0;                                      // pending mask | states of the node on top of the stack
  uint4 cur = make_uint4(0, 0, 0, 0);                   // its 4 elements
  uint32_t path = 0;                                    // chars of the DFS path, level l at bits 2l
  L.pat[lane] = chars;
  if (mine) {
    const BitVec bv = make_bitvec(chars, pw);
    for (uint32_t d = 0; d <= pw; d++) {
      uint32_t w = 0;
      for (uint32_t nt = 0; nt < 4; nt++) w |= lev_row_index(bv, d, nt, pw) << (5 * nt);
      L.rt[d * 64 + lane] = w;
    }
    sp = 0; L.stk[lane] = 0; cur = *reinterpret_cast<const uint4*>(arena + root); n_node = 1;
    st = node_states(cur, L.rt[lane], 0, s_row);
  }
  SPH(0)
  for (;;) {
    // ---------- (1) node walk: collect this lane's next buckets ----------
    uint32_t nb = 0, my_total = 0;
    uint32_t nent_pk = 0, nnode_pk = 0;                    // per collected bucket: entries / nodes visited so far in this round (8 bits each)
    const uint32_t n_node0 = n_node;
    while (sp >= 0 && nb < SEED_K && my_total < SEED_GATHER) {
      // one DFS move per iteration; the expensive part -- fetch a node and compute its 4 element states -- is shared by
      // "descend into a child" and "return to a parent that still has pending elements"
      uint32_t ld_off = NONE, ld_piv = 0, ld_mask = 15u;
      if ((st & 15u) == 0) {                               // node exhausted: back to the nearest ancestor with pending elements
        sp--;
        if (sp >= 0) {
          const uint32_t sw = L.stk[sp * 64 + lane];
          ld_mask = (sw >> 22) & 15u;
          if (ld_mask != 0) { ld_off = sw & ELEM_OFF_MASK; ld_piv = sw >> 26; }
        }
      } else {
        const uint32_t ne = (uint32_t)__ffs((int)(st & 15u)) - 1;
        st &= ~(1u << ne);
        const uint32_t e = ne == 0 ? cur.x : (ne == 1 ? cur.y : (ne == 2 ? cur.z : cur.w));
        const uint32_t lev_t = (st >> (4 + 4 * ne)) & 15u;
        if ((e >> ELEM_FLAG_SHIFT) == 1) {                  // child node
          L.stk[sp * 64 + lane] = (L.stk[sp * 64 + lane] & ~(15u << 22)) | ((st & 15u) << 22);
          path = (path & ((1u << (2 * sp)) - 1u)) | (ne << (2 * sp));
          sp++;
          ld_off = e & ELEM_OFF_MASK; ld_piv = lev_t;
          L.stk[sp * 64 + lane] = ld_off | (lev_t << 26);
          n_node++;
        } else {                                            // bucket
          const uint32_t nent = (e >> ELEM_NENT_SHIFT) & 0xFFu;
          L.pb[lane * SEED_K + nb] = root + (e & ELEM_OFF_MASK);
          L.pth[lane * SEED_K + nb] = (path & ((1u << (2 * sp)) - 1u)) | (ne << (2 * sp)) | ((uint32_t)(sp + 1) << 24);
          L.pref[lane * SEED_K + nb] = my_total;             // lane-relative for now
          nent_pk |= nent << (8 * nb); nnode_pk |= min(n_node - n_node0, 255u) << (8 * nb);
          my_total += nent; nb++;
        }
      }
      if (ld_off != NONE) {
        cur = *reinterpret_cast<const uint4*>(arena + root + ld_off);
        st = (node_states(cur, L.rt[sp * 64 + lane], ld_piv, s_row) & (~15u | ld_mask));
      } else if (sp >= 0 && (st & 15u) == 0 && ld_mask == 0) {
        st = 0;                                             // exhausted ancestor: keep climbing
      }
    }
    SPH(1)
    if (!__any(nb > 0)) break;
    // ---------- (2) flatten: lane-major prefix, entry -> bucket map ----------
    uint32_t incl = my_total;
    for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    const uint32_t T = __shfl(incl, 63, 64);
    const uint32_t my_first = incl - my_total;
    for (uint32_t k = 0; k < SEED_K; k++) L.pref[lane * SEED_K + k] = k < nb ? L.pref[lane * SEED_K + k] + my_first : my_first + my_total;
    const bool direct = T <= SEED_OWN_CAP;
    if (direct) {
      for (uint32_t k = 0; k < nb; k++) {
        const uint32_t f = L.pref[lane * SEED_K + k], c = (nent_pk >> (8 * k)) & 0xFFu;
        for (uint32_t q = 0; q < c; q++) L.own[f + q] = (uint8_t)(lane * SEED_K + k);
      }
    }
    __syncthreads();
    SPH(2)
    bool zero_round = false;                               // a 0-error match was found in this round (owner lane)
    // entry scan: one lane per entry; owner map, bucket descriptor and the entry itself are fetched one chunk ahead
    uint32_t f_bk = 0, f_q = 0, f_pbv = 0, f_str = 0, f_id = 0;
    auto fetch = [&](uint32_t base) {
      const uint32_t e = base + lane;
      const bool v = e < T;
      uint32_t bk = 0;
      if (v) {
        if (direct) bk = L.own[e];
        else for (uint32_t step = 128; step > 0; step >>= 1) { const uint32_t t = bk + step; if (t < 64 * SEED_K && L.pref[t] <= e) bk = t; }
      }
      f_bk = bk; f_q = e - L.pref[bk]; f_pbv = L.pth[bk];
      f_str = 0; f_id = 0;
      if (v) { const uint2 en = *reinterpret_cast<const uint2*>(arena + L.pb[bk] + 2 * f_q); f_str = en.x; f_id = en.y; }
    };
    if (T > 0) fetch(0);
    for (uint32_t base = 0; base < T; base += 64) {
      const bool v = base + lane < T;
      const uint32_t bk = f_bk, q = f_q, pbv = f_pbv, str = f_str, id = f_id;
      if (base + 64 < T) fetch(base + 64);
      const uint32_t olane = bk / SEED_K;
      const uint32_t nchar = pbv >> 24;                      // chars of the trie path in front of the tail
      const uint32_t tstr = (pbv & 0xFFFFFFu) | (str << (2 * nchar));
      const uint32_t r = v ? lev1_entry(L.pat[olane], tstr, pw) : 0u;
      const uint32_t kind = ((r & 2u) && !full) ? CK_COND : CK_PLAIN;   // a 0-error match is accepted one step before state 9 shows
      // hand the candidates back to their owners, in entry order
      unsigned long long cm = __ballot((r & 1u) != 0);
      while (cm) {
        const int c = __ffsll((long long)cm) - 1; cm &= cm - 1;
        const uint32_t o = __shfl(olane, c, 64), idc = __shfl(id, c, 64), kc = __shfl(kind, c, 64), qc = __shfl(q, c, 64), bc = __shfl(bk, c, 64);
        if ((uint32_t)lane == o && !zero) {
          bool present = false;
          for (uint32_t f = 0; f < nh; f++) if (L.hl[f * 64 + lane] == idc) { present = true; break; }
          if (kc == CK_UNCOND || (kc == CK_COND && !present)) {
            L.hl[lane] = idc; nh = 1; zero = true; zero_round = true;
            // the reference stops at the 0-error entry: count the buckets before it, this one up to the entry, no later node
            const uint32_t zs = bc % SEED_K;
            for (uint32_t k = 0; k < zs; k++) n_entry += (nent_pk >> (8 * k)) & 0xFFu;
            n_entry += qc + 1;
            n_node = n_node0 + ((nnode_pk >> (8 * zs)) & 0xFFu);
          } else if (!present) {
            if (nh < hcap) { L.hl[nh * 64 + lane] = idc; nh++; } else overflow = true;
          }
        }
      }
      SPH(3)
    }
    if (!zero_round) n_entry += my_total;
    if (zero) sp = -1;                                   // 0-error match: the reference unwinds the recursion (:167,256-262)
    __syncthreads();
  }
  out.nh = nh; out.zero = zero; out.overflow = overflow; out.n_node = n_node; out.n_entry = n_entry;
}

template <int DIR>
__global__ void __launch_bounds__(64) k_seed_search(DIndex ix, DParams P, int pass, SeedBufs sb, uint32_t hcap,
                                                    uint32_t* __restrict__ pool, uint32_t pool_words, unsigned long long* __restrict__ ctr,
                                                    const uint32_t* __restrict__ redo) {
  // this phase's tuples: forward bins first, reverse bins after them
  const uint32_t n_all = min(sb.sn[SN_TUPLES], sb.cap_tuples), n_fwd = min(sb.sn[SN_FWD], n_all);
  const uint32_t first = DIR ? n_fwd : 0u, n_tup = DIR ? n_all - n_fwd : n_fwd;
  uint32_t wave = blockIdx.x;
  if (redo) {                                            // only the waves listed by k_seed_pg (its candidate pool overflowed)
    if (blockIdx.x >= min(sb.sn[SN_REDO], sb.cap_redo)) return;
    wave = redo[blockIdx.x];
  }
  if (wave * 64u >= n_tup) return;
  SMR_DYN_LDS(uint32_t, lds_dyn);
  SeedLds L;
  L.hl = lds_dyn;
  L.stk = L.hl + 64 * hcap;
  L.rt = L.stk + SEED_STK * 64;
  L.pref = L.rt + (SEED_MAXPW + 1) * 64;
  L.pb = L.pref + 64 * SEED_K;
  L.pth = L.pb + 64 * SEED_K;
  L.pat = L.pth + 64 * SEED_K;
  L.own = reinterpret_cast<uint8_t*>(L.pat + 64);
  uint32_t* hl = L.hl;
  __shared__ unsigned long long s_row[LEV_ROWS];
  const int lane = lane_id();
  build_lev_rows(s_row);
  const uint32_t pos = first + wave * 64u + lane;
  bool mine = wave * 64u + lane < n_tup;
  uint32_t r = 0, win_pos = 0, chars = 0;
  uint32_t root = 0;
  size_t slot = 0;
  SeedLane sl; sl.nh = 0; sl.zero = false; sl.overflow = false; sl.n_node = 0; sl.n_entry = 0;
  uint32_t n_prev = 0;
  if (mine) {
    const SeedTmp tp = sb.srt[pos];
    const unsigned long long pl = seed_payload(tp);
    const Lookup lk = ix.lookup[tp.key - (DIR ? sb.nkh : 0u)];
    root = DIR == 0 ? lk.rootF : lk.rootR;
    r = (uint32_t)(pl & 0xFFFFFFull); win_pos = (uint32_t)((pl >> 24) & 0xFFFFull); chars = (uint32_t)(pl >> 40);
    slot = wseg_slot(sb, r, win_pos / P.skip[pass]);
    if (DIR == 1) {                                      // the window's list so far = the forward search's hits
      const uint32_t seg = sb.wseg[slot];
      if (seg != NONE && (seg & SEED_ZERO_BIT)) mine = false;     // accept_zero_kmer: no reverse search (paralleltraversal.cpp:188)
      else if (seg != NONE) {
        n_prev = pool[seg + 1];
        for (uint32_t q = 0; q < n_prev && q < hcap; q++) hl[q * 64 + lane] = pool[seg + 2 + 2 * q];
        if (n_prev > hcap) { sl.overflow = true; n_prev = hcap; }
        sl.nh = n_prev;
      }
    }
  }
  __syncthreads();
#ifdef SMR_SEED_PHASES
  unsigned long long sph[6] = {0, 0, 0, 0, 0, 0}, slast = clock64();
  seed_search_wave(ix.trie, root, mine, chars, P.partialwin, P.is_full_search != 0, s_row, L, hcap, sl, sph, slast);
#else
  seed_search_wave(ix.trie, root, mine, chars, P.partialwin, P.is_full_search != 0, s_row, L, hcap, sl);
#endif
  // ---- write the windows' hit segments: [unused, count, (id, win_pos) x count] ----
  const bool wr = mine && (DIR == 0 ? sl.nh > 0 : (sl.zero || sl.nh > n_prev));
  const uint32_t need = wr ? 2 + 2 * sl.nh : 0;
  uint32_t incl = need;
  for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
  const uint32_t total = __shfl(incl, 63, 64);
  uint32_t base = 0;
  if (total) {
    if (lane == 0) {                                     // the pool is split into C_NSHARD regions, each with its own cursor
      const uint32_t shard = blockIdx.x & (C_NSHARD - 1), region = pool_words / C_NSHARD;
      const unsigned long long old = atomicAdd(&ctr[C_PCUR + shard], (unsigned long long)total);
      if (old + total > region) { atomicAdd(&ctr[C_ERR_POOL], 1ull); base = NONE; } else base = shard * region + (uint32_t)old;
    }
    base = __shfl(base, 0, 64);
  }
  if (wr && base != NONE) {
    const uint32_t o = base + incl - need;
    pool[o] = NONE; pool[o + 1] = sl.nh;
    for (uint32_t q = 0; q < sl.nh; q++) { pool[o + 2 + 2 * q] = hl[q * 64 + lane]; pool[o + 3 + 2 * q] = win_pos; }
    sb.wseg[slot] = o | (sl.zero ? SEED_ZERO_BIT : 0u);
  }
  if (__any(sl.overflow) && lane == 0) atomicAdd(&ctr[C_ERR_HITCAP], 1ull);
  // algorithmic bytes of this wave (C_B_PG0/1): tuple + lookup entry per search, 16 B per node, 8 B per entry, the forward list read (DIR 1),
  // the segment written + its window slot
  unsigned long long v[3] = {sl.n_node, sl.n_entry, 0};
  v[2] = (wave * 64u + lane < n_tup ? sizeof(SeedTmp) + sizeof(Lookup) + (DIR ? 4u + (n_prev ? 4u + 8u * n_prev : 0u) : 0u) : 0u) + 16ull * sl.n_node + 8ull * sl.n_entry + 4ull * need + (wr ? 4u : 0u);
  for (int c = 0; c < 3; c++) {
    unsigned long long x = v[c];
    for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
    if (lane == 0 && x) ctr_add(ctr, c < 2 ? C_NODE + c : (DIR ? C_B_PG1 : C_B_PG0), x);
  }
#ifdef SMR_SEED_PHASES
  SPH(5)
  if (lane == 0) for (int q = 0; q < 6; q++) atomicAdd(&ctr[C_SHARDS + (blockIdx.x & (C_NSHARD - 1)) * C_SHARD_W + C_SHARD_PH + q], sph[q]);
#endif
}

// per read: copy the hit segments of this pass's windows into ONE contiguous block (k_chain then reads a strand's
// cumulative hits with coalesced loads instead of chasing a list), count seeds/hits (++read.hit_seeds per window with
// hits, paralleltraversal.cpp:242-249), make the 0..3 view persistent (Read::flip34, read.cpp:379-401)
__global__ void __launch_bounds__(256) k_seed_finish(DReads rd, DParams P, int pass, SeedBufs sb, RState* __restrict__ work,
                                                     RWork* __restrict__ rw, uint32_t* __restrict__ pool, uint32_t pool_words,
                                                     unsigned long long* __restrict__ ctr) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long hits = 0, bytes = 0, looks = 0, moved = 0;      // moved: algorithmic bytes of this read (C_B_FIN)
  if (r < rd.n) {
    RWork w = rw[r];
    moved = sizeof(RWork);
    if (w.strand_active && w.search && w.pass_n == (uint32_t)pass) {
      const uint32_t len = rd.len[r], stride = P.skip[pass];
      const uint32_t numwin = (len - P.lnwin + stride) / stride;
      uint32_t seeds = 0, total = 0, rlook = 0;
      for (uint32_t k = 0; k < numwin; k++) {
        const uint32_t s = sb.wseg[wseg_slot(sb, r, k)];
        bool searched = true;                                // windows of this pass: not searched by an earlier pass (:128-131)
        for (int q = 0; q < pass; q++) if ((k * stride) % P.skip[q] == 0) searched = false;
        if (searched && !(s != NONE && (s & SEED_ZERO_BIT))) rlook++;     // the reverse lookup happens unless the forward search hit exactly (:188-198)
        if (s == NONE) continue;
        seeds++; total += pool[(s & ~SEED_ZERO_BIT) + 1];
      }
      looks = rlook;
      // length, one window slot per window, per segment its count word, every (id, win_pos) pair read and written, per-read state written, hit_seeds
      moved += 4u + 4ull * numwin + 4ull * seeds + 16ull * total + sizeof(RWork) + 8u;
      uint32_t base = 0;
      if (total) {
        const uint32_t shard = blockIdx.x & (C_NSHARD - 1), region = pool_words / C_NSHARD;
        const unsigned long long old = atomicAdd(&ctr[C_PCUR + shard], 2ull * total);
        if (old + 2ull * total > region) { atomicAdd(&ctr[C_ERR_POOL], 1ull); total = 0; seeds = 0; }
        else base = shard * region + (uint32_t)old;
      }
      uint32_t o = base;
      if (total) for (uint32_t k = 0; k < numwin; k++) {
        const uint32_t s = sb.wseg[wseg_slot(sb, r, k)];
        if (s == NONE) continue;
        const uint32_t sg = s & ~SEED_ZERO_BIT, c = pool[sg + 1];
        for (uint32_t q = 0; q < 2 * c; q++) pool[o + q] = pool[sg + 2 + q];
        o += 2 * c;
      }
      w.aval = w.is04 ? 0 : w.aval; w.is04 = 0;
      w.blk_off[pass] = base; w.blk_cnt[pass] = total; w.hit_total += total;
      rw[r] = w;
      work[r].hit_seeds += seeds;
      hits = total; bytes = (len + 3) / 4;
    }
  }
  for (int d = 32; d > 0; d >>= 1) { hits += __shfl_down(hits, d, 64); bytes += __shfl_down(bytes, d, 64); looks += __shfl_down(looks, d, 64); moved += __shfl_down(moved, d, 64); }
  if (lane_id() == 0) { if (hits) ctr_add(ctr, C_HIT, hits); if (bytes) ctr_add(ctr, C_READ_BYTES, bytes); if (looks) ctr_add(ctr, C_LOOKUP, looks); if (moved) ctr_add(ctr, C_B_FIN, moved); }
}

}  // namespace smr
