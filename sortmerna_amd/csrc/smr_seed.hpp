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
// The three run conditions without counting runs: with a = the common prefix, "a + s0 >= pw-1" says that nothing differs behind char a
// (at most ONE differing char), "a + s1 >= pw" that P and T shifted by one char agree from char a on, "a + s2 >= pw-1" the same for P
// shifted -- three shifts of the xor words by 2a bits (raw xor bits do: a nonzero bit at or behind bit 2a is a differing char).  17 VALU
// instructions instead of 45 (three count-leading-zeros, minima, sums); checked equal to the run form on every (P, T) for pw = 4..6 and on
// 1.6e8 random near-matches for pw = 7..10 (round 4), and through it to the tables (tests/test_lev_closed_form.py).
// returns bit 0 = accepted, bit 1 = 0-error match
__device__ __forceinline__ uint32_t lev1_entry(uint32_t P, uint32_t T, uint32_t pw) {
  const uint32_t m2 = (1u << (2 * pw)) - 1u, m2b = m2 >> 2;
  const uint32_t x0 = (P ^ T) & m2, x1 = (P ^ (T >> 2)) & m2, x2 = ((P >> 2) ^ T) & m2b;
  const uint32_t a2 = (uint32_t)__builtin_ctz(x0 | (1u << (2 * pw))) & ~1u;          // 2 * common prefix length
  const bool acc = (((x0 >> a2) >> 2) == 0) || ((x1 >> a2) == 0) || ((x2 >> a2) == 0);
  return (acc ? 1u : 0u) | (x0 == 0 ? 2u : 0u);
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
enum { SN_TUPLES = 0, SN_REDO = 1, SN_FWD = 2, SN_PIECES = 3, SN_HOTBINS = 4, SN_COUNT = 8 };      // device counters of the seed stage (u32): tuples, redo waves, forward tuples, pieces of hot keys, large coarse bins

// The tuples of the seed stage are 8 bytes (round 3: 12).  A tuple = one half-seed search of one window: its 9-mer key (direction in the top
// bit: forward keys [0, nkh), reverse keys [nkh, 2 nkh)), the pw chars that feed the automaton, and the window's SLOT = read * maxwin +
// window index (< 2^31: what wseg / fbits are indexed with -- nobody downstream needs read and window apart).  Key + chars + slot are
// 2 L + 1 + 32 bits, too many; but a block of k_seed_keys owns a contiguous run of reads, so in ITS region of tmp the slot is stored relative
// to the block's first slot, and from the first sort pass on the coarse key bits are implied by where the tuple lies:
//   tmp        key (kbits = L + 1) | chars (cb = L) << kbits | block-relative slot << (kbits + cb)
//   mid, srt   slot (32) | chars << 32 | fine key bits (fb) << (32 + cb)            coarse bin of srt[i]: wbin[i / 64], then cbase
//   srt, a tuple that k_seed_dedup found to repeat another one of its key:   slot (32) | slot of that REPRESENTATIVE << 32 | 1 << 63
//   (slots are < 2^31, chars | fine bits < 2^29: bit 63 tells the two apart).  Such a tuple is not searched: its window gets the
//   representative's hit segment (k_seed_prop).
typedef unsigned long long SeedTup;
#define SEED_TUP_DUP (1ull << 63)
struct SeedKey { uint32_t slot, chars, key; bool dup; };          // a decoded tuple of srt

#define SEED_KEY_BLOCKS 2048u                             // most blocks of k_seed_keys / k_seed_split (rows of the histogram matrix)
#define SEED_WAVES 16u                                    // waves per block of k_seed_keys
#define SEED_STAGE_WORDS 1280u                            // LDS words in which a wave of k_seed_keys stages the packed records of the reads of one trip (64 reads of <= 208 letters)
#ifndef SEED_PIECE
#define SEED_PIECE 16384u                                 // tuples the second sort pass (512 fine bins) stages in LDS at a time: 128 KB, one 1024-thread block per CU, sixteen loads per thread in flight
#endif
#ifndef SEED_SPLIT_PIECE
#define SEED_SPLIT_PIECE 16384u                           // ... and the first (1 024 coarse bins: runs of 16 tuples per bin; 140 KB with its tables).  Measured per 8 M-read step, builds side by side on one box:
#endif                                                    // round 6 before the loads of a piece left together 8 192 / 12 288 were best (profiles/r6s17_*); since then 12 288 -> 16 384 takes 0.5 ms off the first pass and
                                                          // 8 192 -> 16 384 0.9 ms off the second (profiles/r6s53_* - r6s55_*: 19.1 -> 17.6 ms for both), 18 432 no more and the second pass spills
#define SEED_SPLIT_PIECE_MANY_BINS 12288u                 // ... with more than 2 048 coarse bins (seed length 20: 4 096 bins = 48 KB of tables): 144 KB
#define SEED_SEG_MERGED 0x80000000u                       // header bit of a reverse segment whose list is final (written by k_seed_search<1>: forward hits included)
#define SEED_CAND_COND 0x80000000u                        // bit of an id in a reverse segment of k_seed_pg: this candidate is a 0-error match
// A window whose search leaves ONE hit (most windows of a read sampled from the DB: the 0-error match) needs no segment: its wseg word IS the hit --
// SEED_SEG_INLINE | id (| SEED_ZERO_BIT forward, | SEED_CAND_COND reverse).  k_seed_pg then writes no pool words for it and k_seed_finish reads none: a
// 128-byte line per window less in the kernel furthest from its bytes (round 5: 7.8 GB moved for 1.3 GB).  Needs ids and pool offsets below 2^30
// (SeedBufs::seg_inline; the host checks both).
#define SEED_SEG_INLINE 0x40000000u
#define SEED_SEG_ID 0x3FFFFFFFu
struct SeedBufs {
  uint32_t* chist;           // [nc + 1] tuples per COARSE bin (key >> fb)
  uint32_t* cbase;           // [nc + 1] exclusive scan of chist
  uint32_t* rows;            // [kb][nc] tuples of block b of k_seed_keys per coarse bin; after k_seed_colscan: where its first tuple of that bin goes in mid, counted from the bin's base
  uint32_t* bcnt;            // [kb] tuples block b wrote (at tmp[2 * rpb * maxwin * b ...), compact)
  SeedTup* tmp;              // unsorted tuples, one compact region per block of k_seed_keys
  SeedTup* mid;              // tuples grouped by coarse bin
  SeedTup* srt;              // tuples in key order
  uint32_t* wseg[2];         // [slot] pool offset of the window's forward / reverse hit segment | SEED_ZERO_BIT; valid where the slot's bit in fbits[d] is set
  uint32_t* fbits[2];        // bit per slot (read-major: a read's windows are consecutive bits): the window has a forward / reverse segment
  uint32_t* zbits;           // bit per slot: the window's forward search ended with a 0-error match (no reverse search, paralleltraversal.cpp:188)
  uint32_t* gflag;           // bit per 64 slots: one of them has its zbit set (what a reverse search asks first: under a megabyte)
  uint16_t* wbin;            // [ceil(tuples / 64)] coarse bin of srt[64 w]
  uint32_t* emap;            // 2 bits per 9-mer: its forward / reverse tuple is emitted (lookup_tbl[kmer].count > minoccur and the mini-trie exists)
  uint32_t* sn;              // SN_* counters
  uint32_t* redo;            // waves of k_seed_pg to be searched again by k_seed_search
  uint32_t nk, nkh, maxwin, cap_tuples, cap_redo;     // nk = 2 * nkh bins: forward keys [0, nkh), reverse keys [nkh, 2 nkh)
  uint32_t fb, nc;           // fine bits (min(9, L): a coarse bin never mixes forward and reverse keys), nc = nk >> fb coarse bins (<= 4096)
  uint32_t n;                // reads in the batch
  uint32_t kb, rpb;          // blocks of k_seed_keys / k_seed_split, reads per block
  uint32_t cb, kbits;        // bits of the automaton chars (2 pw = L) and of the key (L + 1)
  uint32_t g_shift;          // k_seed_keys: log2 of the lanes per read (0: a lane walks all windows of its read; 6: one read per wave)
  // skewed batches (amplicons, a sample dominated by one organism's rRNA: thousands of windows share a key, and most of them the whole seed)
  uint32_t* hpre;            // [nc + 1] coarse bins far larger than the average are sorted by several blocks: exclusive prefix of their numbers of sub-ranges (0 for the others)
  uint32_t* hlist;           // [nc] the large coarse bins (sn[SN_HOTBINS] of them, in no particular order)
  uint32_t* hh;              // [cap_hent][2^fb] per sub-range of a large coarse bin: its histogram of the fine bits, then where its first tuple of every fine bin goes
  uint2* pieces;             // {first tuple, tuples <= SEED_DD_PIECE} of the keys with at least hot_min tuples (and four times the average): where k_seed_dedup looks for repeated seeds
  uint32_t cap_hent, cap_pieces;
  // one sort for several index parts (launch_seed in smr_engine.hip): the searches of a part walk sorted arrays that hold the tuples of EVERY read and skip the
  // tuples of the reads that are not in this (part, strand, pass)
  const uint32_t* abits;     // bit per read: it is searched in this launch (nullptr: every tuple is)
  double inv_maxwin;         // 1 / maxwin: a tuple's read = slot / maxwin
  uint32_t seg_inline;       // 1: a one-hit window's hit lies in its wseg word (SEED_SEG_INLINE)
  uint32_t hot_min;          // 0: no search for repeated seeds
  uint32_t hbin_min, hsub;   // a coarse bin is "large" from twice the average size and at least hbin_min tuples (SEED_HOT_BIN_MIN); tuples per sub-range (SEED_HOT_SUB)
};
__device__ __forceinline__ bool wseg_has(const SeedBufs& sb, int d, uint32_t slot) { return (sb.fbits[d][slot >> 5] >> (slot & 31u)) & 1u; }
__device__ __forceinline__ void wseg_put(const SeedBufs& sb, int d, uint32_t slot, uint32_t v, bool zero) {
  sb.wseg[d][slot] = v; atomicOr(&sb.fbits[d][slot >> 5], 1u << (slot & 31u));
  if (d == 0 && zero) { atomicOr(&sb.zbits[slot >> 5], 1u << (slot & 31u)); atomicOr(&sb.gflag[slot >> 11], 1u << ((slot >> 6) & 31u)); }
}
// is the read of window slot `slot` searched in this launch?  (slot / maxwin through a double: exact for slots < 2^31 and maxwin < 2^16)
__device__ __forceinline__ bool seed_read_active(const SeedBufs& sb, uint32_t slot) {
  if (!sb.abits) return true;
  const uint32_t r = (uint32_t)(((double)slot + 0.5) * sb.inv_maxwin);
  return (sb.abits[r >> 5] >> (r & 31u)) & 1u;
}
// tuple i of srt: its fields, the key completed from the position (wave chunk -> first coarse bin, then the bin boundaries)
__device__ __forceinline__ SeedKey seed_decode(const SeedBufs& sb, uint32_t i) {
  const SeedTup t = sb.srt[i];
  uint32_t c = sb.wbin[i >> 6];
  uint32_t nx = sb.cbase[c + 1];
  while (i >= nx) { c++; nx = sb.cbase[c + 1]; }           // (i < cbase[nc]: ends; a chunk of 64 tuples rarely spans more than two bins)
  SeedKey k;
  k.slot = (uint32_t)t; k.chars = (uint32_t)(t >> 32) & ((1u << sb.cb) - 1u); k.key = (c << sb.fb) | ((uint32_t)(t >> (32u + sb.cb)) & ((1u << sb.fb) - 1u));
  k.dup = (t & SEED_TUP_DUP) != 0;                          // (chars and key of such a tuple mean nothing)
  return k;
}

// nbits <= 40 bits starting at bit `bit0` of a little-endian word stream (reads up to 2 words past the first)
template <class PTR>
__device__ __forceinline__ unsigned long long extract_bits(PTR w, uint32_t bit0, uint32_t nbits) {
  const uint32_t i = bit0 >> 5, s = bit0 & 31u;
  unsigned long long v = ((unsigned long long)w[i] | ((unsigned long long)w[i + 1] << 32)) >> s;
  if (s) v |= (unsigned long long)w[i + 2] << (64 - s);
  return v & ((1ull << nbits) - 1ull);
}
// the 2-bit groups of the low 2n bits of x in reverse order
__device__ __forceinline__ uint32_t rev_groups(uint32_t x, uint32_t n) {
  uint32_t r = __brev(x) >> (32 - 2 * n);
  return ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
}
// bit i of m (i < 32) to bit 2 i
__device__ __forceinline__ unsigned long long spread_bits(uint32_t m) {
  unsigned long long v = m;
  v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
  v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
  v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
  v = (v | (v << 2)) & 0x3333333333333333ull;
  v = (v | (v << 1)) & 0x5555555555555555ull;
  return v;
}

// window content of a read in the CURRENT strand/encoding state: 2 bits per nt, char i at bits 2i -- one 2L-bit extraction from the packed
// record (reverse strand: 2-bit groups reversed and complemented, Read::revIntStr read.cpp:350-357); ambiguous letters (packed as code 0 +
// mask bit) read as `aval` (Read::flip34, read.cpp:379-401), also one extraction.  Same values as L calls of read_nt.
template <class PTR>
__device__ __forceinline__ unsigned long long window_chars(PTR rec, uint32_t len, uint32_t win_pos, uint32_t L, uint32_t reversed, uint32_t aval) {
  const uint32_t cw = (len + 15) >> 4;
  const uint32_t j0 = reversed ? (len - win_pos - L) : win_pos;           // first read position covered, ascending
  uint32_t m = (uint32_t)extract_bits(rec + cw, j0, L);
  unsigned long long x = extract_bits(rec, 2 * j0, 2 * L);
  if (reversed) {
    x = __brevll(x) >> (64 - 2 * L);                                      // reverse the bit order ...
    x = ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);   // ... and restore each 2-bit group
    x = ~x & ((1ull << (2 * L)) - 1ull);                                   // complement: 3 - code
    m = __brev(m) >> (32 - L);
  }
  if (m) { const unsigned long long sp = spread_bits(m); x = (x & ~(3ull * sp)) | (sp * aval); }
  return x;
}

// 2 bits per 9-mer key for k_seed_keys: is the forward / the reverse tuple of a window with this key emitted?  (lookup_tbl[kmer].count > minoccur
// and the mini-trie exists: paralleltraversal.cpp:155-160, 186-192.)  64 KB for L = 18: every block of k_seed_keys keeps a copy in LDS.
__global__ void k_seed_emap(const uint32_t* __restrict__ lkc, uint32_t nkh, uint32_t minoccur, uint32_t* __restrict__ emap) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (16u * i >= nkh) return;
  uint32_t w = 0;
  for (uint32_t q = 0; q < 16; q++) {
    const uint32_t key = 16u * i + q;
    if (key >= nkh) break;
    const uint32_t v = lkc[key];
    const bool ok = (v & 0x3FFFFFFFu) > minoccur;
    w |= ((ok && ((v >> 30) & 1u) ? 1u : 0u) | (ok && (v >> 31) ? 2u : 0u)) << (2 * q);
  }
  emap[i] = w;
}

// Both half-seed searches of a window become tuples in ONE pass (so the sort runs once per stage):
//   forward: key = first 9-mer, automaton fed by chars [pw, 2pw)          (init_win_f bitvector.cpp:57-91)  -> bins [0, nkh)
//   reverse: key = second 9-mer, automaton fed by chars pw-1 .. 0         (init_win_r :99-132)              -> bins [nkh, 2 nkh)
// READ-centric (round 3: one thread per window slot, each running the chain state -> length / record offset -> record words -> two lookups):
// a block owns `rpb` consecutive reads; per trip a wave takes 64 >> g_shift of them, asks for their states, lengths and record offsets
// together, copies their packed records -- contiguous in the batch -- into LDS with 16-byte loads (STAGED; records too long for that are read
// where they lie), and then 2^g_shift lanes per read (ONE: a lane per read) walk the read's windows: placement (paralleltraversal.cpp:118-131),
// window extraction from LDS, both 9-mer keys, both emit bits from the block's LDS copy of emap (MAPPED) -- no global load depends on another
// after the first three.  The tuples of a block go compactly into ITS region of tmp (a wave reserves its slots with one LDS atomic per window
// round), counted per COARSE bin (key >> fb) in LDS; that histogram is the block's row of sb.rows: the first sort pass needs no counting pass.
// `mode` (one sort for several index parts): SEED_KEYS_ALL = the tuples of every read of the part's (strand, pass), as described; SEED_KEYS_SHARED (| strand << 4) =
// the tuples of EVERY read long enough to have a window (reverse strand: and free of ambiguous letters), for strand and pass as given, whatever state the reads are in and whether or not
// a part's lookup table has the key (the searches of a part skip what is not theirs: seed_read_active, a missing mini-trie); SEED_KEYS_AMB = like ALL, but only
// the reads WITH ambiguous letters -- on the reverse strand what such a letter reads as depends on the read's history in the part (Read::flip34), so their tuples are made per part.
enum { SEED_KEYS_ALL = 0, SEED_KEYS_SHARED = 1, SEED_KEYS_AMB = 2 };
template <bool ONE, bool STAGED, bool MAPPED>
__global__ void __launch_bounds__(64 * SEED_WAVES) k_seed_keys(DReads rd, DParams P, int pass, SeedBufs sb, const RWork* __restrict__ rw, unsigned long long* __restrict__ ctr, int mode) {
  SMR_DYN_LDS(uint32_t, lds);                             // lh [nc rounded to 4] | emap copy [nkh / 16] (MAPPED) | per wave SEED_STAGE_WORDS + 8 (STAGED)
  __shared__ uint32_t s_cur;
  uint32_t* const lh = lds;
  const uint32_t lh_words = (sb.nc + 3u) & ~3u, map_words = MAPPED ? sb.nkh >> 4 : 0u;
  uint32_t* const map = lds + lh_words;
  const uint32_t wv = threadIdx.x >> 6;
  uint32_t* const stage = lds + lh_words + map_words + wv * (SEED_STAGE_WORDS + 8u);
  for (uint32_t c = threadIdx.x; c < sb.nc; c += blockDim.x) lh[c] = 0;
  if (MAPPED) for (uint32_t c = threadIdx.x; c < map_words; c += blockDim.x) map[c] = sb.emap[c];
  if (threadIdx.x == 0) s_cur = 0;
  __syncthreads();
  const int lane = lane_id();
  const uint32_t pw = P.partialwin, L = P.lnwin;
  const uint32_t s0 = P.skip[0], s1 = P.skip[1], stride = pass == 0 ? s0 : pass == 1 ? s1 : P.skip[2];
  const uint32_t rb0 = blockIdx.x * sb.rpb, rb1 = min(rb0 + sb.rpb, rd.n);
  SeedTup* const region = sb.tmp + (size_t)2 * rb0 * sb.maxwin;
  const uint32_t gsh = ONE ? 0u : sb.g_shift, G = 1u << gsh, RW = 64u >> gsh;
  const uint32_t sub = ONE ? 0u : (uint32_t)lane & (G - 1u), rl = (uint32_t)lane >> gsh;
  const uint32_t half = (1u << (2 * pw)) - 1u;
  uint32_t nwin = 0;                                      // windows of this wave (uniform)
  for (uint32_t rt0 = rb0 + wv * RW; rt0 < rb1; rt0 += SEED_WAVES * RW) {
    const uint32_t r = rt0 + rl;
    const bool have = r < rb1;
    RWork w; w.strand_active = 0; w.search = 0; w.pass_n = 0; w.is04 = 0; w.aval = 0; w.reversed = 0;
    uint32_t len = 0; uint64_t off = 0;
    if (have) { w = rw[r]; len = rd.len[r]; off = rd.rec_off[r]; }       // (asked for together: one round trip)
    bool active = have && w.strand_active && w.search && w.pass_n == (uint32_t)pass;
    // (forward strand: an ambiguous letter always reads 0 -- after Smith-Waterman is04 is set, and k_seed_finish puts 0 back --, so those reads share too)
    if ((mode & 15) == SEED_KEYS_SHARED) { active = have && len >= L && (!w.has_amb || (mode >> 4) == 0); w.reversed = (uint8_t)(mode >> 4); w.is04 = 0; w.aval = 0; }
    else if ((mode & 15) == SEED_KEYS_AMB) active = active && w.has_amb;
    if (!__any(active)) continue;
    const uint32_t numwin = active ? (len - L + stride) / stride : 0u;     // paralleltraversal.cpp:118-120
    const uint32_t* const grec = rd.words + off;
    uint32_t* srec = stage;
    if (STAGED) {
      const uint64_t wb = rd.rec_off[rt0], we = rd.rec_off[min(rt0 + RW, rb1)];   // the records of the trip's reads: one contiguous run of words
      const uint64_t a0 = wb & ~3ull;
      const uint32_t sh = (uint32_t)(wb - a0), nq = (sh + (uint32_t)(we - wb) + 3u) >> 2;
      const uint4* const src = reinterpret_cast<const uint4*>(rd.words + a0);
      uint4* const dst = reinterpret_cast<uint4*>(stage);
      __builtin_amdgcn_wave_barrier();                    // (the previous trip's extractions are done: LDS operations of a wave complete in order)
      for (uint32_t q = (uint32_t)lane; q < nq; q += 64) dst[q] = src[q];
      __builtin_amdgcn_wave_barrier();
      srec = stage + sh + (uint32_t)(off - wb);
    }
    // traverse(): `if (read.is04) read.flip34()` before every window (:126) -> ambiguous positions read as 0 / 3
    const uint32_t aval = w.is04 ? 0u : (uint32_t)w.aval;
    const uint32_t rel0 = (r - rb0) * sb.maxwin;
    for (uint32_t kk = 0; kk < sb.maxwin; kk += G) {
      const uint32_t k = kk + sub, win_pos = k * stride;
      bool mine = k < numwin;
      if (pass >= 1 && win_pos % s0 == 0) mine = false;                     // read_pos_searched (:128-131)
      if (pass >= 2 && win_pos % s1 == 0) mine = false;
      if (!__any(mine)) continue;
      bool e0 = false, e1 = false;
      uint32_t ra = 0, rb = 0, a = 0, b = 0;
      if (mine) {
        const unsigned long long wc = STAGED ? window_chars(srec, len, win_pos, L, w.reversed, aval) : window_chars(grec, len, win_pos, L, w.reversed, aval);
        // first / second 9-mer with char i at bits 2i; hashKmer is MSB-first (read.cpp:601-611) = the 2-bit groups reversed
        a = (uint32_t)wc & half; b = (uint32_t)(wc >> (2 * pw)) & half;
        ra = rev_groups(a, pw); rb = rev_groups(b, pw);
        const uint32_t wa = MAPPED ? map[ra >> 4] : sb.emap[ra >> 4], wb_ = MAPPED ? map[rb >> 4] : sb.emap[rb >> 4];
        e0 = (wa >> (2u * (ra & 15u))) & 1u;
        e1 = (wb_ >> (2u * (rb & 15u) + 1u)) & 1u;
        if ((mode & 15) == SEED_KEYS_SHARED) e0 = e1 = true;
      }
      const unsigned long long em0 = __ballot(e0), em1 = __ballot(e1);
      nwin += (uint32_t)__popcll(__ballot(mine));
      const uint32_t c0 = (uint32_t)__popcll(em0), c1 = (uint32_t)__popcll(em1);
      uint32_t base = 0;
      if (lane == 0 && c0 + c1) base = atomicAdd(&s_cur, c0 + c1);         // the wave's slots in the block's region: its forward tuples, then its reverse tuples
      base = (uint32_t)__shfl((int)base, 0, 64);
      const unsigned long long below = (1ull << lane) - 1ull;
      const unsigned long long relk = (unsigned long long)(rel0 + k) << (sb.kbits + sb.cb);
      if (e0) {                                            // forward: second half in order
        region[base + (uint32_t)__popcll(em0 & below)] = (unsigned long long)ra | ((unsigned long long)b << sb.kbits) | relk;
        atomicAdd(&lh[ra >> sb.fb], 1u);
      }
      if (e1) {                                            // reverse: first half walked backwards
        const uint32_t key = sb.nkh + rb;
        region[base + c0 + (uint32_t)__popcll(em1 & below)] = (unsigned long long)key | ((unsigned long long)ra << sb.kbits) | relk;
        atomicAdd(&lh[key >> sb.fb], 1u);
      }
    }
  }
  (void)nwin;                                              // (the windows are counted by k_seed_finish: a shared sort is not a part's work)
  __syncthreads();
  uint32_t* const row = sb.rows + (size_t)blockIdx.x * sb.nc;
  for (uint32_t c = threadIdx.x; c < sb.nc; c += blockDim.x) row[c] = lh[c];       // (the bins' totals are k_seed_colscan's column sums: no atomics here)
  if (threadIdx.x == 0) {
    sb.bcnt[blockIdx.x] = s_cur;
  }
  (void)ctr;
}

// The tuples are brought into key order by a two-level counting sort.  No pass writes a tuple with a store of its own lane's choosing
// (scattered stores run at 0.9 TB/s on the MI355X whatever the run length, coalesced ones at 5.6 TB/s; profiles/r03a_pmc_calibration_*):
// a pass stages SEED_PIECE tuples in LDS in bin order and copies them out with consecutive lanes on consecutive tuples.
//   k_seed_keys     leaves per block the number of its tuples per COARSE bin (key >> fb, <= 4096 bins)       -> rows
//   k_seed_colscan  rows[b][c] = the tuples of bin c in the rows above (where block b's tuples of bin c go inside the bin), the bin's total -> chist
//   k_seed_cscan    exclusive scan of the coarse counts (one block)                                            -> cbase, SN_TUPLES, SN_FWD
//   k_seed_wbin     coarse bin of every 64th tuple of the sorted array (the key of a sorted tuple = its bin | its fine bits)
//   k_seed_split    block b reads ITS tuples once, piece by piece, and moves them to their coarse bins (block-relative slot -> slot, key -> fine bits)
//   k_seed_bins     one block per coarse bin: histogram of the fine key bits (<= 512 bins), then the same staged move to the final places
// History per 60 M tuples on the MI355X: one counting sort with a returning global atomic per tuple 2.4 + 1.9 ms; two-level with LDS
// atomics and per-lane stores 1.3 + 0.9 + 1.0 ms (HBM writes 3.0 x and 2.7 x the tuple bytes, profiles/r03a_sort_variants_*).
// + which coarse bins are far larger than the average (a key that thousands of windows share): k_seed_bins leaves those to the k_seed_hbins_*
// kernels, which sort one bin with several blocks -- hpre[c] = the sub-ranges of SEED_HOT_SUB tuples of the large bins in front of c
#define SEED_HOT_SUB 65536u                               // tuples of a large coarse bin that one block of k_seed_hbins_* takes   (SeedBufs::hsub; SMR_SEED_HOT_SUB: the tests' small batches)
#define SEED_HOT_BIN_MIN 262144u                          // a coarse bin is "large" from twice the average size and at least this many tuples   (SeedBufs::hbin_min; SMR_SEED_HOT_BIN)
#define SEED_DD_PIECE 16384u                              // tuples of one key that a block of k_seed_dedup looks at together
#define SEED_DD_TAB 4096u                                 // ... slots of its hash table in LDS (8 bytes each)
__device__ __forceinline__ uint32_t seed_hot_bin(const SeedBufs& sb, uint32_t n_tup) { return max(sb.hbin_min, 2u * (n_tup / sb.nc)); }
__global__ void __launch_bounds__(1024) k_seed_cscan(SeedBufs sb, unsigned long long* __restrict__ ctr) {
  __shared__ uint32_t s_part[16], s_part2[16];
  // <= 4096 bins: 4 consecutive bins per thread
  const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  uint32_t v[4], sum = 0;
  for (int q = 0; q < 4; q++) { const uint32_t c = 4 * t + q; v[q] = c < sb.nc ? sb.chist[c] : 0u; sum += v[q]; }
  uint32_t incl = sum;
  for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d, 64); if ((int)lane >= d) incl += x; }
  if (lane == 63) s_part[wv] = incl;
  __syncthreads();
  uint32_t pre = incl - sum, n_all = 0;
  for (uint32_t q = 0; q < 16; q++) { if (q < wv) pre += s_part[q]; n_all += s_part[q]; }
  // the sub-ranges of the large bins
  const uint32_t big = seed_hot_bin(sb, n_all);
  uint32_t ns[4], nsum = 0;
  for (int q = 0; q < 4; q++) { ns[q] = v[q] >= big ? (v[q] + sb.hsub - 1u) / sb.hsub : 0u; nsum += ns[q]; }
  uint32_t nincl = nsum;
  for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(nincl, d, 64); if ((int)lane >= d) nincl += x; }
  if (lane == 63) s_part2[wv] = nincl;
  __syncthreads();
  uint32_t npre = nincl - nsum;
  for (uint32_t q = 0; q < wv; q++) npre += s_part2[q];
  for (int q = 0; q < 4; q++) {
    const uint32_t c = 4 * t + q;
    if (c < sb.nc) {
      sb.cbase[c] = pre; sb.hpre[c] = npre;
      if (ns[q]) sb.hlist[atomicAdd(&sb.sn[SN_HOTBINS], 1u)] = c;
      if (c == (sb.nkh >> sb.fb)) { sb.sn[SN_FWD] = pre; if (pre) ctr_add(ctr, C_TUP_F, pre); }   // the forward tuples lie in front of coarse bin nkh >> fb
    }
    pre += v[q]; npre += ns[q];
    if (c + 1 == sb.nc) { sb.cbase[sb.nc] = pre; sb.hpre[sb.nc] = npre; sb.sn[SN_TUPLES] = pre; if (pre) ctr_add(ctr, C_TUP_ALL, pre); }
  }
}

// one block per 64 coarse bins (lane = bin); wave w takes the rows [w * kb / 16, (w + 1) * kb / 16): their sum, then -- offset by the waves
// above -- the running place of every row inside its bin; the column's sum is the bin's size
__global__ void __launch_bounds__(1024) k_seed_colscan(SeedBufs sb) {
  __shared__ uint32_t s_sum[16][64];
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  const uint32_t c = blockIdx.x * 64u + lane;
  const uint32_t per = (sb.kb + 15u) / 16u, r0 = min(wv * per, sb.kb), r1 = min(r0 + per, sb.kb);
  uint32_t sum = 0;
  if (c < sb.nc) for (uint32_t r = r0; r < r1; r++) sum += sb.rows[(size_t)r * sb.nc + c];
  s_sum[wv][lane] = sum;
  __syncthreads();
  if (c >= sb.nc) return;
  uint32_t run = 0;
  for (uint32_t q = 0; q < wv; q++) run += s_sum[q][lane];
  if (wv == 15) sb.chist[c] = run + sum;
  for (uint32_t r = r0; r < r1; r++) { const size_t o = (size_t)r * sb.nc + c; const uint32_t v = sb.rows[o]; sb.rows[o] = run; run += v; }
}

// wbin[w] = the coarse bin of sorted tuple 64 w: the last bin that begins at or before it (empty bins begin where the next one does)
__global__ void __launch_bounds__(256) k_seed_wbin(SeedBufs sb) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n_all = min(sb.sn[SN_TUPLES], sb.cap_tuples);
  if ((unsigned long long)w * 64ull >= n_all) return;
  const uint32_t pos = w * 64u;
  uint32_t lo = 0, hi = sb.nc;                            // cbase[0] = 0 <= pos < n_all = cbase[nc]
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sb.cbase[mid] <= pos) lo = mid; else hi = mid; }
  sb.wbin[w] = (uint16_t)lo;
}

// exclusive prefix of cnt[0..nb) (nb <= 4096, block of 1024 threads) into out[], plus `add`; all threads must call it
__device__ __forceinline__ void block_excl_scan(const uint32_t* cnt, uint32_t* out, uint32_t nb, uint32_t* s_part /* [16] */) {
  const uint32_t t = threadIdx.x;
  uint32_t v[4], sum = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) { const uint32_t k = 4 * t + q; v[q] = k < nb ? cnt[k] : 0u; sum += v[q]; }
  uint32_t incl = sum;
  for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d, 64); if ((int)(t & 63u) >= d) incl += x; }
  if ((t & 63u) == 63u) s_part[t >> 6] = incl;
  __syncthreads();
  uint32_t pre = incl - sum;
  for (uint32_t q = 0; q < (t >> 6); q++) pre += s_part[q];
#pragma unroll
  for (int q = 0; q < 4; q++) { const uint32_t k = 4 * t + q; if (k < nb) out[k] = pre; pre += v[q]; }
  __syncthreads();
}

// One staged move of the tuples src[i0, i1) into `nb` bins whose next free places are cur[] (LDS, global indices into dst): piece by piece --
// load (each tuple once), take a place in its bin (LDS atomic), put the piece into LDS in bin order, copy it out with consecutive lanes on
// consecutive staged tuples (lanes of one bin's run write neighbouring addresses), rewritten by `conv` on the way.
// LDS: cur / pc0 / pst [nb], stage [SEED_PIECE].
// (Blocks of 1024 threads, and the code says so: `blockDim.x` is a global load from the dispatch packet, and the compiler put one -- with its s_waitcnt
// vmcnt(0), which on this target also waits for the piece's stores -- in front of every loop that strides by it.  All loads of a piece first and none of
// them under a branch (a lane past the piece reads the piece's last tuple): with `if (i < np) { load; atomic }` the compiler emitted load, s_waitcnt
// vmcnt(0), ds_add_rtn PER times over -- twelve round trips to memory one after the other per piece; round 6, profiles/r6s26_*: k_seed_split - 10 %.)
#ifndef SMR_SORT_PREFETCH
#define SMR_SORT_PREFETCH 0                               // 1: the next piece's loads are issued before the copy-out of this one (their registers are free by then)
#endif
#ifndef SMR_SORT_COPY_UNROLLED
#define SMR_SORT_COPY_UNROLLED 0                          // 1: the copy-out as PER unrolled steps (the staged tuples of all steps read first)
#endif
template <uint32_t PIECE, class BINOF, class CONV>
__device__ __forceinline__ void staged_move(const SeedTup* __restrict__ src, uint32_t i0, uint32_t i1, SeedTup* __restrict__ dst, uint32_t nb,
                                            uint32_t* cur, uint32_t* pc0, uint32_t* pst, SeedTup* stage, uint32_t* s_part, BINOF binof, CONV conv) {
  constexpr int PER = PIECE / 1024;
  const uint32_t tid = threadIdx.x;
  SeedTup mine[PER]; uint32_t place[PER];
  if (SMR_SORT_PREFETCH && i0 < i1) {
    const uint32_t np = min(PIECE, i1 - i0);
#pragma unroll
    for (int j = 0; j < PER; j++) mine[j] = src[i0 + min((uint32_t)j * 1024u + tid, np - 1u)];
  }
  for (uint32_t p0 = i0; p0 < i1; p0 += PIECE) {
    const uint32_t np = min(PIECE, i1 - p0);
    for (uint32_t q = tid; q < nb; q += 1024u) pc0[q] = cur[q];
    __syncthreads();
    if (!SMR_SORT_PREFETCH) {
#pragma unroll
      for (int j = 0; j < PER; j++) mine[j] = src[p0 + min((uint32_t)j * 1024u + tid, np - 1u)];
    }
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const uint32_t i = (uint32_t)j * 1024u + tid;
      if (i < np) place[j] = atomicAdd(&cur[binof(mine[j])], 1u);
    }
    __syncthreads();
    for (uint32_t q = tid; q < nb; q += 1024u) pst[q] = cur[q] - pc0[q];                   // the piece's tuples per bin ...
    __syncthreads();
    block_excl_scan(pst, pst, nb, s_part);                                                 // ... become where the bin's run starts in the staged piece
#pragma unroll
    for (int j = 0; j < PER; j++) {
      const uint32_t i = (uint32_t)j * 1024u + tid;
      if (i < np) { const uint32_t f = binof(mine[j]); stage[pst[f] + (place[j] - pc0[f])] = mine[j]; }
    }
    __syncthreads();
    if (SMR_SORT_PREFETCH && p0 + PIECE < i1) {
      const uint32_t q0 = p0 + PIECE, nq = min(PIECE, i1 - q0);
#pragma unroll
      for (int j = 0; j < PER; j++) mine[j] = src[q0 + min((uint32_t)j * 1024u + tid, nq - 1u)];
    }
    if (SMR_SORT_COPY_UNROLLED) {
      SeedTup tq[PER];
#pragma unroll
      for (int j = 0; j < PER; j++) tq[j] = stage[min((uint32_t)j * 1024u + tid, np - 1u)];
#pragma unroll
      for (int j = 0; j < PER; j++) {
        const uint32_t sidx = (uint32_t)j * 1024u + tid;
        const uint32_t f = binof(tq[j]);
        if (sidx < np) dst[pc0[f] + (sidx - pst[f])] = conv(tq[j]);
      }
    } else
    for (uint32_t sidx = tid; sidx < np; sidx += 1024u) {
      const SeedTup t = stage[sidx];
      const uint32_t f = binof(t);
      dst[pc0[f] + (sidx - pst[f])] = conv(t);
    }
    __syncthreads();
  }
}

template <uint32_t PIECE>                                   // (SEED_SPLIT_PIECE, or SEED_SPLIT_PIECE_MANY_BINS when the tables of 4 096 coarse bins leave less room)
__global__ void __launch_bounds__(1024) k_seed_split(SeedBufs sb) {
  SMR_DYN_LDS(uint32_t, lds);                             // cur | pc0 | pst [nc each, rounded to an even number of words] | stage
  __shared__ uint32_t s_part[16];
  const uint32_t ncw = (sb.nc + 1u) & ~1u;
  uint32_t* cur = lds; uint32_t* pc0 = lds + ncw; uint32_t* pst = lds + 2 * ncw;
  SeedTup* stage = reinterpret_cast<SeedTup*>(lds + 3 * ncw);
  const uint32_t nmine = sb.bcnt[blockIdx.x];
  if (nmine == 0) return;
  const uint32_t* row = sb.rows + (size_t)blockIdx.x * sb.nc;
  for (uint32_t c = threadIdx.x; c < sb.nc; c += 1024u) cur[c] = sb.cbase[c] + row[c];
  __syncthreads();
  const uint32_t fb = sb.fb, kbits = sb.kbits, cb = sb.cb;
  const uint32_t slot0 = blockIdx.x * sb.rpb * sb.maxwin;                  // the block's first slot
  staged_move<PIECE>(sb.tmp + (size_t)2 * slot0, 0u, nmine, sb.mid, sb.nc, cur, pc0, pst, stage, s_part,
              [fb, kbits](SeedTup t) { return ((uint32_t)t & ((1u << kbits) - 1u)) >> fb; },
              [fb, kbits, cb, slot0](SeedTup t) {
                const uint32_t key = (uint32_t)t & ((1u << kbits) - 1u), chars = (uint32_t)(t >> kbits) & ((1u << cb) - 1u), rel = (uint32_t)(t >> (kbits + cb));
                return (unsigned long long)(slot0 + rel) | ((unsigned long long)chars << 32) | ((unsigned long long)(key & ((1u << fb) - 1u)) << (32u + cb));
              });
}

// The keys of a batch that many windows share -- at least sb.hot_min tuples and four times the average per key -- are where k_seed_dedup looks
// for repeated seeds: their tuples as pieces of at most SEED_DD_PIECE
__device__ __forceinline__ uint32_t seed_hot_key(const SeedBufs& sb) { return max(sb.hot_min, min(sb.sn[SN_TUPLES], sb.cap_tuples) >> (sb.kbits - 2u)); }
__device__ __forceinline__ void seed_push_pieces(const SeedBufs& sb, uint32_t start, uint32_t cnt) {
  const uint32_t np = (cnt + SEED_DD_PIECE - 1u) / SEED_DD_PIECE;
  const uint32_t b = atomicAdd(&sb.sn[SN_PIECES], np);
  for (uint32_t q = 0; q < np && b + q < sb.cap_pieces; q++) sb.pieces[b + q] = make_uint2(start + q * SEED_DD_PIECE, min(SEED_DD_PIECE, cnt - q * SEED_DD_PIECE));
}

#ifndef SMR_BINS_BLOCKS_PER_CU
#define SMR_BINS_BLOCKS_PER_CU 1                         // (2: 8 waves per SIMD = at most 64 VGPRs)
#endif
__global__ void __launch_bounds__(1024, 4 * SMR_BINS_BLOCKS_PER_CU) k_seed_bins(SeedBufs sb) {
  __shared__ uint32_t cur[512], pc0[512], pst[512], s_part[16];
  SMR_DYN_LDS(uint32_t, lds);
  SeedTup* stage = reinterpret_cast<SeedTup*>(lds);
  const uint32_t c = blockIdx.x, lo = sb.cbase[c], hi = sb.cbase[c + 1];
  if (lo == hi || sb.hpre[c + 1] != sb.hpre[c]) return;     // (a large bin: k_seed_hbins_*)
  const uint32_t t = threadIdx.x, nf = 1u << sb.fb, sh = 32u + sb.cb;
  if (t < 512) pst[t] = 0;
  __syncthreads();
  for (uint32_t i = lo + t; i < hi; i += 1024u) atomicAdd(&pst[(uint32_t)(sb.mid[i] >> sh)], 1u);
  __syncthreads();
  block_excl_scan(pst, cur, nf, s_part);
  if (t < nf) {
    cur[t] += lo;                                         // the next free place of every fine bin
    if (sb.hot_min && pst[t] >= seed_hot_key(sb)) seed_push_pieces(sb, cur[t], pst[t]);
  }
  __syncthreads();
  staged_move<SEED_PIECE>(sb.mid, lo, hi, sb.srt, nf, cur, pc0, pst, stage, s_part, [sh](SeedTup x) { return (uint32_t)(x >> sh); }, [](SeedTup x) { return x; });
}

// A coarse bin far larger than the others (cscan: hpre) would keep ONE block of k_seed_bins busy long after the rest of the launch is over --
// on the reference's amplicon fixture a handful of keys hold most of the batch's windows: 8.7 ms for a pass that takes 1.6 ms on evenly spread keys
// (profiles/r5s37_bench_config2.json).  Such a bin is cut into sub-ranges of SEED_HOT_SUB tuples: k_seed_hbins_hist counts the fine bits of each,
// k_seed_hbins_scan turns the counts into the places of every sub-range's tuples (per fine bin in sub-range order: the sort stays stable), and
// k_seed_hbins_move is the staged move of k_seed_bins, one block per sub-range.
__device__ __forceinline__ bool seed_hot_entry(const SeedBufs& sb, uint32_t e, uint32_t& c, uint32_t& i0, uint32_t& i1) {
  uint32_t lo = 0, hi = sb.nc;                            // hpre[0] = 0 <= e < hpre[nc]: the last bin whose prefix is <= e
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (sb.hpre[mid] <= e) lo = mid; else hi = mid; }
  c = lo;
  const uint32_t s = e - sb.hpre[c], b0 = sb.cbase[c], b1 = sb.cbase[c + 1];
  i0 = b0 + s * sb.hsub; i1 = min(b1, i0 + sb.hsub);
  return i0 < i1;
}
__global__ void __launch_bounds__(1024) k_seed_hbins_hist(SeedBufs sb) {
  __shared__ uint32_t cnt[512];
  const uint32_t n_ent = min(sb.hpre[sb.nc], sb.cap_hent), nf = 1u << sb.fb, sh = 32u + sb.cb;
  for (uint32_t e = blockIdx.x; e < n_ent; e += gridDim.x) {
    uint32_t c, i0, i1;
    seed_hot_entry(sb, e, c, i0, i1);
    if (threadIdx.x < 512) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += 1024u) atomicAdd(&cnt[(uint32_t)(sb.mid[i] >> sh)], 1u);
    __syncthreads();
    if (threadIdx.x < nf) sb.hh[(size_t)e * nf + threadIdx.x] = cnt[threadIdx.x];
    __syncthreads();
  }
}
// a block per large coarse bin: per fine bin the running sum over the bin's sub-ranges (like k_seed_colscan: lane = fine bin, the 16 waves take a
// sixteenth of the sub-ranges each), then the fine bins' places in the coarse bin
__global__ void __launch_bounds__(1024) k_seed_hbins_scan(SeedBufs sb) {
  __shared__ uint32_t tot[512], fbase[512], s_part[16], s_sum[16][64];
  const uint32_t n_hot = min(sb.sn[SN_HOTBINS], sb.nc), nf = 1u << sb.fb;
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  for (uint32_t k = blockIdx.x; k < n_hot; k += gridDim.x) {
    const uint32_t c = sb.hlist[k], e0 = sb.hpre[c], e1 = min(sb.hpre[c + 1], sb.cap_hent), lo = sb.cbase[c];
    const uint32_t nsub = e1 > e0 ? e1 - e0 : 0u, per = (nsub + 15u) / 16u, r0 = e0 + min(wv * per, nsub), r1 = e0 + min(wv * per + per, nsub);
    if (threadIdx.x < 512) tot[threadIdx.x] = 0;
    for (uint32_t g = 0; g < nf; g += 64) {
      const uint32_t f = g + lane;
      uint32_t sum = 0;
      if (f < nf) for (uint32_t e = r0; e < r1; e++) sum += sb.hh[(size_t)e * nf + f];
      s_sum[wv][lane] = sum;
      __syncthreads();
      uint32_t run = 0;
      for (uint32_t q = 0; q < wv; q++) run += s_sum[q][lane];
      if (wv == 15 && f < nf) tot[f] = run + sum;
      if (f < nf) for (uint32_t e = r0; e < r1; e++) { const size_t o = (size_t)e * nf + f; const uint32_t v = sb.hh[o]; sb.hh[o] = run; run += v; }
      __syncthreads();
    }
    block_excl_scan(tot, fbase, nf, s_part);
    for (uint32_t g = 0; g < nf; g += 64) {
      const uint32_t f = g + lane;
      if (f < nf) { const uint32_t b = lo + fbase[f]; for (uint32_t e = r0; e < r1; e++) sb.hh[(size_t)e * nf + f] += b; }
    }
    if (threadIdx.x < nf && sb.hot_min && tot[threadIdx.x] >= seed_hot_key(sb)) seed_push_pieces(sb, lo + fbase[threadIdx.x], tot[threadIdx.x]);
    __syncthreads();
  }
}
__global__ void __launch_bounds__(1024) k_seed_hbins_move(SeedBufs sb) {
  __shared__ uint32_t cur[512], pc0[512], pst[512], s_part[16];
  SMR_DYN_LDS(uint32_t, lds);
  SeedTup* stage = reinterpret_cast<SeedTup*>(lds);
  const uint32_t n_ent = min(sb.hpre[sb.nc], sb.cap_hent), nf = 1u << sb.fb, sh = 32u + sb.cb;
  for (uint32_t e = blockIdx.x; e < n_ent; e += gridDim.x) {
    uint32_t c, i0, i1;
    seed_hot_entry(sb, e, c, i0, i1);
    if (threadIdx.x < nf) cur[threadIdx.x] = sb.hh[(size_t)e * nf + threadIdx.x];
    __syncthreads();
    staged_move<SEED_PIECE>(sb.mid, i0, i1, sb.srt, nf, cur, pc0, pst, stage, s_part, [sh](SeedTup x) { return (uint32_t)(x >> sh); }, [](SeedTup x) { return x; });
  }
}

// Repeated seeds.  The two half-seed searches of a window depend on nothing but the window's 18 letters (key + automaton chars = the tuple without
// its slot), and real samples repeat them: the reference's amplicon fixture has 664 145 first-pass windows per 100 000 reads and 17 948 different
// ones, the most frequent 35 186 times.  After the sort the tuples of a key lie together (in slot order, repeats interleaved with the key's other
// seeds); a block takes a piece of the tuples of ONE key with many tuples (sb.pieces) and enters their chars into a hash table in LDS: the first
// tuple with given chars stays what it was -- the REPRESENTATIVE, searched by k_seed_pg --, a later one is rewritten in place as {slot,
// representative's slot, SEED_TUP_DUP}.  No search looks at it; k_seed_prop<DIR> hands its window the representative's segment after the
// launch of its direction (forward before the reverse launch: that one asks for the window's zero bit).  Which of several equal tuples becomes
// the representative is a race the results do not depend on.  A piece with more different seeds than the table takes leaves the rest as they are.
__global__ void __launch_bounds__(256) k_seed_dedup(SeedBufs sb) {
  __shared__ unsigned long long tab[SEED_DD_TAB];
  const uint32_t n_p = min(sb.sn[SN_PIECES], sb.cap_pieces);
  for (uint32_t p = blockIdx.x; p < n_p; p += gridDim.x) {
    const uint2 pc = sb.pieces[p];
    for (uint32_t q = threadIdx.x; q < SEED_DD_TAB; q += blockDim.x) tab[q] = 0ull;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < pc.y; i += blockDim.x) {
      const SeedTup t = sb.srt[pc.x + i];
      const uint32_t v = (uint32_t)(t >> 32) + 1u, slot = (uint32_t)t;        // chars | fine bits (the same for the whole piece), never 0
      uint32_t h = (v * 2654435761u) >> 20;                                   // 12 bits
      for (int probe = 0; probe < 8; probe++, h = (h + 1u) & (SEED_DD_TAB - 1u)) {
        const unsigned long long old = atomicCAS(&tab[h], 0ull, ((unsigned long long)v << 32) | slot);
        if (old == 0ull) break;                                               // the first of its kind
        if ((uint32_t)(old >> 32) == v) { sb.srt[pc.x + i] = (unsigned long long)slot | ((unsigned long long)(uint32_t)old << 32) | SEED_TUP_DUP; break; }
      }
    }
    __syncthreads();
  }
}
// the windows of the repeated seeds of direction DIR get their representatives' hit segments (the pool segment itself is shared: k_seed_finish only reads it)
template <int DIR>
__global__ void __launch_bounds__(256) k_seed_prop(SeedBufs sb, unsigned long long* __restrict__ ctr) {
  const uint32_t n_p = min(sb.sn[SN_PIECES], sb.cap_pieces);
  const uint32_t n_fwd = min(sb.sn[SN_FWD], sb.cap_tuples);
  unsigned long long moved = 0;
  for (uint32_t p = blockIdx.x; p < n_p; p += gridDim.x) {
    const uint2 pc = sb.pieces[p];
    if ((pc.x >= n_fwd) != (DIR == 1)) continue;           // (a key is forward or reverse: so is a piece)
    for (uint32_t i = threadIdx.x; i < pc.y; i += blockDim.x) {
      const SeedTup t = sb.srt[pc.x + i];
      if (!(t & SEED_TUP_DUP)) continue;
      const uint32_t slot = (uint32_t)t, rep = (uint32_t)(t >> 32) & 0x7FFFFFFFu;
      moved += 9u;                                         // the tuple, the representative's bit
      if (wseg_has(sb, DIR, rep)) { const uint32_t v = sb.wseg[DIR][rep]; wseg_put(sb, DIR, slot, v, (v & SEED_ZERO_BIT) != 0); moved += 8u; }
    }
  }
  for (int d = 32; d > 0; d >>= 1) moved += __shfl_down(moved, d, 64);
  if (lane_id() == 0 && moved) ctr_add(ctr, DIR ? C_B_PG1 : C_B_PG0, moved);
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
__device__ __forceinline__ void seed_search_wave(const uint32_t* __restrict__ arena, const uint32_t* __restrict__ pos_off, uint32_t root, bool mine, uint32_t chars, uint32_t pw, bool full,
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
      const uint32_t bk = f_bk, q = f_q, pbv = f_pbv, str = f_str;
      uint32_t id = f_id;
      if (base + 64 < T) fetch(base + 64);
      const uint32_t olane = bk / SEED_K;
      const uint32_t nchar = pbv >> 24;                      // chars of the trie path in front of the tail
      const uint32_t tstr = (pbv & 0xFFFFFFu) | (str << (2 * nchar));
      const uint32_t r = v ? lev1_entry(L.pat[olane], tstr, pw) : 0u;
      if (r & 1u) id = pos_off[id] + id;                       // (what the searches hand on is the place of the seed's position list: k_pos2_build; the arena holds the index's ids)
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

// DIR 0: the forward searches of a wave chunk of the sorted tuples, DIR 1: its reverse searches (the lanes of the other direction idle: the
// chunk that holds the last forward and the first reverse tuple is the only mixed one).  The reverse search starts from the window's forward
// list and leaves the FINAL list of the window (SEED_SEG_MERGED), as the reference's two calls of traversetrie_align on one list do.
template <int DIR>
__global__ void __launch_bounds__(64) k_seed_search(DIndex ix, DParams P, int pass, SeedBufs sb, uint32_t hcap,
                                                    uint32_t* __restrict__ pool, uint32_t pool_words, unsigned long long* __restrict__ ctr,
                                                    const uint32_t* __restrict__ redo) {
  const uint32_t n_tup = min(sb.sn[SN_TUPLES], sb.cap_tuples);
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
  const uint32_t pos = wave * 64u + lane;
  bool mine = pos < n_tup;
  uint32_t chars = 0, root = 0, slot = 0;
  SeedLane sl; sl.nh = 0; sl.zero = false; sl.overflow = false; sl.n_node = 0; sl.n_entry = 0;
  uint32_t n_prev = 0;
  bool counted = false;                                  // a tuple of this direction (its algorithmic bytes are counted here)
  if (mine) {
    if ((pos >= min(sb.sn[SN_FWD], n_tup)) != (DIR == 1)) mine = false;      // the forward tuples lie in front
    else {
      const SeedKey tk = seed_decode(sb, pos);
      if (tk.dup || !seed_read_active(sb, tk.slot)) mine = false;      // a repeated seed: its window gets the representative's segment (k_seed_prop); a read that is not in this launch
      else {
      counted = true;
      const Lookup lk = ix.lookup[tk.key - (DIR ? sb.nkh : 0u)];
      root = DIR == 0 ? lk.rootF : lk.rootR;
      chars = tk.chars; slot = tk.slot;
      if (root == NONE) mine = false;                      // (a shared sort holds the tuples of every key; a part without the mini-trie has nothing to search)
      }
      if (mine && DIR == 1 && wseg_has(sb, 0, slot)) {     // the window's list so far = the forward search's hits
        const uint32_t seg = sb.wseg[0][slot];
        if (seg & SEED_ZERO_BIT) mine = false;             // accept_zero_kmer: no reverse search (paralleltraversal.cpp:188)
        else if (seg & SEED_SEG_INLINE) { n_prev = 1; hl[lane] = seg & SEED_SEG_ID; sl.nh = 1; }
        else {
          const uint32_t o = seg & ~SEED_ZERO_BIT;
          n_prev = pool[o] & 0xFFFFu;
          for (uint32_t q = 0; q < n_prev && q < hcap; q++) hl[q * 64 + lane] = pool[o + 1 + q];
          if (n_prev > hcap) { sl.overflow = true; n_prev = hcap; }
          sl.nh = n_prev;
        }
      }
    }
  }
  __syncthreads();
#ifdef SMR_SEED_PHASES
  unsigned long long sph[6] = {0, 0, 0, 0, 0, 0}, slast = clock64();
  seed_search_wave(ix.trie, ix.pos_off, root, mine, chars, P.partialwin, P.is_full_search != 0, s_row, L, hcap, sl, sph, slast);
#else
  seed_search_wave(ix.trie, ix.pos_off, root, mine, chars, P.partialwin, P.is_full_search != 0, s_row, L, hcap, sl);
#endif
  // ---- write the windows' hit segments: [count (| SEED_SEG_MERGED), id x count] ----
  const bool wr = mine && (DIR == 0 ? sl.nh > 0 : (sl.zero || sl.nh > n_prev));
  const uint32_t need = wr ? 1 + sl.nh : 0;
  uint32_t incl = need;
  for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
  const uint32_t total = __shfl(incl, 63, 64);
  uint32_t base = 0;
  if (total) {
    if (lane == 0) {                                     // the pool is split into C_NSHARD regions, each with its own cursor
      const uint32_t shard = blockIdx.x & (C_NSHARD - 1), region = pool_words / C_NSHARD;
      const unsigned long long old = atomicAdd(&ctr[C_PCUR + shard * C_PCUR_STRIDE], (unsigned long long)total);
      if (old + total > region) { atomicAdd(&ctr[C_ERR_POOL], 1ull); base = NONE; } else base = shard * region + (uint32_t)old;
    }
    base = __shfl(base, 0, 64);
  }
  if (wr && base != NONE) {
    const uint32_t o = base + incl - need;
    pool[o] = sl.nh | (DIR ? SEED_SEG_MERGED : 0u);
    for (uint32_t q = 0; q < sl.nh; q++) pool[o + 1 + q] = hl[q * 64 + lane];
    wseg_put(sb, DIR, slot, o | (sl.zero ? SEED_ZERO_BIT : 0u), sl.zero);
  }
  if (__any(sl.overflow) && lane == 0) atomicAdd(&ctr[C_ERR_HITCAP], 1ull);
  // algorithmic bytes of this wave (C_B_PG0/1): tuple + lookup entry per search, 16 B per node, 8 B per entry, the forward list read (DIR 1),
  // the segment written + its window slot
  unsigned long long v[3] = {sl.n_node, sl.n_entry, 0};
  v[2] = (counted ? sizeof(SeedTup) + sizeof(Lookup) + (DIR ? 1u + (n_prev ? 8u + 4u * n_prev : 0u) : 0u) : 0u) + 16ull * sl.n_node + 8ull * sl.n_entry + 4ull * need + (wr ? 4u : 0u);
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

// per read: merge the forward and reverse hit segments of this pass's windows into ONE contiguous block of (id, win_pos) pairs (k_chain then
// reads a strand's cumulative hits with coalesced loads instead of chasing a list), count seeds/hits (++read.hit_seeds per window with hits,
// paralleltraversal.cpp:242-249), make the 0..3 view persistent (Read::flip34, read.cpp:379-401).
// The window's list is what the reference's two searches leave on one list (traverse_bursttrie.cpp:256-277): the forward hits; nothing more if
// the forward search ended with a 0-error match (:188); else the reverse search's candidates in its DFS order -- one already present is
// skipped, a 0-error candidate (SEED_CAND_COND) that is not present REPLACES the list and ends the window, any other is appended.  A reverse
// segment marked SEED_SEG_MERGED (k_seed_search<1>) is that final list already.
// bit per read: its windows are searched in this launch through the shared sort (one sort for several index parts: SeedBufs::abits)
__global__ void __launch_bounds__(256) k_seed_active(uint32_t n, int pass, int with_amb, const RWork* __restrict__ rw, uint32_t* __restrict__ abits) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  bool a = false;
  if (r < n) { const RWork w = rw[r]; a = w.strand_active && w.search && w.pass_n == (uint32_t)pass && (with_amb || !w.has_amb); }
  const unsigned long long m = __ballot(a);
  if ((threadIdx.x & 63u) == 0) { abits[r >> 5] = (uint32_t)m; abits[(r >> 5) + 1u] = (uint32_t)(m >> 32); }      // (abits has room for the grid's last wave)
}

#define FIN_KEEP 8u                                       // segments per read and pass whose place k_seed_finish remembers (20 KB of LDS per block; a read from the DB has one per window of the first pass)
__global__ void __launch_bounds__(256) k_seed_finish(DReads rd, DParams P, int pass, SeedBufs sb, RState* __restrict__ work,
                                                     RWork* __restrict__ rw, uint32_t* __restrict__ pool, uint32_t pool_words,
                                                     unsigned long long* __restrict__ ctr) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  __shared__ uint32_t s_sf[FIN_KEEP][256], s_sr[FIN_KEEP][256];     // the first windows with segments a thread met: forward / reverse segment (NONE: none),
  __shared__ uint16_t s_k[FIN_KEEP][256];                           // ... window (reads <= 65 535 letters; 16 bits: 20.5 KB per block = 7 blocks per CU, with 32 bits 24.6 KB = 6)
  __shared__ unsigned long long s_st[6];
  if (threadIdx.x < 6) s_st[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long hits = 0, bytes = 0, looks = 0, moved = 0, kin = 0, wins = 0;      // moved: algorithmic bytes of this read (C_B_FIN); kin: what k_seed_keys read for it (C_B_KEYS)
  if (r < rd.n) {
    RWork w = rw[r];
    const uint32_t len = rd.len[r];                        // (asked for with the state, not after it)
    moved = sizeof(RWork);
    kin = sizeof(RWork) + 12u;                             // k_seed_keys looks at every read's state, length and record offset ...
    if (w.strand_active && w.search && w.pass_n == (uint32_t)pass) {
      const uint32_t k0s = P.skip[0], k1s = P.skip[1], stride = pass == 0 ? k0s : pass == 1 ? k1s : P.skip[2];
      const uint32_t numwin = (len - P.lnwin + stride) / stride;
      const uint32_t slot0 = r * sb.maxwin;
      uint32_t seeds = 0, upper = 0, rlook = 0, nsearched = 0;
      uint32_t rem0 = 0, rem1 = 0;                           // (k * stride) % skip[0], % skip[1], kept by addition
      // the windows' bits, 32 at a time: a read's windows are consecutive bits of fbits[0] / fbits[1]
      auto bits32 = [&](int d, uint32_t b0) -> uint32_t {
        const uint32_t i = b0 >> 5, s = b0 & 31u;
        const uint32_t lo = sb.fbits[d][i], hi = sb.fbits[d][i + 1];
        return s ? (lo >> s) | (hi << (32u - s)) : lo;
      };
      // the merged list of window k (segments sf / sr, NONE = none) appended at pool[o...] as (id, win_pos) pairs when o != NONE; returns its length
      auto merge = [&](uint32_t sf, uint32_t sr, uint32_t win_pos, uint32_t o) -> uint32_t {
        uint32_t n = 0;
        const bool fzero = sf != NONE && (sf & SEED_ZERO_BIT);
        // (a one-hit window's hit lies in the word itself: SEED_SEG_INLINE -- no pool line is touched for it)
        const bool fin = sf != NONE && (sf & SEED_SEG_INLINE), rin = sr != NONE && (sr & SEED_SEG_INLINE);
        const uint32_t of = sf & ~SEED_ZERO_BIT, orv = sr & ~SEED_ZERO_BIT;
        const uint32_t hr = sr == NONE ? 0u : rin ? 1u : pool[orv];
        const uint32_t nf = sf == NONE ? 0u : fin ? 1u : (pool[of] & 0xFFFFu), nr = hr & 0xFFFFu;
        auto fwd_id = [&](uint32_t q) -> uint32_t { return fin ? (sf & SEED_SEG_ID) : pool[of + 1 + q]; };
        auto rev_id = [&](uint32_t q) -> uint32_t { return rin ? (sr & (SEED_SEG_ID | SEED_CAND_COND)) : pool[orv + 1 + q]; };
        if (sr != NONE && (hr & SEED_SEG_MERGED) && !fzero) {
          for (uint32_t q = 0; q < nr; q++) { if (o != NONE) { pool[o + 2 * n] = pool[orv + 1 + q]; pool[o + 2 * n + 1] = win_pos; } n++; }
          return n;
        }
        for (uint32_t q = 0; q < nf; q++) { if (o != NONE) { pool[o + 2 * n] = fwd_id(q); pool[o + 2 * n + 1] = win_pos; } n++; }
        if (fzero || sr == NONE) return n;
        for (uint32_t q = 0; q < nr; q++) {
          const uint32_t c = rev_id(q), id = c & ~SEED_CAND_COND;
          bool present = false;
          for (uint32_t f = 0; f < nf; f++) if (fwd_id(f) == id) { present = true; break; }
          if (present) continue;
          if (c & SEED_CAND_COND) { if (o != NONE) { pool[o] = id; pool[o + 1] = win_pos; } return 1u; }
          if (o != NONE) { pool[o + 2 * n] = id; pool[o + 2 * n + 1] = win_pos; }
          n++;
        }
        return n;
      };
      for (uint32_t kb = 0; kb < numwin; kb += 32) {
        const uint32_t nk = min(32u, numwin - kb);
        const uint32_t vm = nk < 32 ? (1u << nk) - 1u : 0xFFFFFFFFu;
        const uint32_t mf = bits32(0, slot0 + kb) & vm, mr = bits32(1, slot0 + kb) & vm;
        uint32_t srch = 0;                                   // windows of this pass: not searched by an earlier pass (:128-131)
        for (uint32_t j = 0; j < nk; j++) {
          const bool earlier = (pass >= 1 && rem0 == 0) || (pass >= 2 && rem1 == 0);
          srch |= (earlier ? 0u : 1u) << j;
          rem0 += stride; while (rem0 >= k0s) rem0 -= k0s;
          rem1 += stride; while (rem1 >= k1s) rem1 -= k1s;
        }
        nsearched += (uint32_t)__popc(srch);
        rlook += (uint32_t)__popc(srch & ~mf);             // no forward segment: the reverse lookup happened (:188-198)
        for (uint32_t mm = mf | mr; mm; mm &= mm - 1) {
          const uint32_t j = (uint32_t)__ffs((int)mm) - 1u, k = kb + j;
          const uint32_t sf = ((mf >> j) & 1u) ? sb.wseg[0][slot0 + k] : NONE, sr = ((mr >> j) & 1u) ? sb.wseg[1][slot0 + k] : NONE;
          if (sf != NONE && ((srch >> j) & 1u) && !(sf & SEED_ZERO_BIT)) rlook++;        // ... unless the forward search hit exactly
          const uint32_t cf = sf == NONE ? 0u : (sf & SEED_SEG_INLINE) ? 1u : (pool[sf & ~SEED_ZERO_BIT] & 0xFFFFu), cr = sr == NONE ? 0u : (sr & SEED_SEG_INLINE) ? 1u : (pool[sr & ~SEED_ZERO_BIT] & 0xFFFFu);
          if (seeds < FIN_KEEP) { s_sf[seeds][threadIdx.x] = sf; s_sr[seeds][threadIdx.x] = sr; s_k[seeds][threadIdx.x] = (uint16_t)k; }
          seeds++; upper += cf + cr;
        }
      }
      looks = rlook + nsearched;                             // the forward lookup of every window of this pass + the reverse ones that happened
      wins = nsearched;
      kin += 4u * (((len + 15) >> 4) + ((len + 31) >> 5));      // ... and of an active read its packed record
      uint32_t base = 0, total = 0;
      if (upper) {                                           // room for the longest the merged lists can be; blk_cnt is what they are
        const uint32_t shard = blockIdx.x & (C_NSHARD - 1), region = pool_words / C_NSHARD;
        const unsigned long long old = atomicAdd(&ctr[C_PCUR + shard * C_PCUR_STRIDE], 2ull * upper);
        if (old + 2ull * upper > region) { atomicAdd(&ctr[C_ERR_POOL], 1ull); upper = 0; seeds = 0; }
        else base = shard * region + (uint32_t)old;
      }
      unsigned long long segw = 0;                           // segment words read
      if (upper && seeds <= FIN_KEEP) {
        for (uint32_t i = 0; i < seeds; i++) {
          const uint32_t sf = s_sf[i][threadIdx.x], sr = s_sr[i][threadIdx.x];
          total += merge(sf, sr, (uint32_t)s_k[i][threadIdx.x] * stride, base + 2 * total);
        }
      } else if (upper) for (uint32_t kb = 0; kb < numwin; kb += 32) {
        const uint32_t nk = min(32u, numwin - kb);
        const uint32_t vm = nk < 32 ? (1u << nk) - 1u : 0xFFFFFFFFu;
        const uint32_t mf = bits32(0, slot0 + kb) & vm, mr = bits32(1, slot0 + kb) & vm;
        for (uint32_t mm = mf | mr; mm; mm &= mm - 1) {
          const uint32_t j = (uint32_t)__ffs((int)mm) - 1u, k = kb + j;
          const uint32_t sf = ((mf >> j) & 1u) ? sb.wseg[0][slot0 + k] : NONE, sr = ((mr >> j) & 1u) ? sb.wseg[1][slot0 + k] : NONE;
          total += merge(sf, sr, k * stride, base + 2 * total);
        }
      }
      segw = upper + seeds;
      // length, two bits per window, per segment its slot and its words, every (id, win_pos) pair written, per-read state written, hit_seeds
      moved += 4u + (numwin + 3u) / 4u + 8ull * seeds + 4ull * segw + 8ull * total + sizeof(RWork) + 8u;
      w.aval = w.is04 ? 0 : w.aval; w.is04 = 0;
      // (not w.blk_off[pass]: an index the compiler cannot resolve moves the whole state to LDS, 12 KB per block)
      if (pass == 0) { w.blk_off[0] = base; w.blk_cnt[0] = total; } else if (pass == 1) { w.blk_off[1] = base; w.blk_cnt[1] = total; } else { w.blk_off[2] = base; w.blk_cnt[2] = total; }
      w.hit_total += total;
      rw[r] = w;
      work[r].hit_seeds += seeds;
      hits = total; bytes = (len + 3) / 4;
    }
  }
  for (int d = 32; d > 0; d >>= 1) { hits += __shfl_down(hits, d, 64); bytes += __shfl_down(bytes, d, 64); looks += __shfl_down(looks, d, 64); moved += __shfl_down(moved, d, 64); kin += __shfl_down(kin, d, 64); wins += __shfl_down(wins, d, 64); }
  // the block's six sums leave with one atomic each (not one per wave)
  if (lane_id() == 0) { atomicAdd(&s_st[0], hits); atomicAdd(&s_st[1], bytes); atomicAdd(&s_st[2], looks); atomicAdd(&s_st[3], moved); atomicAdd(&s_st[4], kin); atomicAdd(&s_st[5], wins); }
  __syncthreads();
  if (threadIdx.x < 6 && s_st[threadIdx.x]) {
    const int which = threadIdx.x == 0 ? C_HIT : threadIdx.x == 1 ? C_READ_BYTES : threadIdx.x == 2 ? C_LOOKUP : threadIdx.x == 3 ? C_B_FIN : threadIdx.x == 4 ? C_B_KEYS : C_WINDOWS;
    ctr_add(ctr, which, s_st[threadIdx.x]);
  }
}

}  // namespace smr
