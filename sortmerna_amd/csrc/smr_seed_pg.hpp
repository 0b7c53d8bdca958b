// smr_seed_pg.hpp -- the pigeonhole seed search (included by smr_kernels.hpp after smr_seed.hpp).
#pragma once

namespace smr {

// ------------------------------------------------------------------------------------------------
// k_seed_pg<DIR>: the same searches as k_seed_search (one lane = one window's half-seed search), without walking the trie.
//
// lev1_entry (smr_seed.hpp) says which complete candidate strings T (pw+1 chars) a pattern P (pw chars) accepts: with a = the
// common prefix of P and T,  a + s0 >= pw-1  or  a + s1 >= pw  or  a + s2 >= pw-1,  s0/s1/s2 = the lengths of the runs of
// T[i]==P[i] / T[i+1]==P[i] ending at i = pw-1 / T[i]==P[i+1] ending at i = pw-2.  With h = pw/2 either
//   A   a >= h:  T[0..h-1] == P[0..h-1]
// or a <= h-1, and then the run of the accepting term covers every position >= h:
//   S0  T[h..pw-1] == P[h..pw-1]        S1  T[h..pw-1] == P[h-1..pw-2]        S2  T[h..pw-2] == P[h+1..pw-1]
// Each is an EXACT key, so the index is stored a third time (smr_host.hpp: per mini-trie the entries sorted by string with a
// directory over the first cA <= h chars, and sorted by chars h.. with a directory over cB <= pw-h of them): a search reads four
// directory ranges, applies lev1_entry to the few entries inside (about n/4^cA + 3 n/4^cB instead of all n of the mini-trie) and
// collects the accepted ones.  The reference's sequential semantics (traverse_bursttrie.cpp:100-298: DFS order A<C<G<T, 0-error
// match clears the list and ends the search, duplicate `break`) are restored afterwards: each search applies ITS candidates in
// DFS order, the order key being the entry's rank in the reference's traversal stored with it.  An entry reachable through several of
// the four keys is taken from the first only.  Results are identical to k_seed_search (tests compare both); the work counters
// count the entries looked at and the directory ranges read.  The candidates of the 64 searches share a pool of `ccap` records in
// LDS, each search chaining its own; a wave whose pool overflows hands its 64 tuples to k_seed_search through the redo list (and
// the host doubles ccap for the next launches when that happens to more than a few waves).
// ------------------------------------------------------------------------------------------------
#define PG_CAND_CAP0 256u                                 // initial pool size
#define PG_CAND_CAP_MAX 2048u
#define PG_NIL 0xFFFFu
#ifndef PG_OCC
#define PG_OCC 7                                          // waves per SIMD the kernel is compiled for: at 8 its ~115 wave-uniform values do not fit the 96 SGPRs a wave
#endif                                                    // then gets (37 spilled to VGPR lanes, 124 v_readlane / v_writelane): 7 x 112 SGPRs was 1.5 - 3 % faster (r4s41, r4s42)
#define PG_WAVES 1                                        // (measured in round 2: 4 waves per block 1.70 ms, one wave per block 1.57 ms; the chunk loop of k_seed_pg assumes one)
#ifndef PG_TRIP
#define PG_TRIP 4
#endif
// dynamic LDS words: candidates (rank, id, next | kind << 16), the hit lists (in all as many entries as there are candidates: a search's list is
// as long as its chain at most, so the lists lie back to back at offsets taken from the chain lengths -- no capacity per search, and 4.6 KB per wave
// whatever the longest list of the batch: with 64 x hcap words per wave the bench batch, whose longest list has 16 entries, ran 5 waves per SIMD
// instead of the 7 the registers allow, 16 % slower than at 7: profiles/r5s30_*), chain heads of the 64 searches, a row's owners / the chain lengths
#define PG_LDS_WORDS(ccap) (4u * (ccap) + 128u)

// inclusive prefix sum / maximum over the 64 lanes in six DPP steps: row_shr:1/2/4/8 inside the rows of 16, then row_bcast:15 into rows 1 and 3 and
// row_bcast:31 into rows 2 and 3 (a lane without a source adds / compares 0)
__device__ __forceinline__ uint32_t pg_scan_add(uint32_t x) { return wave_scan_add(x); }     // (smr_device_ops.hpp: six v_add_u32_dpp)
__device__ __forceinline__ uint32_t pg_scan_max(uint32_t x) {
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false));
  return x;
}

// the chars of a 2-bit packed string (char j at bits 2j) with char j moved to bits 30-2j: any run of chars is then a number with its
// first char most significant
__device__ __forceinline__ uint32_t pg_reversed(uint32_t s) {
  const uint32_t rv = __brev(s);
  return ((rv & 0xAAAAAAAAu) >> 1) | ((rv & 0x55555555u) << 1);
}
// chars from..from+cnt-1 (cnt >= 1) of a string given as pg_reversed
__device__ __forceinline__ uint32_t pg_rkey(uint32_t rev, uint32_t from, uint32_t cnt) { return (rev >> (32u - 2u * (from + cnt))) & ((1u << (2u * cnt)) - 1u); }

struct __attribute__((aligned(4))) PgPair { uint32_t lo, hi; };       // two neighbouring directory words (4-byte aligned: global_load_dwordx2 takes that)

// One row of 64 strings of a wave's searches (see the string loop of k_seed_pg): what the lane that got string g of the wave knows about it.
struct PgRow { uint32_t T, P, m, u, w; int s; SMR_GLOBAL_U32* tt; bool have; };     // tt: the strings of the search's block
// the searches' state that a string's lane fetches from its owner, prepared so that the lane has little left to do: string g of the wave is
// string g + d[w] of the search's block, w = the number of thresholds t1 <= t2 <= t3 (first wave string of the ranges S0, S1, S2) that g has
// reached; blo / bhi: the ADDRESS of the block's strings; mab: the masks of the two directory keys (cA, cB <= 8 chars: 16 bits each);
// exm: the first wave string of the search, 2^31 when it has none
struct PgOwn { uint32_t exm, P9, mab, blo, bhi, t1, t2, t3, d0, d1, d2, d3; };
// where the strings of a block begin: behind its two directories when it has them
__device__ __forceinline__ const uint32_t* pg_strings_at(const uint32_t* pg, uint32_t rtx, uint32_t rty) {
  const uint32_t cA = (rty >> 24) & 15u, cB = rty >> 28;
  return pg + (size_t)rtx * 4 + (cA ? (1u << (2 * cA)) + (1u << (2 * cB)) + 2u : 0u);
}
__device__ __forceinline__ SMR_GLOBAL_U32* pg_ptr(uint32_t lo, uint32_t hi) { return (SMR_GLOBAL_U32*)((unsigned long long)hi << 32 | lo); }

// Row g0: find the owners, ISSUE the loads of the strings (nothing here waits for them).
// The last search whose strings start at or before g: the searches that start inside the row leave their number at their first string, a
// prefix maximum spreads it to the strings behind (the searches are in the order of their first strings; one without strings starts where
// the next one does and loses against it); carry = 1 + the last search that starts before the row.
__device__ __forceinline__ void pg_row_fetch(PgRow& R, uint32_t g0, uint32_t wtot, int lane, uint32_t* own, const PgOwn& O, uint32_t& carry, const uint32_t* pg) {
  // (A row past the last string only issues its load.  What a row's lane knows about its string is written on ONE path and otherwise left as it
  // was -- nobody looks at it when `have` is false --: values that come out of both sides of a branch cost the wave a register copy each, and
  // round 5's version, which cleared the row first and filled it under `g < wtot`, spent ten of its ninety vector instructions per row on them.)
  uint32_t u = 0;
  SMR_GLOBAL_U32* tt = (SMR_GLOBAL_U32*)pg;
  bool have = false;
  if (g0 < wtot) {
    const uint32_t g = g0 + (uint32_t)lane;
    own[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    if (O.exm - g0 < 64u) own[O.exm - g0] = (uint32_t)lane + 1u;
    __builtin_amdgcn_wave_barrier();
    const uint32_t ow = max(pg_scan_max(own[lane]), carry);
    carry = (uint32_t)__builtin_amdgcn_readlane((int)ow, 63);
    const int s = (int)ow - 1;
    const uint32_t oP = __shfl(O.P9, s, 64), om = __shfl(O.mab, s, 64), olo = __shfl(O.blo, s, 64), ohi = __shfl(O.bhi, s, 64);
    const uint32_t t1 = __shfl(O.t1, s, 64), t2 = __shfl(O.t2, s, 64), t3 = __shfl(O.t3, s, 64);
    const uint32_t d0 = __shfl(O.d0, s, 64), d1 = __shfl(O.d1, s, 64), d2 = __shfl(O.d2, s, 64), d3 = __shfl(O.d3, s, 64);
    have = g < wtot;
    const bool p1 = g >= t1, p2 = g >= t2, p3 = g >= t3;
    u = have ? g + (p3 ? d3 : p2 ? d2 : p1 ? d1 : d0) : 0u;
    tt = have ? pg_ptr(olo, ohi) : (SMR_GLOBAL_U32*)pg;
    R.P = oP; R.m = om; R.w = p3 ? 3u : p2 ? 2u : p1 ? 1u : 0u; R.s = s;
  }
  // exactly ONE load per call whatever the path (a lane without a string reads word 0 of the layout): only then can the compiler let the
  // wave wait for the older of two loads in flight (s_waitcnt vmcnt(1)) instead of for all of them
  R.u = u; R.tt = tt; R.have = have;
  R.T = tt[u];
}
// The automaton over the strings of a row; an accepted one becomes a candidate record of its search.
__device__ __forceinline__ void pg_row_apply(const PgRow& R, uint32_t pw, uint32_t h, bool full, uint32_t ccap, uint32_t* s_ncand,
                                             uint32_t* cdk, uint32_t* cdv, uint32_t* cdn, uint32_t* hd) {
  if (!R.have) return;
  const uint32_t T = R.T, oP = R.P, u = R.u, w = R.w;
  const uint32_t mA = R.m & 0xFFFFu, mB = R.m >> 16;
  // reachable through an earlier key of its search?  Ranges S0, S1, S2: under key A; S2: also under S0 / S1
  const uint32_t tb = (T >> (2 * h)) & mB;
  const bool dup = (w != 0 && ((T ^ oP) & mA) == 0) || (w == 3 && (tb == ((oP >> (2 * h)) & mB) || tb == ((oP >> (2 * h - 2)) & mB)));
  const uint32_t r = dup ? 0u : lev1_entry(oP, T, pw);
  if (r & 1u) {
    const uint32_t p = atomicAdd(s_ncand, 1u);
    if (p < ccap) {
      // (only WHERE its {rank, id} lies: the load itself would sit in the pipelined loop and make the wave wait for it -- and, loads
      // completing in order, for the next row's strings -- in every row with an accepted string; pg_resolve fetches them all at once)
      cdk[p] = u | ((uint32_t)R.s << 25);
      cdn[p] = atomicExch(&hd[R.s], p) | ((((r & 2u) && !full) ? CK_COND : CK_PLAIN) << 16);
    }
  }
}

// The {DFS rank, id} of every accepted string of the wave, 64 records per trip: record p holds its string's number in the block and its
// search; the search's lane has the block ({offset, n | cA << 24 | cB << 28}).
__device__ __forceinline__ void pg_resolve(uint32_t nrec, int lane, uint32_t blo, uint32_t bhi, uint32_t rty, uint32_t* cdk, uint32_t* cdv, uint32_t* cnt) {
  for (uint32_t p0 = 0; p0 < nrec; p0 += 64) {
    const uint32_t p = p0 + (uint32_t)lane;
    const uint32_t rec = p < nrec ? cdk[p] : 0u;
    const int s = (int)(rec >> 25) & 63;
    const uint32_t olo = __shfl(blo, s, 64), ohi = __shfl(bhi, s, 64), om = __shfl(rty, s, 64);
    if (p < nrec) {
      const uint32_t u = rec & 0x1FFFFFFu, on = om & 0xFFFFFFu, ocA = (om >> 24) & 15u;
      SMR_GLOBAL_U32* ri = pg_ptr(olo, ohi) + (ocA ? 2 : 1) * (size_t)on + 2 * (size_t)u;      // (two neighbouring words: one 8-byte load)
      cdk[p] = ri[0]; cdv[p] = ri[1];
      atomicAdd(&cnt[s], 1u);                              // the length of the search's chain (one LDS instruction per 64 records)
    }
  }
}

// DIR 0: the forward searches (the sorted tuples in front of SN_FWD), DIR 1: the reverse searches, launched after them.  Round 3's reverse
// search started from the window's forward list: per tuple a random probe for the window's segment, then the list itself.  A reverse search
// does not need the forward list to FIND its candidates, only to apply them; so it leaves them, in its DFS order and without repeats, as the
// window's reverse segment (a 0-error candidate carries SEED_CAND_COND), and k_seed_finish, which walks the windows of a read anyway,
// applies them to the forward list (see there).  What it does need to know is whether the forward search ended with a 0-error match -- then
// there is no reverse search (paralleltraversal.cpp:188), and those are exactly the searches with many accepted strings --: one bit per
// window (zbits), asked for only where the bit of the window's group of 64 slots is set (gflag: under a megabyte, stays in the L2s).
// `swz`: wave it works on chunk (it % 8) * ceil(chunks / 8) + it / 8 -- blocks run on XCD b % 8, so every XCD's L2 sees one contiguous
// eighth of the key range.
template <int DIR>
__global__ void __launch_bounds__(64 * PG_WAVES, PG_OCC / PG_WAVES) k_seed_pg(DIndex ix, DParams P, int pass, SeedBufs sb, uint32_t ccap,
                                                uint32_t* __restrict__ pool, uint32_t pool_words, unsigned long long* __restrict__ ctr, int swz) {
  const uint32_t n_tup = min(sb.sn[SN_TUPLES], sb.cap_tuples), n_fwd = min(sb.sn[SN_FWD], n_tup);
  // this launch's wave chunks of 64 tuples: [c0, c0 + nw) (the chunk that holds the last forward and the first reverse tuple belongs to both)
  const uint32_t c0 = DIR ? n_fwd >> 6 : 0u, nw = (DIR ? (n_tup + 63u) >> 6 : (n_fwd + 63u) >> 6) - c0, per = (nw + 7u) >> 3;
  SMR_DYN_LDS(uint32_t, lds_dyn);
  uint32_t* cdk = lds_dyn + (threadIdx.x >> 6) * PG_LDS_WORDS(ccap);    // rank in the reference's traversal order
  uint32_t* cdv = cdk + ccap;                              // id
  uint32_t* cdn = cdv + ccap;                              // next record of the same search | kind << 16
  uint32_t* hp = cdn + ccap;                               // the hit lists of the 64 searches, back to back
  uint32_t* hd = hp + ccap;                                // [64] newest record of each search
  uint32_t* own = hd + 64;                                 // [64] 1 + the search whose strings start at string g0 + i of the wave (0: none does); after the rows: the chain lengths
  __shared__ uint32_t s_ncand_[PG_WAVES];
  uint32_t& s_ncand = s_ncand_[threadIdx.x >> 6];
  const int lane = lane_id();
  const uint32_t pw = P.partialwin, h = pw / 2;
  const bool full = P.is_full_search != 0;
#ifdef SMR_SEED_PHASES                                    // per-phase cycle accounting (build with -DSMR_SEED_PHASES, run with SMR_DEBUG_PHASES=1)
  unsigned long long tph[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#define GPH(i) { const unsigned long long tn_ = clock64(); tph[i] += tn_ - tlast; tlast = tn_; }
#else
#define GPH(i)
#endif

  // A wave takes the chunks it, it + stride, ... (stride = all waves of the launch): the grid is a few waves per wave slot of the machine, not
  // one block per chunk the batch COULD have (most of which would find nothing to do: the number of tuples is only known on the device).
  // chunk of walk step `it`: 1 = vb is it, 2 = none at this step (the XCD's share is shorter), 0 = the walk is over
  auto chunk_of = [&](uint32_t it, uint32_t& vb) -> int {
    if (swz) { if ((it >> 3) >= per) return 0; vb = (it & 7u) * per + (it >> 3); if (vb >= nw) return 2; }
    else { vb = it; if (vb >= nw) return 0; }
    vb += c0;
    return 1;
  };
  // The next chunk's tuple and its coarse bin are asked for while this chunk's candidates are being sorted out (pf_*): the head of the
  // dependency chain tuple -> bin boundaries -> block table of the next walk step is under way before the step begins
  bool pf = false, pf_bounds = false;
  uint32_t pf_vb = 0, pf_c = 0, pf_nx = 0;
  SeedTup pf_t = 0;
  const uint32_t stride = gridDim.x * (blockDim.x >> 6);
  // the wave's work counters: summed in LDS (not in registers: the kernel sits on its 64-VGPR budget), added to the shard once after its last chunk
  __shared__ unsigned long long s_acc[PG_WAVES][3];
  unsigned long long* acc = s_acc[threadIdx.x >> 6];
  if (lane == 0) { acc[0] = 0; acc[1] = 0; acc[2] = 0; }
  for (uint32_t it = uni(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); ; it += stride) {      // (uni: the wave's number -- the chunk arithmetic belongs on the scalar unit)
  uint32_t vb = 0;                                         // this wave's 64 tuples
  { const int ck = chunk_of(it, vb); if (ck == 0) break; if (ck == 2) continue; }
  __syncthreads();                                         // (the previous chunk's LDS is free)
  // ---- the wave's 64 searches ----
  const uint32_t pos = vb * 64u + lane;
  bool mine = DIR ? (pos >= n_fwd && pos < n_tup) : pos < n_fwd;
  uint32_t nh = 0, P9 = 0, slot = 0, gfl = 0;
  uint2 rt = make_uint2(NONE, 0);
  uint32_t n_rep = 0;                                      // tuples of this launch in the chunk that repeat another one's seed (k_seed_dedup): read, not searched
  {
    const bool hit = pf && pf_vb == vb;
    const SeedTup t = hit ? pf_t : (pos < n_tup ? sb.srt[pos] : 0ull);
    const bool rep = mine && ((t & SEED_TUP_DUP) || !seed_read_active(sb, (uint32_t)t));      // (or the tuple of a read that is not in this launch: one sort for several parts)
    n_rep = (uint32_t)__popcll(__ballot(rep));
    if (rep) mine = false;
    if (!__any(mine)) {                                    // (whole chunks of a hot key: nothing but repeats)
      pf = false; pf_bounds = false;
      if (lane == 0) acc[2] += 8ull * n_rep + 2u;
      continue;
    }
    uint32_t c = hit ? pf_c : (uint32_t)sb.wbin[vb];
    uint32_t nx = (hit && pf_bounds) ? pf_nx : sb.cbase[c + 1];
    pf = false; pf_bounds = false;
    if (mine) {
      while (pos >= nx) { c++; nx = sb.cbase[c + 1]; }          // (pos < cbase[nc]: ends; a chunk of 64 tuples rarely spans more than two bins)
      SeedKey tk;
      tk.slot = (uint32_t)t; tk.chars = (uint32_t)(t >> 32) & ((1u << sb.cb) - 1u); tk.key = (c << sb.fb) | (uint32_t)(t >> (32u + sb.cb));
      rt = ix.root3[2 * (tk.key - (DIR ? sb.nkh : 0u)) + DIR];
      P9 = tk.chars; slot = tk.slot;
      if (DIR == 1) gfl = sb.gflag[slot >> 11];              // (asked for with the block table entry, not after it)
    }
  }
  const bool counted = mine;
  if (mine) {
    if (DIR == 1 && ((gfl >> ((slot >> 6) & 31u)) & 1u) && ((sb.zbits[slot >> 5] >> (slot & 31u)) & 1u)) mine = false;   // accept_zero_kmer: no reverse search (paralleltraversal.cpp:188)
  }
  if (lane == 0) s_ncand = 0;
  __syncthreads();
  GPH(0)

  // ---- the four directory ranges of the search: [0] in TA, [1..3] (S0, S1, S2) in TB ----
  uint32_t rs0 = 0, rs1 = 0, rs2 = 0, rs3 = 0, rn0 = 0, rn1 = 0, rn2 = 0, rn3 = 0;
  uint32_t kA = 0, kb0 = 0, kb1 = 0, cA = 0, cB = 0, n = 0;
  if (mine && rt.x != NONE) {
    n = rt.y & 0xFFFFFFu; cA = (rt.y >> 24) & 15u; cB = rt.y >> 28;
    const uint32_t* blk = ix.pg + (size_t)rt.x * 4;
    if (cA == 0) { rn0 = n; }
    else {
      const uint32_t nA = (1u << (2 * cA)) + 1u, nB = (1u << (2 * cB)) + 1u;
      const uint32_t* dirA = blk; const uint32_t* dirB = blk + nA;
      const uint32_t rev = pg_reversed(P9);
      kA = pg_rkey(rev, 0, cA); kb0 = pg_rkey(rev, h, cB); kb1 = pg_rkey(rev, h - 1, cB);
      uint32_t lo2, hi2;
      if (cB == pw - h) { const uint32_t k2 = pg_rkey(rev, h + 1, cB - 1); lo2 = 4 * k2; hi2 = lo2 + 4; }      // T[pw-1] is free under S2
      else { lo2 = pg_rkey(rev, h + 1, cB); hi2 = lo2 + 1; }
      // (a range = two neighbouring directory words: one 8-byte load each -- five gathers per search instead of eight)
      const PgPair pa = *reinterpret_cast<const PgPair*>(dirA + kA), pb = *reinterpret_cast<const PgPair*>(dirB + kb0), pc = *reinterpret_cast<const PgPair*>(dirB + kb1);
      const uint32_t d0 = dirB[lo2], d1 = dirB[hi2];
      rs0 = pa.lo; rn0 = pa.hi - pa.lo;
      rs1 = pb.lo; rn1 = pb.hi - pb.lo;
      if (kb1 != kb0) { rs2 = pc.lo; rn2 = pc.hi - pc.lo; }
      rs3 = d0; rn3 = d1 - d0;
      (void)nB;
    }
  }
  const uint32_t tot = rn0 + rn1 + rn2 + rn3;
  // wave totals (uniform): directory ranges read (4 per search with directories, 1 without)
  const bool srch = mine && rt.x != NONE;
  const unsigned long long w_node = 4ull * (uint32_t)__popcll(__ballot(srch && cA)) + (uint32_t)__popcll(__ballot(srch && !cA));
  GPH(1)
  // ---- the strings of the wave's 64 searches, 64 at a time whichever search they belong to (a search has 5 strings on average, the
  // busiest of 64 about 13: lane = search would run the wave as long as that one).  String g of the wave belongs to the search s with
  // excl[s] <= g < excl[s] + tot[s]; its lane fetches what it needs of s's state with ds_bpermute ----
  const uint32_t sinc = pg_scan_add(tot);
  const uint32_t excl = sinc - tot, wtot = (uint32_t)__builtin_amdgcn_readlane((int)sinc, 63);
  const unsigned long long w_entry = wtot;
  const uint32_t c1 = rn0, c2 = c1 + rn1, c3 = c2 + rn2;                 // where the ranges S0, S1, S2 begin in the search's own numbering
  const uint32_t u1 = n + rs1, u2 = n + rs2, u3 = n + rs3;               // ... and in the block's strings (TA TB)
  hd[lane] = PG_NIL;
  uint32_t carry = 0;
  // The loop is pipelined by one row and unrolled by two (A and B take turns, nothing is copied): the loads of row i are issued, then the
  // automaton runs over the strings of row i - 1, which were asked for a step earlier -- a wave waits for a string load once, not per row.
  PgOwn O; O.exm = tot ? excl : 0x80000000u; O.P9 = P9;
  {
    const uint32_t* at = pg_strings_at(ix.pg, rt.x == NONE ? 0u : rt.x, rt.y);
    O.blo = (uint32_t)(unsigned long long)at; O.bhi = (uint32_t)((unsigned long long)at >> 32);
    O.mab = ((1u << (2 * cA)) - 1u) | (((1u << (2 * cB)) - 1u) << 16);
  }
  O.t1 = excl + c1; O.t2 = excl + c2; O.t3 = excl + c3;
  O.d0 = rs0 - excl; O.d1 = u1 - O.t1; O.d2 = u2 - O.t2; O.d3 = u3 - O.t3;
  PgRow A, B;
  A.T = A.P = A.m = A.u = A.w = 0; A.s = 0; A.tt = (SMR_GLOBAL_U32*)ix.pg; A.have = false;
  B = A;
  for (uint32_t g0 = 0; g0 < wtot + 64u; g0 += 128) {
    pg_row_fetch(B, g0, wtot, lane, own, O, carry, ix.pg);
    GPH(2)
    pg_row_apply(A, pw, h, full, ccap, &s_ncand, cdk, cdv, cdn, hd);
    GPH(3)
    pg_row_fetch(A, g0 + 64u, wtot, lane, own, O, carry, ix.pg);
    GPH(2)
    pg_row_apply(B, pw, h, full, ccap, &s_ncand, cdk, cdv, cdn, hd);
    GPH(3)
  }
  __syncthreads();
  GPH(2)
  {                                                 // the next walk step's tuple and coarse bin (see pf_* above)
    uint32_t nvb = 0;
    if (chunk_of(it + stride, nvb) == 1) {
      const uint32_t npos = nvb * 64u + lane;
      pf_t = npos < n_tup ? sb.srt[npos] : 0ull;
      pf_c = sb.wbin[nvb];                          // (its bin's upper boundary is asked for once this has arrived: after the selection below)
      pf_vb = nvb; pf = true;
    }
  }
  if (s_ncand > ccap) {                             // hand the wave to k_seed_search
    if (lane == 0) {
      atomicAdd(&ctr[C_SEED_REDO], 1ull);
      const uint32_t p = atomicAdd(&sb.sn[SN_REDO], 1u);
      if (p < sb.cap_redo) sb.redo[p] = vb; else atomicAdd(&ctr[C_ERR_REDO], 1ull);
    }
    continue;
  }
  own[lane] = 0;
  __builtin_amdgcn_wave_barrier();
  pg_resolve(s_ncand, lane, O.blo, O.bhi, rt.y, cdk, cdv, own);
  __syncthreads();
  const uint32_t hb = pg_scan_add(own[lane]) - own[lane];   // where this search's hit list begins: it has at most one entry per candidate
  GPH(6)
  // ---------- every search takes its candidates in DFS order (selection by increasing rank).  Forward: applied to its list -- the window's
  // list so far -- with the reference's rules.  Reverse: collected without repeats, the kind of the first occurrence kept (a later occurrence
  // of an id changes nothing whatever the forward list holds: the id is present by then, or the list was replaced and the search over) ----------
  bool zero = false;
  {
    const uint32_t head = hd[lane];
    uint32_t last = 0;                                     // ranks already applied are < last
    bool more = head != PG_NIL;
    while (__any(more)) {
      uint32_t best = 0xFFFFFFFFu, bi = 0;
      if (more) for (uint32_t q = head; q != PG_NIL; q = cdn[q] & 0xFFFFu) {
        const uint32_t k = cdk[q];
        if (k >= last && k < best) { best = k; bi = q; }
      }
      if (!more || best == 0xFFFFFFFFu) { more = false; continue; }
      const uint32_t idc = cdv[bi], kc = cdn[bi] >> 16;
      bool present = false;
      for (uint32_t f = 0; f < nh; f++) if ((hp[hb + f] & ~SEED_CAND_COND) == idc) { present = true; break; }
      if (DIR == 0 && kc == CK_COND && !present) { hp[hb] = idc; nh = 1; zero = true; more = false; }
      else if (!present) { hp[hb + nh] = idc | ((DIR && kc == CK_COND) ? SEED_CAND_COND : 0u); nh++; }
      last = best + 1;
    }
  }
  GPH(4)
  if (pf) { pf_nx = sb.cbase[pf_c + 1]; pf_bounds = true; }
  // ---- write the windows' hit segments: [count, id x count] ----
  const bool wr = mine && nh > 0;
  const bool inl = wr && nh == 1 && sb.seg_inline;          // one hit: it goes into the window's wseg word, no segment (SEED_SEG_INLINE)
  const uint32_t need = (wr && !inl) ? 1 + nh : 0;
  const uint32_t incl = pg_scan_add(need);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  uint32_t base = 0;
  if (total) {
    if (lane == 0) {
      const uint32_t shard = vb & (C_NSHARD - 1), region = pool_words / C_NSHARD;
      const unsigned long long old = atomicAdd(&ctr[C_PCUR + shard * C_PCUR_STRIDE], (unsigned long long)total);
      if (old + total > region) { atomicAdd(&ctr[C_ERR_POOL], 1ull); base = NONE; } else base = shard * region + (uint32_t)old;
    }
    base = __shfl(base, 0, 64);
  }
  if (inl) wseg_put(sb, DIR, slot, SEED_SEG_INLINE | (hp[hb] & (SEED_SEG_ID | (DIR ? SEED_CAND_COND : 0u))) | (zero ? SEED_ZERO_BIT : 0u), zero);
  else if (wr && base != NONE) {
    const uint32_t o = base + incl - need;
    pool[o] = nh;
    for (uint32_t q = 0; q < nh; q++) pool[o + 1 + q] = hp[hb + q];
    wseg_put(sb, DIR, slot, o | (zero ? SEED_ZERO_BIT : 0u), zero);
  }
  // algorithmic bytes of this wave (C_B_PG0/1): per tuple 8 B + its block-table entry (8 B); a search with directories reads 8 directory
  // words; 4 B per string looked at; {rank, id} = 8 B per accepted string; the segment written (4 B per word) and the window slot pointing
  // to it; the chunk's coarse bin; DIR 1: the window's group bit (what does not come out of the scans above is summed per lane: < 2^32 per wave)
  unsigned long long w_bytes = (uint32_t)__popcll(__ballot(counted)) * ((uint32_t)sizeof(SeedTup) + 8u + (DIR ? 1u : 0u))
                             + 32u * (uint32_t)__popcll(__ballot(srch && cA)) + 4u * (uint32_t)__popcll(__ballot(wr));
  w_bytes += 4ull * wtot + 4ull * total + 8ull * min(s_ncand, ccap) + 2u + 8ull * n_rep;
  if (lane == 0) { acc[0] += w_node; acc[1] += w_entry; acc[2] += w_bytes; }
#ifdef SMR_SEED_PHASES
  GPH(5)
  if (lane == 0) for (int q = 0; q < 7; q++) if (tph[q]) { atomicAdd(&ctr[C_SHARDS + (vb & (C_NSHARD - 1)) * C_SHARD_W + C_SHARD_PH + q], tph[q]); tph[q] = 0; }
#endif
  }
  if (lane == 0) { if (acc[0]) ctr_add(ctr, C_NODE, acc[0]); if (acc[1]) ctr_add(ctr, C_ENTRY, acc[1]); if (acc[2]) ctr_add(ctr, DIR ? C_B_PG1 : C_B_PG0, acc[2]); }
}

}  // namespace smr
