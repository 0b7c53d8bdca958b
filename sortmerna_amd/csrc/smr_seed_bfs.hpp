// smr_seed_bfs.hpp -- the fast seed search (included by smr_kernels.hpp after smr_seed.hpp).
#pragma once

namespace smr {

// ------------------------------------------------------------------------------------------------
// k_seed_bfs<DIR>: the same searches as k_seed_search, organised as wave-wide WORK QUEUES instead of one DFS per lane.
//
// With the closed forms of the LEV(1) automaton (lev1_alive for a trie path, lev1_entry for a complete candidate
// string) a search needs no per-lane automaton state, so the 64 searches of a wave can be taken apart into independent
// work items that any lane can process:
//   node items   (search, node, path)   popped 16 at a time from a LIFO in LDS; lane = (item, element A/C/G/T): one 16-byte
//                                       node is read by 4 adjacent lanes; alive child nodes are pushed back, alive buckets
//                                       go to the bucket queue
//   bucket items (search, bucket, path) taken 64 at a time, ONE LANE PER BUCKET: the index is stored a second time with small
//                                       subtrees collapsed into buckets of <= 32 entries kept BIT-SLICED (smr_host.hpp), and
//                                       lev1_unit evaluates the closed form for all 32 entries of a unit at once with bitwise
//                                       logic (tests/test_lev_closed_form.py::test_bitsliced_unit_equals_closed_form)
//   candidates   (search, order key, id, kind)  accepted entries (rare), collected in a small pool
// The reference's sequential semantics (traverse_bursttrie.cpp:100-298: DFS order A<C<G<T, 0-error match clears the list
// and ends the search, duplicate `break`) are restored at the end: each search applies ITS candidates in DFS order, the
// order key being the bucket's path (most significant char first; a bucket's path is never a prefix of another's) and the
// entry's position in the bucket.  Results are identical to k_seed_search (tests compare both); the work counters are
// not: a search that ends with a 0-error match has visited nodes/entries the reference would have skipped, so exact
// algorithmic-byte counts come from k_seed_search (smr_set_seed_mode).  A wave whose queues overflow hands its 64
// tuples to k_seed_search through the redo list.
// ------------------------------------------------------------------------------------------------
// The LDS queues are small on purpose: occupancy is worth more than the (rare) redo of an overflowing wave
// (12 KB/wave -> 13 waves/CU: 20.7 ms per stage; 8 KB -> 18+ waves/CU: 16.8 ms).
#ifndef BFS_NS_CAP
#define BFS_NS_CAP 256u         // node items
#endif
#define BFS_BQ_CAP 128u         // bucket items
#ifndef BFS_CAND_CAP
#define BFS_CAND_CAP 64u        // candidates
#endif

// dynamic LDS words: pat, root, hit lists, node LIFO (2 words), bucket queue (3 words), candidates (3 words)
#define BFS_LDS_WORDS(hcap) (64u + 64u + 64u * (hcap) + 2u * BFS_NS_CAP + 3u * BFS_BQ_CAP + 3u * BFS_CAND_CAP)

// most-significant-char-first version of a 2-bit packed path (char l at bits 2l) in the top 20 bits of a word
__device__ __forceinline__ uint32_t path_order_key(uint32_t path) {
  const uint32_t rv = __brev(path);
  return (((rv & 0xAAAAAAAAu) >> 1) | ((rv & 0x55555555u) << 1)) >> 12;
}

// The closed form of lev1_entry for the <= 32 candidate strings of one bit-sliced unit at once: bit e of every mask = entry e.
// The unit holds a {lo_j, hi_j} plane pair for every string position j = 0..pw (smr_host.hpp).  With E0/E1/E2_j = entries whose
// char j equals P[j] / P[j-1] / P[j+1], Pr = running AND of E0 from the front, S0/S1/S2 = running ANDs from the back:
//   accepted = OR_k Pr_{k-1} & (S1_{k+1} | S0_{k+1} | S2_k),   0-error = Pr_{pw-1}.
__device__ __forceinline__ void lev1_unit(const uint32_t* __restrict__ unit, uint32_t P, uint32_t pw, uint32_t& acc_out, uint32_t& zero_out) {
  constexpr int NP = SEED_MAXPW + 1;                       // string positions 0..SEED_MAXPW
  uint32_t pl[2 * NP + 2];                                 // the planes: pl[2j] = lo_j, pl[2j+1] = hi_j
  const uint32_t nq = (2 * (pw + 1) + 3) >> 2;             // 16-byte chunks of the plane block
#pragma unroll
  for (int q = 0; q < (2 * NP + 2) / 4; q++) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if ((uint32_t)q < nq) v = *reinterpret_cast<const uint4*>(unit + 4 * q);
    pl[4 * q] = v.x; pl[4 * q + 1] = v.y; pl[4 * q + 2] = v.z; pl[4 * q + 3] = v.w;
  }
  // entries whose char at string position j equals pattern char pj: xnor(lo, b_lo) & xnor(hi, b_hi), b = that bit of P broadcast to a
  // whole word by a 1-bit signed field extract (v_bfe_i32)
  auto eq = [&](int j, int pj) -> uint32_t {
    const uint32_t blo = (uint32_t)__builtin_amdgcn_sbfe((int)P, 2 * pj, 1), bhi = (uint32_t)__builtin_amdgcn_sbfe((int)P, 2 * pj + 1, 1);
    return ~(pl[2 * j] ^ blo) & ~(pl[2 * j + 1] ^ bhi);
  };
  // front to back: Pr[k] = Pr_{k-1}
  uint32_t Pr[NP + 1];
  uint32_t pr = ~0u;
#pragma unroll
  for (int k = 0; k < NP; k++) {
    Pr[k] = pr;
    if ((uint32_t)k < pw) pr &= eq(k, k);
  }
  zero_out = pr;                                           // all pw chars equal
  // back to front: s0 = S0_{j+1}, s1 = S1_{j+1}, s2 = S2_{j+1} when position j is entered
  uint32_t acc = 0, s0 = ~0u, s1 = ~0u, s2 = ~0u;
#pragma unroll
  for (int j = NP - 1; j >= 0; j--) {
    if ((uint32_t)j <= pw) {
      if ((uint32_t)j + 2 <= pw) s2 &= eq(j, j + 1);        // S2_j
      uint32_t t = s1;
      if ((uint32_t)j < pw) t |= s0 | s2;
      acc |= Pr[j] & t;                                     // the term k = j
      if ((uint32_t)j < pw) s0 &= eq(j, j);
      if (j >= 1) s1 &= eq(j, j - 1);
    }
  }
  acc_out = acc;
}

template <int DIR>
__global__ void __launch_bounds__(64, 5) k_seed_bfs(DIndex ix, DParams P, int pass, SeedBufs sb, uint32_t hcap,
                                                 uint32_t* __restrict__ pool, uint32_t pool_words, unsigned long long* __restrict__ ctr) {
  // this phase's tuples: forward bins first, reverse bins after them
  const uint32_t n_all = min(sb.sn[SN_TUPLES], sb.cap_tuples), n_fwd = min(sb.bin_off[sb.nkh], n_all);
  const uint32_t first = DIR ? n_fwd : 0u, n_tup = DIR ? n_all - n_fwd : n_fwd;
  if (blockIdx.x * 64u >= n_tup) return;
  SMR_DYN_LDS(uint32_t, lds_dyn);
  uint32_t* pat = lds_dyn;
  uint32_t* rootw = pat + 64;
  uint32_t* hl = rootw + 64;
  uint32_t* ns0 = hl + 64 * hcap;
  uint32_t* ns1 = ns0 + BFS_NS_CAP;
  uint32_t* bq0 = ns1 + BFS_NS_CAP;
  uint32_t* bq1 = bq0 + BFS_BQ_CAP;
  uint32_t* bq2 = bq1 + BFS_BQ_CAP;
  uint32_t* cd0 = bq2 + BFS_BQ_CAP;
  uint32_t* cd1 = cd0 + BFS_CAND_CAP;
  uint32_t* cd2 = cd1 + BFS_CAND_CAP;
  __shared__ uint32_t s_ncand;
  const int lane = lane_id();
  const uint32_t pw = P.partialwin;
  const bool full = P.is_full_search != 0;
  const uint32_t* __restrict__ arena = ix.trie2;           // the bit-sliced arena
  const unsigned long long lt = (1ull << lane) - 1ull;
#ifdef SMR_SEED_PHASES                                    // per-phase cycle accounting (build with -DSMR_SEED_PHASES, run with SMR_DEBUG_PHASES=1)
  unsigned long long tph[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#define SPH(i) { const unsigned long long tn_ = clock64(); tph[i] += tn_ - tlast; tlast = tn_; }
#define SCN(v) { tph[6] += (v); }
#else
#define SPH(i)
#define SCN(v)
#endif

  // ---- the wave's 64 searches ----
  const uint32_t pos = first + blockIdx.x * 64u + lane;
  bool mine = blockIdx.x * 64u + lane < n_tup;
  uint32_t win_pos = 0, nh = 0, n_prev = 0, root = 0;
  size_t slot = 0;
  bool hl_over = false;
  if (mine) {
    const unsigned long long pl = sb.tup[pos];
    root = ix.root2[2 * (sb.tkey[pos] - (DIR ? sb.nkh : 0u)) + DIR];
    const uint32_t r = (uint32_t)(pl & 0xFFFFFFull);
    win_pos = (uint32_t)((pl >> 24) & 0xFFFFull);
    pat[lane] = (uint32_t)(pl >> 40);
    slot = wseg_slot(sb, r, win_pos / P.skip[pass]);
    if (DIR == 1) {                                      // the window's list so far = the forward search's hits
      const uint32_t seg = sb.wseg[slot];
      if (seg != NONE && (seg & SEED_ZERO_BIT)) mine = false;     // accept_zero_kmer: no reverse search (paralleltraversal.cpp:188)
      else if (seg != NONE) {
        n_prev = pool[seg + 1];
        for (uint32_t q = 0; q < n_prev && q < hcap; q++) hl[q * 64 + lane] = pool[seg + 2 + 2 * q];
        if (n_prev > hcap) { hl_over = true; n_prev = hcap; }
        nh = n_prev;
      }
    }
  }
  rootw[lane] = root;
  // root node items
  uint32_t top = 0, bqn = 0;
  if (lane == 0) s_ncand = 0;
  unsigned long long w_node = 0, w_entry = 0;            // wave totals (uniform)
  bool overflow = false;                                 // a queue overflowed: the wave is redone by k_seed_search
  {
    const unsigned long long mm = __ballot(mine);
    if (mine) { const uint32_t p = (uint32_t)__popcll(mm & lt); ns0[p] = root; ns1[p] = (uint32_t)lane << 24; }
    top = (uint32_t)__popcll(mm);
  }
  __syncthreads();
  SPH(0)

  while ((top > 0 || bqn > 0) && !overflow) {
    if (top > 0 && bqn + 64 <= BFS_BQ_CAP) {
      // ---------- node step: 16 nodes x 4 elements ----------
      const uint32_t cnt = min(16u, top), base = top - cnt;
      const uint32_t it = (uint32_t)lane >> 2, ne = (uint32_t)lane & 3u;
      const bool v = it < cnt;
      uint32_t w0 = 0, w1 = 0, e = 0;
      if (v) { w0 = ns0[base + it]; w1 = ns1[base + it]; e = arena[w0 + ne]; }
      const uint32_t slane = w1 >> 24, depth = (w1 >> 20) & 15u;
      const uint32_t T = (w1 & 0xFFFFFu) | (ne << (2 * depth));
      const uint32_t flag = e >> ELEM_FLAG_SHIFT;
      const bool alive = v && flag != 0 && lev1_alive(pat[slane], T, depth + 1);
      const bool is_c = alive && flag == 1, is_b = alive && flag == 2;
      const unsigned long long cm = __ballot(is_c), bm = __ballot(is_b);
      const uint32_t ncn = (uint32_t)__popcll(cm), nbn = (uint32_t)__popcll(bm);
      w_node += cnt;
      __syncthreads();                                     // all reads of the popped items are done before their slots are reused
      if (base + ncn > BFS_NS_CAP) overflow = true;
      else {
        const uint32_t ro = rootw[slane];
        if (is_c) { const uint32_t p = base + (uint32_t)__popcll(cm & lt); ns0[p] = ro + (e & ELEM_OFF_MASK); ns1[p] = T | ((depth + 1) << 20) | (slane << 24); }
        if (is_b) { const uint32_t p = bqn + (uint32_t)__popcll(bm & lt); bq0[p] = ro + (e & ELEM_OFF_MASK); bq1[p] = T | ((depth + 1) << 20) | (slane << 24); bq2[p] = (e >> ELEM_NENT_SHIFT) & 0xFFu; }
        top = base + ncn; bqn += nbn;
      }
      __syncthreads();
      SPH(1) SCN(1ull)
    } else {
      // ---------- bucket batch: up to 64 buckets, one lane per bucket (its <= 32-entry units bit-sliced) ----------
      const uint32_t nb = min(64u, bqn);
      uint32_t my_n = 0, boff = 0, meta = 0;
      if ((uint32_t)lane < nb) { my_n = bq2[lane]; boff = bq0[lane]; meta = bq1[lane]; }
      const uint32_t slane = meta >> 24, bpath = meta & 0xFFFFFu;
      const uint32_t P9 = pat[slane];
      const uint32_t uw = bs_unit_words(pw), pwords = bs_plane_words(pw);
      {
        uint32_t wsum = my_n;
        for (int d = 32; d > 0; d >>= 1) wsum += __shfl_xor(wsum, d, 64);
        w_entry += wsum;
      }
      for (uint32_t u0 = 0; __any(u0 < my_n); u0 += 32) {       // a bucket has more than one unit only at the deepest level
        uint32_t acc = 0, zr = 0;
        if (u0 < my_n) {
          const uint32_t* unit = arena + boff + (u0 >> 5) * uw;
          lev1_unit(unit, P9, pw, acc, zr);
          const uint32_t c = my_n - u0;
          acc &= c >= 32 ? ~0u : ((1u << c) - 1u);
          while (acc) {
            const uint32_t b = (uint32_t)__builtin_ctz(acc); acc &= acc - 1;
            const uint32_t p = atomicAdd(&s_ncand, 1u);
            if (p < BFS_CAND_CAP) {
              cd0[p] = slane | (((((zr >> b) & 1u) && !full) ? CK_COND : CK_PLAIN) << 8);
              cd1[p] = (path_order_key(bpath) << 8) | (u0 + b);
              cd2[p] = unit[pwords + b];
            }
          }
        }
      }
      __syncthreads();
      SPH(2) SCN(1ull << 32)
      if (s_ncand > BFS_CAND_CAP) overflow = true;
      // drop the processed buckets: move the rest (< 64) to the front
      const uint32_t rest = bqn - nb;
      uint32_t m0 = 0, m1 = 0, m2 = 0;
      if ((uint32_t)lane < rest) { m0 = bq0[nb + lane]; m1 = bq1[nb + lane]; m2 = bq2[nb + lane]; }
      __syncthreads();
      if ((uint32_t)lane < rest) { bq0[lane] = m0; bq1[lane] = m1; bq2[lane] = m2; }
      bqn = rest;
      __syncthreads();
      SPH(3)
    }
  }
  if (overflow) {                                          // hand the wave to k_seed_search
    if (lane == 0) {
      atomicAdd(&ctr[C_SEED_REDO], 1ull);
      const uint32_t p = atomicAdd(&sb.sn[SN_REDO], 1u);
      if (p < sb.cap_redo) sb.redo[p] = blockIdx.x; else atomicAdd(&ctr[C_ERR_REDO], 1ull);
    }
    return;
  }
  // ---------- every search applies its candidates in DFS order (selection by increasing key) ----------
  bool zero = false;
  {
    const uint32_t ncand = min(s_ncand, BFS_CAND_CAP);
    uint32_t last = 0;                                     // keys already applied are < last
    bool more = mine;
    while (__any(more)) {
      uint32_t best = 0xFFFFFFFFu, bi = 0;
      for (uint32_t c = 0; c < ncand; c++) {
        const uint32_t k = cd1[c];
        if ((cd0[c] & 63u) == (uint32_t)lane && k >= last && k < best) { best = k; bi = c; }
      }
      if (!more || best == 0xFFFFFFFFu) { more = false; continue; }
      const uint32_t idc = cd2[bi], kc = cd0[bi] >> 8;
      bool present = false;
      for (uint32_t f = 0; f < nh; f++) if (hl[f * 64 + lane] == idc) { present = true; break; }
      if (kc == CK_COND && !present) { hl[lane] = idc; nh = 1; zero = true; more = false; }
      else if (!present) { if (nh < hcap) { hl[nh * 64 + lane] = idc; nh++; } else hl_over = true; }
      last = best + 1;
    }
  }
  SPH(4)
  // ---- write the windows' hit segments: [unused, count, (id, win_pos) x count] ----
  const bool wr = mine && (DIR == 0 ? nh > 0 : (zero || nh > n_prev));
  const uint32_t need = wr ? 2 + 2 * nh : 0;
  uint32_t incl = need;
  for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
  const uint32_t total = __shfl(incl, 63, 64);
  uint32_t base = 0;
  if (total) {
    if (lane == 0) {
      const uint32_t shard = blockIdx.x & (C_NSHARD - 1), region = pool_words / C_NSHARD;
      const unsigned long long old = atomicAdd(&ctr[C_PCUR + shard], (unsigned long long)total);
      if (old + total > region) { atomicAdd(&ctr[C_ERR_POOL], 1ull); base = NONE; } else base = shard * region + (uint32_t)old;
    }
    base = __shfl(base, 0, 64);
  }
  if (wr && base != NONE) {
    const uint32_t o = base + incl - need;
    pool[o] = NONE; pool[o + 1] = nh;
    for (uint32_t q = 0; q < nh; q++) { pool[o + 2 + 2 * q] = hl[q * 64 + lane]; pool[o + 3 + 2 * q] = win_pos; }
    sb.wseg[slot] = o | (zero ? SEED_ZERO_BIT : 0u);
  }
  if (__any(hl_over) && lane == 0) atomicAdd(&ctr[C_ERR_HITCAP], 1ull);
  if (lane == 0) { if (w_node) ctr_add(ctr, C_NODE, w_node); if (w_entry) ctr_add(ctr, C_ENTRY, w_entry); }
#ifdef SMR_SEED_PHASES
  SPH(5)
  if (lane == 0) for (int q = 0; q < 7; q++) if (tph[q]) atomicAdd(&ctr[C_SHARDS + (blockIdx.x & (C_NSHARD - 1)) * 16 + 9 + q], tph[q]);
#endif
}

}  // namespace smr
