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
//   bucket items (search, bucket, path) taken 64 at a time; their entries are flattened (prefix sum + owner byte map) and
//                                       evaluated one lane per entry
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
#ifndef BFS_OWN_CAP
#define BFS_OWN_CAP 1024u       // entries of one bucket batch with a direct entry -> bucket byte map
#endif
// dynamic LDS words: pat, root, hit lists, node LIFO (2 words), bucket queue (3 words), pref, candidates (3 words), owner map
#define BFS_LDS_WORDS(hcap) (64u + 64u + 64u * (hcap) + 2u * BFS_NS_CAP + 3u * BFS_BQ_CAP + 64u + 3u * BFS_CAND_CAP + BFS_OWN_CAP / 4u)

// most-significant-char-first version of a 2-bit packed path (char l at bits 2l) in the top 20 bits of a word
__device__ __forceinline__ uint32_t path_order_key(uint32_t path) {
  const uint32_t rv = __brev(path);
  return (((rv & 0xAAAAAAAAu) >> 1) | ((rv & 0x55555555u) << 1)) >> 12;
}

template <int DIR>
__global__ void __launch_bounds__(64) k_seed_bfs(DIndex ix, DParams P, int pass, SeedBufs sb, uint32_t hcap,
                                                 uint32_t* __restrict__ pool, uint32_t pool_words, unsigned long long* __restrict__ ctr) {
  // this phase's tuples: forward bins first, reverse bins after them
  const uint32_t n_all = min(sb.sn[SN_TUPLES], sb.cap_tuples), n_fwd = min(sb.bin_off[sb.nkh], n_all);
  const uint32_t first = DIR ? n_fwd : 0u, n_tup = DIR ? n_all - n_fwd : n_fwd;
  if (blockIdx.x * 64u >= n_tup) return;
  extern __shared__ __align__(16) uint32_t lds_dyn[];
  uint32_t* pat = lds_dyn;
  uint32_t* rootw = pat + 64;
  uint32_t* hl = rootw + 64;
  uint32_t* ns0 = hl + 64 * hcap;
  uint32_t* ns1 = ns0 + BFS_NS_CAP;
  uint32_t* bq0 = ns1 + BFS_NS_CAP;
  uint32_t* bq1 = bq0 + BFS_BQ_CAP;
  uint32_t* bq2 = bq1 + BFS_BQ_CAP;
  uint32_t* pref = bq2 + BFS_BQ_CAP;
  uint32_t* cd0 = pref + 64;
  uint32_t* cd1 = cd0 + BFS_CAND_CAP;
  uint32_t* cd2 = cd1 + BFS_CAND_CAP;
  uint8_t* own = reinterpret_cast<uint8_t*>(cd2 + BFS_CAND_CAP);
  const int lane = lane_id();
  const uint32_t pw = P.partialwin;
  const bool full = P.is_full_search != 0;
  const uint32_t* __restrict__ arena = ix.trie;
  const unsigned long long lt = (1ull << lane) - 1ull;

  // ---- the wave's 64 searches ----
  const uint32_t pos = first + blockIdx.x * 64u + lane;
  bool mine = blockIdx.x * 64u + lane < n_tup;
  uint32_t win_pos = 0, nh = 0, n_prev = 0, root = 0;
  size_t slot = 0;
  bool hl_over = false;
  if (mine) {
    const unsigned long long pl = sb.tup[pos];
    const Lookup lk = ix.lookup[sb.tkey[pos] - (DIR ? sb.nkh : 0u)];
    root = DIR == 0 ? lk.rootF : lk.rootR;
    const uint32_t r = (uint32_t)(pl & 0xFFFFFFull);
    win_pos = (uint32_t)((pl >> 24) & 0xFFFFull);
    pat[lane] = (uint32_t)(pl >> 40);
    slot = (size_t)r * sb.maxwin + win_pos / P.skip[pass];
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
  uint32_t top = 0, bqn = 0, ncand = 0;
  unsigned long long w_node = 0, w_entry = 0;            // wave totals (uniform)
  bool overflow = false;                                 // a queue overflowed: the wave is redone by k_seed_search
  {
    const unsigned long long mm = __ballot(mine);
    if (mine) { const uint32_t p = (uint32_t)__popcll(mm & lt); ns0[p] = root; ns1[p] = (uint32_t)lane << 24; }
    top = (uint32_t)__popcll(mm);
  }
  __syncthreads();

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
    } else {
      // ---------- bucket batch: up to 64 buckets, one lane per entry ----------
      const uint32_t nb = min(64u, bqn);
      const uint32_t my_n = (uint32_t)lane < nb ? bq2[lane] : 0u;
      uint32_t incl = my_n;
      for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
      const uint32_t Tn = __shfl(incl, 63, 64);
      const uint32_t my_first = incl - my_n;
      pref[lane] = my_first;
      const bool direct = Tn <= BFS_OWN_CAP;
      if (direct) for (uint32_t q = 0; q < my_n; q++) own[my_first + q] = (uint8_t)lane;
      w_entry += Tn;
      __syncthreads();
      for (uint32_t eb = 0; eb < Tn; eb += 64) {
        const uint32_t e = eb + lane;
        const bool v = e < Tn;
        uint32_t bk = 0;
        if (v) {
          if (direct) bk = own[e];
          else for (uint32_t step = 32; step > 0; step >>= 1) { const uint32_t t = bk + step; if (t < 64 && pref[t] <= e) bk = t; }
        }
        const uint32_t q = e - pref[bk];
        const uint32_t meta = bq1[bk];
        uint32_t str = 0, id = 0;
        if (v) { const uint2 en = *reinterpret_cast<const uint2*>(arena + bq0[bk] + 2 * q); str = en.x; id = en.y; }
        const uint32_t nchar = (meta >> 20) & 15u, slane = meta >> 24;
        const uint32_t tstr = (meta & 0xFFFFFu) | (str << (2 * nchar));
        const uint32_t r = v ? lev1_entry(pat[slane], tstr, pw) : 0u;
        const bool acc = (r & 1u) != 0;
        const unsigned long long am = __ballot(acc);
        if (am) {
          const uint32_t p = ncand + (uint32_t)__popcll(am & lt);
          if (acc && p < BFS_CAND_CAP) {
            cd0[p] = slane | ((((r & 2u) && !full) ? CK_COND : CK_PLAIN) << 8);
            cd1[p] = (path_order_key(meta & 0xFFFFFu) << 8) | q;
            cd2[p] = id;
          }
          ncand += (uint32_t)__popcll(am);
          if (ncand > BFS_CAND_CAP) overflow = true;
        }
      }
      __syncthreads();
      // drop the processed buckets: move the rest (< 64) to the front
      const uint32_t rest = bqn - nb;
      uint32_t m0 = 0, m1 = 0, m2 = 0;
      if ((uint32_t)lane < rest) { m0 = bq0[nb + lane]; m1 = bq1[nb + lane]; m2 = bq2[nb + lane]; }
      __syncthreads();
      if ((uint32_t)lane < rest) { bq0[lane] = m0; bq1[lane] = m1; bq2[lane] = m2; }
      bqn = rest;
      __syncthreads();
    }
  }
  if (overflow) {                                          // hand the wave to k_seed_search
    if (lane == 0) {
      const uint32_t p = atomicAdd(&sb.sn[SN_REDO], 1u);
      if (p < sb.cap_redo) sb.redo[p] = blockIdx.x; else atomicAdd(&ctr[C_ERR_REDO], 1ull);
    }
    return;
  }
  // ---------- every search applies its candidates in DFS order (selection by increasing key) ----------
  bool zero = false;
  {
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
}

}  // namespace smr
