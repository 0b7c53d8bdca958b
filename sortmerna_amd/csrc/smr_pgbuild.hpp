// smr_pgbuild.hpp -- the pigeonhole layout of an index part (smr_host.hpp: per mini-trie the complete candidate strings in two orders,
// their {DFS rank, id} pairs and two directories) built ON THE DEVICE from the reference-shaped arena that is uploaded anyway.
// Replaces the host transform smr_build_pigeonhole (kept as the checker of this one, smr_index_selfcheck / smr_index_check_device): no host
// pass over the tries and no 24 bytes per entry over PCIe -- 0.7 s + 0.25 s of the 2.2 s an index part of 190 M entries took to get ready.
//
//   k_pgb_sizes     thread per (key, direction): entries of its mini-trie (iterative DFS), words of its block
//   (scans)         entry offsets, block offsets
//   k_pgb_collect   thread per mini-trie: the block table entry; its entries in DFS order -> estr / eid / eblk; blocks without directories
//                   (n <= PG_SCAN) are written here
//   k_pgb_keys<O>   thread per entry: sort key = block << bits | string key of order O (A: the whole string, B: chars h..pw-1), value = entry
//   (stable LSD radix sort of keys / values: blocks stay grouped, equal keys stay in DFS order)
//   k_pgb_emit<O>   thread per sorted entry: string and {rank, id} to their places, directory slots up to its key, tail of the directory
//                   and the block's padding by its last entry
#pragma once
#include "smr_host.hpp"

namespace smr {

__host__ __device__ inline uint64_t pgb_block_words(uint32_t n, uint32_t pw) {
  uint32_t cA, cB;
  pg_chars(n, pw, cA, cB);
  const uint64_t w = cA ? (uint64_t)(1u << (2 * cA)) + 1 + (1u << (2 * cB)) + 1 + 6 * (uint64_t)n : 3 * (uint64_t)n;
  return (w + 3) & ~(uint64_t)3;
}

// the reference's DFS over a mini-trie (A<C<G<T, bucket order): f(path chars, path length, bucket words, entries) per bucket
template <class F> __device__ __forceinline__ void pgb_walk(const uint32_t* __restrict__ t, F f) {
  uint32_t s_node[12], s_pre[12], s_ne[12];
  int sp = 0;
  uint32_t node = 0, pre = 0, plen = 0, ne = 0;
  for (;;) {
    if (ne == 4) {
      if (sp == 0) return;
      sp--; node = s_node[sp]; pre = s_pre[sp]; ne = s_ne[sp]; plen--;
      continue;
    }
    const uint32_t e = t[node + ne], fl = e >> ELEM_FLAG_SHIFT;
    const uint32_t p2 = pre | (ne << (2 * plen));
    if (fl == 2) { f(p2, plen + 1, t + (e & ELEM_OFF_MASK), (e >> ELEM_NENT_SHIFT) & 0xFFu); ne++; }
    else if (fl == 1 && sp < 12) { s_node[sp] = node; s_pre[sp] = pre; s_ne[sp] = ne + 1; sp++; node = e & ELEM_OFF_MASK; pre = p2; plen++; ne = 0; }
    else ne++;
  }
}

__global__ void __launch_bounds__(256) k_pgb_sizes(const Lookup* __restrict__ lookup, const uint32_t* __restrict__ trie, uint32_t nk, uint32_t pw,
                                                  uint32_t* __restrict__ cnt, u64* __restrict__ words) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * nk) return;
  const Lookup lk = lookup[i >> 1];
  const uint32_t root = (i & 1) ? lk.rootR : lk.rootF;
  uint32_t n = 0;
  if (root != NONE) pgb_walk(trie + root, [&](uint32_t, uint32_t, const uint32_t*, uint32_t ne) { n += ne; });
  cnt[i] = n;
  words[i] = root != NONE ? pgb_block_words(n, pw) : 0;
}

__global__ void __launch_bounds__(256) k_pgb_collect(const Lookup* __restrict__ lookup, const uint32_t* __restrict__ trie, uint32_t nk, uint32_t pw,
                                                    const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ eoff, const u64* __restrict__ woff,
                                                    uint32_t* __restrict__ root3, uint32_t* __restrict__ pg,
                                                    uint32_t* __restrict__ estr, uint32_t* __restrict__ eid, uint32_t* __restrict__ eblk, uint32_t* __restrict__ err,
                                                    const uint32_t* __restrict__ pos_off) {
  // (pos_off: the ids the searches hand on are where a seed's position list lies -- id' = pos_off[id] + id, the list behind a header word: smr_engine.hip k_pos2_build)
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * nk) return;
  const Lookup lk = lookup[i >> 1];
  const uint32_t root = (i & 1) ? lk.rootR : lk.rootF;
  if (root == NONE) { root3[2 * i] = NONE; root3[2 * i + 1] = 0; return; }
  const uint32_t n = cnt[i];
  if (n > 0xFFFFFFu) { atomicOr(err, 1u); root3[2 * i] = NONE; root3[2 * i + 1] = 0; return; }     // "a mini-trie is too large for the pigeonhole layout"
  if ((woff[i] >> 2) > 0xFFFFFFF0ull) { atomicOr(err, 2u); root3[2 * i] = NONE; root3[2 * i + 1] = 0; return; }   // "arena exceeds 2^34 words"
  uint32_t cA, cB;
  pg_chars(n, pw, cA, cB);
  root3[2 * i] = (uint32_t)(woff[i] >> 2);
  root3[2 * i + 1] = n | (cA << 24) | (cB << 28);
  uint32_t* blk = pg + woff[i];
  uint32_t r = 0;
  const uint32_t e0 = eoff[i];
  pgb_walk(trie + root, [&](uint32_t path, uint32_t plen, const uint32_t* b, uint32_t ne) {
    for (uint32_t q = 0; q < ne; q++, r++) {
      const uint32_t str = path | (b[2 * q] << (2 * plen)), id0 = b[2 * q + 1], id = pos_off[id0] + id0;
      estr[e0 + r] = str; eid[e0 + r] = id; eblk[e0 + r] = i;
      if (cA == 0) { blk[r] = str; blk[n + 2 * r] = r; blk[n + 2 * r + 1] = id; }      // a block without directories: one array in DFS order
    }
  });
  if (cA == 0) for (u64 q = 3 * (u64)n; q < pgb_block_words(n, pw); q++) blk[q] = 0;
}

// ORDER 0 (A): the whole string, first char most significant; ORDER 1 (B): chars h..pw-1
template <int ORDER>
__global__ void __launch_bounds__(256) k_pgb_keys(const uint32_t* __restrict__ estr, const uint32_t* __restrict__ eblk, u64 n_ent, uint32_t pw, uint32_t kbits,
                                                 u64* __restrict__ keys, uint32_t* __restrict__ vals) {
  const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ent) return;
  const uint32_t h = pw / 2;
  const uint32_t k = ORDER == 0 ? pg_key(estr[e], 0, pw + 1) : pg_key(estr[e], h, pw - h);
  keys[e] = ((u64)eblk[e] << kbits) | k;
  vals[e] = (uint32_t)e;
}

template <int ORDER>
__global__ void __launch_bounds__(256) k_pgb_emit(const u64* __restrict__ keys, const uint32_t* __restrict__ vals, u64 n_ent, uint32_t pw, uint32_t kbits,
                                                 const uint32_t* __restrict__ estr, const uint32_t* __restrict__ eid, const uint32_t* __restrict__ eoff,
                                                 const uint32_t* __restrict__ root3, uint32_t* __restrict__ pg) {
  const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_ent) return;
  const u64 key = keys[p];
  const uint32_t i = (uint32_t)(key >> kbits), e = vals[p];
  const uint32_t meta = root3[2 * i + 1], n = meta & 0xFFFFFFu, cA = (meta >> 24) & 15u, cB = meta >> 28;
  if (cA == 0) return;                                   // written by k_pgb_collect
  const uint32_t j = (uint32_t)(p - eoff[i]), rank = e - eoff[i];
  const uint32_t h = pw / 2;
  const uint32_t nA = (1u << (2 * cA)) + 1, nB = (1u << (2 * cB)) + 1;
  uint32_t* blk = pg + (u64)root3[2 * i] * 4;
  uint32_t* TT = blk + nA + nB + (ORDER ? n : 0);
  uint32_t* RR = blk + nA + nB + 2 * (u64)n + (ORDER ? 2 * (u64)n : 0);
  TT[j] = estr[e];
  RR[2 * (u64)j] = rank; RR[2 * (u64)j + 1] = eid[e];
  // directory: slot q = the first string of the block whose leading c chars are >= q
  uint32_t* dir = ORDER ? blk + nA : blk;
  const uint32_t nd = ORDER ? nB : nA;
  const uint32_t kmask = (uint32_t)((1ull << kbits) - 1ull);
  const uint32_t sh = ORDER ? 2 * ((pw - h) - cB) : 2 * ((pw + 1) - cA);
  const uint32_t kk = ((uint32_t)key & kmask) >> sh;
  const uint32_t first = j == 0 ? 0u : (((uint32_t)keys[p - 1] & kmask) >> sh) + 1u;
  for (uint32_t q = first; q <= kk; q++) dir[q] = j;
  if (j == n - 1) {
    for (uint32_t q = kk + 1; q < nd; q++) dir[q] = n;
    if (ORDER == 0) for (u64 q = (u64)nA + nB + 6 * (u64)n; q < pgb_block_words(n, pw); q++) blk[q] = 0;
  }
}

}  // namespace smr
