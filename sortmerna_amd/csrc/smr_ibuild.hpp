// smr_ibuild.hpp -- SURVEY.md 8(f) N3: the index of one part built on the device.
//
// Replaces the per-occurrence work of the reference's build_index (indexdb.cpp:1119-2095: CMPH over all 18-mers, one
// trie insertion per 19-mer, 10 s per 3.6 Mnt) with sorting:
//   1. k_ib_keys      every (L+1)-mer window o of the part -> key = (L-mer code << occbits) | o, and its last letter
//   2. radix sort     LSD over the code bits only (the keys start out ordered by o, the sort is stable, so equal L-mers stay in
//                     scan order -- the order the reference's max_pos truncation depends on, indexdb.cpp:318-349)
//   3. k_ib_flags .. k_ib_positions   group heads -> id = rank of the unique L-mer, positions CSR (first max_pos occurrences)
//   4. k_ib_entries   per id and present last letter: the forward entry (first L/2 nt -> key, rest = tail) and the reverse entry
//                     (last L/2 nt -> key, reversed head = tail); forward entries come out sorted, reverse entries are radix-sorted
//   5. k_ib_sizes / k_ib_emit   one thread per (9-mer key, direction): minitrie_layout (smr_trie_layout.hpp, the function the host
//                     builder uses) first for the sizes, then, after a scan, for the words -- so both builders give the same arena.
// The result is copied into the same smr_index the host builder fills: identical index files (tests compare them byte for byte).
#pragma once
#include "smr_trie_layout.hpp"

namespace smr {

typedef uint64_t u64;

// ---- exclusive scan: tiles of 2048 (256 threads x 8), tile sums scanned recursively by the host driver ----------------------
template <class T>
__global__ void __launch_bounds__(256) k_scan_tile(const T* __restrict__ in, T* __restrict__ out, T* __restrict__ tile_sums, u64 n) {
  __shared__ T s_w[4];
  const u64 base = (u64)blockIdx.x * 2048 + (u64)threadIdx.x * 8;
  T v[8], sum = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { v[k] = base + k < n ? in[base + k] : (T)0; sum += v[k]; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T incl = sum;
  for (int d = 1; d < 64; d <<= 1) { const T t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  T pre = incl - sum;
  for (int w = 0; w < wave; w++) pre += s_w[w];
#pragma unroll
  for (int k = 0; k < 8; k++) { if (base + k < n) out[base + k] = pre; pre += v[k]; }
  if (threadIdx.x == 255) tile_sums[blockIdx.x] = pre;
}
template <class T>
__global__ void __launch_bounds__(256) k_scan_add(T* __restrict__ out, const T* __restrict__ tile_prefix, u64 n) {
  const u64 base = (u64)blockIdx.x * 2048 + (u64)threadIdx.x * 8;
  const T add = tile_prefix[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 8; k++) if (base + k < n) out[base + k] += add;
}

// ---- stable LSD radix sort, 8 bits per pass, u64 keys with an optional u32 payload; one wave sorts a tile of 4096 ---------------
#define RS_TILE 4096
__global__ void __launch_bounds__(64) k_rs_hist(const u64* __restrict__ keys, u64 n, int shift, uint32_t* __restrict__ hist, uint32_t n_tiles) {
  __shared__ uint32_t s_h[256];
  const int lane = threadIdx.x;
  for (int d = lane; d < 256; d += 64) s_h[d] = 0;
  __syncthreads();
  const u64 t0 = (u64)blockIdx.x * RS_TILE;
  for (int r = 0; r < RS_TILE / 64; r++) {
    const u64 i = t0 + (u64)r * 64 + lane;
    if (i < n) atomicAdd(&s_h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  for (int d = lane; d < 256; d += 64) hist[(size_t)d * n_tiles + blockIdx.x] = s_h[d];     // digit-major: the scan gives global offsets
}
__global__ void __launch_bounds__(64) k_rs_scatter(const u64* __restrict__ kin, const uint32_t* __restrict__ vin, u64* __restrict__ kout, uint32_t* __restrict__ vout,
                                                   u64 n, int shift, const uint32_t* __restrict__ offs, uint32_t n_tiles) {
  __shared__ uint32_t s_base[257];
  const int lane = threadIdx.x;
  for (int d = lane; d < 256; d += 64) s_base[d] = offs[(size_t)d * n_tiles + blockIdx.x];
  __syncthreads();
  const u64 t0 = (u64)blockIdx.x * RS_TILE;
  const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int r = 0; r < RS_TILE / 64; r++) {
    const u64 i = t0 + (u64)r * 64 + lane;
    const bool valid = i < n;
    const u64 key = valid ? kin[i] : 0ull;
    const uint32_t d = valid ? (uint32_t)(key >> shift) & 255u : 256u;
    u64 same = __ballot(valid);                       // lanes of this round with my digit, in tile order
#pragma unroll
    for (int b = 0; b < 8; b++) { const u64 bb = __ballot((d >> b) & 1u); same &= ((d >> b) & 1u) ? bb : ~bb; }
    const uint32_t rank = (uint32_t)__popcll(same & lt), cnt = (uint32_t)__popcll(same);
    const uint32_t base = s_base[d];
    __syncthreads();
    if (valid && rank == 0) s_base[d] = base + cnt;
    __syncthreads();
    if (valid) { kout[base + rank] = key; if (vin) vout[base + rank] = vin[i]; }
  }
}

// ---- the build ------------------------------------------------------------------------------------------------------------------
struct IBuildDev {
  const uint8_t* codes; const u64* seq_off; const u64* occ_start; uint32_t n_seqs;
  uint32_t L, P, W, T, occbits, max_pos; u64 N;
};

__device__ __forceinline__ uint32_t ib_member_of(const u64* occ_start, uint32_t n_seqs, u64 occ) {      // last m with occ_start[m] <= occ
  uint32_t lo = 0, hi = n_seqs;
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (occ_start[mid] <= occ) lo = mid; else hi = mid; }
  return lo;
}

__global__ void __launch_bounds__(256) k_ib_keys(IBuildDev B, u64* __restrict__ keys, uint8_t* __restrict__ last) {
  const u64 o = (u64)blockIdx.x * 256 + threadIdx.x;
  if (o >= B.N) return;
  const uint32_t m = ib_member_of(B.occ_start, B.n_seqs, o);
  const uint8_t* s = B.codes + B.seq_off[m] + (o - B.occ_start[m]);
  u64 code = 0;
  for (uint32_t k = 0; k < B.L; k++) code = (code << 2) | s[k];
  keys[o] = (code << B.occbits) | o;
  last[o] = s[B.L];
}
__global__ void __launch_bounds__(256) k_ib_flags(const u64* __restrict__ keys, u64 n, uint32_t occbits, uint32_t* __restrict__ flag) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  flag[i] = (i == 0 || (keys[i] >> occbits) != (keys[i - 1] >> occbits)) ? 1u : 0u;
}
// per sorted element: group start of its id; per element: the present-letter mask of its id
__global__ void __launch_bounds__(256) k_ib_groups(const u64* __restrict__ keys, const uint8_t* __restrict__ last, const uint32_t* __restrict__ flag,
                                                    const uint32_t* __restrict__ excl, u64 n, uint32_t occbits, uint32_t* __restrict__ gstart, uint32_t* __restrict__ present) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t id = excl[i] + flag[i] - 1;
  if (flag[i]) gstart[id] = (uint32_t)i;
  const u64 occ = keys[i] & ((1ull << occbits) - 1);
  atomicOr(&present[id], 1u << last[occ]);
}
__global__ void __launch_bounds__(256) k_ib_counts(const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ present, uint32_t n_ids, uint32_t max_pos,
                                                    uint32_t* __restrict__ pcount, uint32_t* __restrict__ ecount) {
  const uint32_t id = blockIdx.x * 256 + threadIdx.x;
  if (id >= n_ids) return;
  const uint32_t cnt = gstart[id + 1] - gstart[id];
  pcount[id] = max_pos == 0 ? cnt : min(cnt, max_pos);              // the first max_pos occurrences in scan order (indexdb.cpp:318-349)
  ecount[id] = (uint32_t)__popc(present[id]);
}
__global__ void __launch_bounds__(256) k_ib_positions(IBuildDev B, const u64* __restrict__ keys, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ excl,
                                                       const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ pcount, const uint32_t* __restrict__ pos_off,
                                                       uint32_t* __restrict__ pos_arr) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= B.N) return;
  const uint32_t id = excl[i] + flag[i] - 1;
  const uint32_t r = (uint32_t)i - gstart[id];
  if (r >= pcount[id]) return;
  const u64 occ = keys[i] & ((1ull << B.occbits) - 1);
  const uint32_t m = ib_member_of(B.occ_start, B.n_seqs, occ);
  const size_t w = 2 * ((size_t)pos_off[id] + r);
  pos_arr[w] = (uint32_t)(occ - B.occ_start[m]);
  pos_arr[w + 1] = m;
}
// forward entries in their final order; reverse entries as (sort key, id)
__global__ void __launch_bounds__(256) k_ib_entries(IBuildDev B, const u64* __restrict__ keys, const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ present,
                                                     const uint32_t* __restrict__ ent_off, uint32_t n_ids, uint32_t* __restrict__ f_key, u64* __restrict__ f_tail_id,
                                                     u64* __restrict__ r_sortkey, uint32_t* __restrict__ r_id) {
  const uint32_t id = blockIdx.x * 256 + threadIdx.x;
  if (id >= n_ids) return;
  const u64 pre = keys[gstart[id]] >> B.occbits;
  uint32_t j = ent_off[id];
  for (uint32_t c = 0; c < 4; c++) {
    if (!((present[id] >> c) & 1u)) continue;
    const u64 code = (pre << 2) | c;                                    // the (L+1)-mer, first nt in the most significant position
    f_key[j] = (uint32_t)(code >> (2 * B.T));
    f_tail_id[j] = ((code & ((1ull << (2 * B.T)) - 1)) << 32) | id;
    const u64 keyR = code & ((1ull << (2 * B.P)) - 1), head = code >> (2 * B.P);
    u64 tailR = 0;                                                      // reversed head (indexdb.cpp:1441-1444)
    for (uint32_t k = 0; k < B.T; k++) tailR = (tailR << 2) | ((head >> (2 * k)) & 3ull);
    r_sortkey[j] = (keyR << (2 * B.T)) | tailR;
    r_id[j] = id;
    j++;
  }
}
__global__ void __launch_bounds__(256) k_ib_rfinal(const u64* __restrict__ r_sortkey, const uint32_t* __restrict__ r_id, uint32_t M, uint32_t T2,
                                                    u64* __restrict__ r_tail_id, uint32_t* __restrict__ cntR, const uint32_t* __restrict__ f_key, uint32_t* __restrict__ cntF) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  const u64 k = r_sortkey[j];
  r_tail_id[j] = ((k & ((1ull << T2) - 1)) << 32) | r_id[j];
  atomicAdd(&cntR[(uint32_t)(k >> T2)], 1u);
  atomicAdd(&cntF[f_key[j]], 1u);
}
// one thread per (key, direction): t = 2 * key + direction
__global__ void __launch_bounds__(256) k_ib_sizes(const u64* __restrict__ f_tail_id, const u64* __restrict__ r_tail_id, const uint32_t* __restrict__ fstart,
                                                   const uint32_t* __restrict__ rstart, uint32_t NK, int T, int burst_depth, u64* __restrict__ size,
                                                   uint32_t* __restrict__ nodes, uint32_t* __restrict__ buckets, uint32_t* __restrict__ status) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 2 * NK) return;
  const uint32_t k = t >> 1, dir = t & 1u;
  const uint32_t lo = dir ? rstart[k] : fstart[k], hi = dir ? rstart[k + 1] : fstart[k + 1];
  size[t] = 0; nodes[t] = 0; buckets[t] = 0;
  if (hi == lo) return;
  int st = TRIE_OK;
  size[t] = minitrie_layout<false>((dir ? r_tail_id : f_tail_id) + lo, hi - lo, T, burst_depth, nullptr, &nodes[t], &buckets[t], &st);
  if (st != TRIE_OK) atomicMax(status, (uint32_t)st);
}
__global__ void __launch_bounds__(256) k_ib_emit(const u64* __restrict__ f_tail_id, const u64* __restrict__ r_tail_id, const uint32_t* __restrict__ fstart,
                                                  const uint32_t* __restrict__ rstart, uint32_t NK, int T, int burst_depth, const u64* __restrict__ toff,
                                                  uint32_t* __restrict__ trie, Lookup* __restrict__ lookup) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 2 * NK) return;
  const uint32_t k = t >> 1, dir = t & 1u;
  const uint32_t lo = dir ? rstart[k] : fstart[k], hi = dir ? rstart[k + 1] : fstart[k + 1];
  uint32_t root = NONE, words = 0;
  if (hi > lo) {
    int st = TRIE_OK;
    root = (uint32_t)toff[t];
    words = minitrie_layout<true>((dir ? r_tail_id : f_tail_id) + lo, hi - lo, T, burst_depth, trie + root, nullptr, nullptr, &st);
  }
  if (dir == 0) { lookup[k].rootF = root; lookup[k].wordsF = words; lookup[k].count = (fstart[k + 1] - fstart[k]) + (rstart[k + 1] - rstart[k]); }
  else { lookup[k].rootR = root; lookup[k].wordsR = words; }
}

}  // namespace smr
