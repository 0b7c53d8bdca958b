// mb.hip -- measurement aids for GPU sessions (not part of libsmr_hip): hipcc --offload-arch=gfx950 -O3 -o build/mb mb.hip
//
//   mb calib [arena_GB]     known-byte access patterns for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
//                           (MI355X_MICROARCH.md: only wide coalesced streaming reads are calibrated; "calibrate on a known byte count in
//                           your own access pattern").  Every pattern is its own kernel (c_*), launched once after one warm-up launch of
//                           the same kernel on OTHER data, prints its algorithmic byte count and HIP-event time; run it again under
//                           `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` and divide (tools/pmc_calib.py).
//   mb sort [Mtuples]       variants of the seed stage's tuple sort (12-byte tuples, 19-bit keys) on synthetic uniform keys: time per
//                           variant, result checked (sorted by key, payload checksum).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

static hipEvent_t ev0, ev1;
template <class F> static float timed(F f) {
  CK(hipEventRecord(ev0, 0)); f(); CK(hipEventRecord(ev1, 0)); CK(hipEventSynchronize(ev1));
  float ms = 0; CK(hipEventElapsedTime(&ms, ev0, ev1)); return ms;
}
__host__ __device__ static inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__host__ __device__ static inline unsigned long long mix64(unsigned long long x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// =====================================================================================================================
// calibration patterns
// =====================================================================================================================
// P1: streaming read, 16 B per lane, grid-stride
__global__ void __launch_bounds__(256) c_stream_read16(const uint4* __restrict__ a, size_t n16, unsigned long long* sink) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = a[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x123456789abcull) *sink = acc;
}
// P2: streaming read, 4 B per lane
__global__ void __launch_bounds__(256) c_stream_read4(const uint32_t* __restrict__ a, size_t n4, unsigned long long* sink) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += a[i];
  if (acc == 0x123456789abcull) *sink = acc;
}
// P2b: streaming read of 12-byte records, one record per lane (the tuple reads of the sort)
struct T12 { uint32_t a, b, c; };
__global__ void __launch_bounds__(256) c_stream_read12(const T12* __restrict__ a, size_t n, unsigned long long* sink) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const T12 v = a[i]; acc += v.a ^ v.b ^ v.c; }
  if (acc == 0x123456789abcull) *sink = acc;
}
// P3: independent random gathers of W bytes (W = 4, 8, 16), `per` per thread, addresses = hash(thread, j) over the whole arena
template <int W> __global__ void __launch_bounds__(256) c_gather(const uint32_t* __restrict__ a, size_t n_words, uint32_t per, uint32_t salt, unsigned long long* sink) {
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long acc = 0;
  const size_t slots = n_words / (W / 4);
  for (uint32_t j = 0; j < per; j++) {
    const size_t s = (size_t)(mix64(t * 1315423911ull + j * 2654435761ull + salt) % slots) * (W / 4);
    if (W == 4) acc += a[s];
    else if (W == 8) { const uint2 v = *reinterpret_cast<const uint2*>(a + s); acc += v.x ^ v.y; }
    else { const uint4 v = *reinterpret_cast<const uint4*>(a + s); acc += v.x ^ v.y ^ v.z ^ v.w; }
  }
  if (acc == 0x123456789abcull) *sink = acc;
}
// P4: DEPENDENT random 4-byte gathers (pointer chase through hashed addresses): the pattern of a directory walk
__global__ void __launch_bounds__(64) c_chase4(const uint32_t* __restrict__ a, size_t n_words, uint32_t steps, uint32_t salt, unsigned long long* sink) {
  const unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long x = t * 0x9E3779B97F4A7C15ull + salt;
  unsigned long long acc = 0;
  for (uint32_t j = 0; j < steps; j++) { const uint32_t v = a[(size_t)(mix64(x) % n_words)]; acc += v; x = x * 6364136223846793005ull + v + 1; }
  if (acc == 0x123456789abcull) *sink = acc;
}
// P5: streaming write, 16 B / 4 B per lane
__global__ void __launch_bounds__(256) c_stream_write16(uint4* __restrict__ a, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) a[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ void __launch_bounds__(256) c_stream_write4(uint32_t* __restrict__ a, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) a[i] = (uint32_t)i;
}
// P6: 12-byte stores in runs: the output is cut into runs of R records; a wave writes 64 records that lie in 64 DIFFERENT runs (lane l ->
// run (base + l * stride)), record j of the run in trip j: exactly the store pattern of a counting-sort scatter with runs of R tuples
__global__ void __launch_bounds__(256) c_scatter12(T12* __restrict__ out, size_t n_runs, uint32_t R) {
  const size_t run = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (run >= n_runs) return;
  const size_t dst_run = (size_t)(mix64(run * 0x9E3779B97F4A7C15ull) % n_runs);           // (not a permutation: some runs are written twice, some never; the byte count holds)
  for (uint32_t j = 0; j < R; j++) { T12 t; t.a = (uint32_t)run; t.b = j; t.c = 7; out[dst_run * R + j] = t; }
}
// P6b: 12-byte stores, consecutive lanes -> consecutive records (coalesced)
__global__ void __launch_bounds__(256) c_stream_write12(T12* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { T12 t; t.a = (uint32_t)i; t.b = 1; t.c = 2; out[i] = t; }
}
__global__ void c_fill(uint32_t* a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = mix32((uint32_t)i);
}

static int run_calib(double arena_gb) {
  const size_t n_words = (size_t)(arena_gb * (1ull << 30)) / 4;
  uint32_t* a = nullptr; unsigned long long* sink = nullptr;
  CK(hipMalloc((void**)&a, n_words * 4)); CK(hipMalloc((void**)&sink, 8));
  hipLaunchKernelGGL(c_fill, dim3(4096), dim3(256), 0, 0, a, n_words); CK(hipDeviceSynchronize());
  const size_t half = n_words / 2;               // warm-up launches use the upper half, measured launches the lower half (both >> 256 MB of Infinity Cache)
  const int G = 256 * 16;
  printf("# pattern                      alg_bytes        ms      GB/s   note\n");
  auto rep = [&](const char* name, double bytes, float ms, const char* note) { printf("%-28s %14.0f %9.3f %9.1f   %s\n", name, bytes, ms, bytes / ms / 1e6, note); fflush(stdout); };
  float ms;
  // ---- reads
  const size_t sb = (size_t)2 << 30;             // 2 GiB streamed
  hipLaunchKernelGGL(c_stream_read16, dim3(G), dim3(256), 0, 0, (const uint4*)(a + half), sb / 16, sink);
  ms = timed([&] { hipLaunchKernelGGL(c_stream_read16, dim3(G), dim3(256), 0, 0, (const uint4*)a, sb / 16, sink); }); rep("c_stream_read16", (double)sb, ms, "2 GiB, 16 B/lane");
  hipLaunchKernelGGL(c_stream_read4, dim3(G), dim3(256), 0, 0, (const uint32_t*)(a + half), sb / 4, sink);
  ms = timed([&] { hipLaunchKernelGGL(c_stream_read4, dim3(G), dim3(256), 0, 0, (const uint32_t*)a, sb / 4, sink); }); rep("c_stream_read4", (double)sb, ms, "2 GiB, 4 B/lane");
  { const size_t n = sb / 12;
    hipLaunchKernelGGL(c_stream_read12, dim3(G), dim3(256), 0, 0, (const T12*)(a + half), n, sink);
    ms = timed([&] { hipLaunchKernelGGL(c_stream_read12, dim3(G), dim3(256), 0, 0, (const T12*)a, n, sink); }); rep("c_stream_read12", (double)n * 12, ms, "2 GiB, 12-byte records, one per lane"); }
  const uint32_t per = 16; const size_t nthr = (size_t)1 << 22;       // 64 M gathers
  hipLaunchKernelGGL(c_gather<4>, dim3(nthr / 256), dim3(256), 0, 0, (const uint32_t*)a, n_words, per, 1u, sink);
  ms = timed([&] { hipLaunchKernelGGL(c_gather<4>, dim3(nthr / 256), dim3(256), 0, 0, (const uint32_t*)a, n_words, per, 2u, sink); }); rep("c_gather<4>", (double)nthr * per * 4, ms, "64 M independent random 4-byte loads: alg = 4 B each; lines touched = 64 M");
  hipLaunchKernelGGL(c_gather<8>, dim3(nthr / 256), dim3(256), 0, 0, (const uint32_t*)a, n_words, per, 3u, sink);
  ms = timed([&] { hipLaunchKernelGGL(c_gather<8>, dim3(nthr / 256), dim3(256), 0, 0, (const uint32_t*)a, n_words, per, 4u, sink); }); rep("c_gather<8>", (double)nthr * per * 8, ms, "64 M independent random 8-byte loads");
  hipLaunchKernelGGL(c_gather<16>, dim3(nthr / 256), dim3(256), 0, 0, (const uint32_t*)a, n_words, per, 5u, sink);
  ms = timed([&] { hipLaunchKernelGGL(c_gather<16>, dim3(nthr / 256), dim3(256), 0, 0, (const uint32_t*)a, n_words, per, 6u, sink); }); rep("c_gather<16>", (double)nthr * per * 16, ms, "64 M independent random 16-byte loads");
  { const uint32_t steps = 32; const size_t nt = (size_t)1 << 21;      // 64 M dependent gathers, 2 M chains (one wave per block, 8 waves per SIMD possible)
    hipLaunchKernelGGL(c_chase4, dim3(nt / 64), dim3(64), 0, 0, (const uint32_t*)a, n_words, steps, 7u, sink);
    ms = timed([&] { hipLaunchKernelGGL(c_chase4, dim3(nt / 64), dim3(64), 0, 0, (const uint32_t*)a, n_words, steps, 8u, sink); }); rep("c_chase4", (double)nt * steps * 4, ms, "64 M DEPENDENT random 4-byte loads (2 M chains of 32)"); }
  // ---- writes
  hipLaunchKernelGGL(c_stream_write16, dim3(G), dim3(256), 0, 0, (uint4*)(a + half), sb / 16);
  ms = timed([&] { hipLaunchKernelGGL(c_stream_write16, dim3(G), dim3(256), 0, 0, (uint4*)a, sb / 16); }); rep("c_stream_write16", (double)sb, ms, "2 GiB, 16 B/lane");
  hipLaunchKernelGGL(c_stream_write4, dim3(G), dim3(256), 0, 0, (uint32_t*)(a + half), sb / 4);
  ms = timed([&] { hipLaunchKernelGGL(c_stream_write4, dim3(G), dim3(256), 0, 0, (uint32_t*)a, sb / 4); }); rep("c_stream_write4", (double)sb, ms, "2 GiB, 4 B/lane");
  { const size_t n = ((size_t)1 << 30) / 12;
    hipLaunchKernelGGL(c_stream_write12, dim3(G), dim3(256), 0, 0, (T12*)(a + half), n);
    ms = timed([&] { hipLaunchKernelGGL(c_stream_write12, dim3(G), dim3(256), 0, 0, (T12*)a, n); }); rep("c_stream_write12", (double)n * 12, ms, "1 GiB of 12-byte records, consecutive lanes -> consecutive records"); }
  const uint32_t Rs[4] = {1, 8, 32, 256};
  for (int q = 0; q < 4; q++) {
    const uint32_t R = Rs[q]; const size_t n = ((size_t)1 << 30) / 12, n_runs = n / R;
    hipLaunchKernelGGL(c_scatter12, dim3((uint32_t)((n_runs + 255) / 256)), dim3(256), 0, 0, (T12*)(a + half), n_runs, R);
    ms = timed([&] { hipLaunchKernelGGL(c_scatter12, dim3((uint32_t)((n_runs + 255) / 256)), dim3(256), 0, 0, (T12*)a, n_runs, R); });
    char nm[64], nt[128]; snprintf(nm, sizeof nm, "c_scatter12 R=%u", R); snprintf(nt, sizeof nt, "1 GiB of 12-byte stores, every lane its own run of %u records", R);
    rep(nm, (double)n_runs * R * 12, ms, nt);
  }
  CK(hipFree(a)); CK(hipFree(sink));
  return 0;
}

// =====================================================================================================================
// sort variants
// =====================================================================================================================
struct Tup { uint32_t key, lo, hi; };
struct Tup8 { uint32_t key, lo; };
template <class T> __global__ void s_gen(T* t, uint32_t n, uint32_t nk) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    T x; memset(&x, 0, sizeof x); x.key = mix32(i * 2654435761u + 12345u) % nk; x.lo = i;
    if (sizeof(T) == 12) ((uint32_t*)&x)[2] = mix32(i);
    t[i] = x;
  }
}
// ---- variant A: the shipped pair (k_seed_split / k_seed_bins), parameterised: nc coarse bins (key >> fb), chunk tuples per block
template <class T> __global__ void __launch_bounds__(1024) sA_chist(const T* __restrict__ tmp, uint32_t n, uint32_t fb, uint32_t nc, uint32_t* __restrict__ chist) {
  extern __shared__ uint32_t lh[];
  for (uint32_t c = threadIdx.x; c < nc; c += blockDim.x) lh[c] = 0;
  __syncthreads();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) atomicAdd(&lh[tmp[i].key >> fb], 1u);
  __syncthreads();
  for (uint32_t c = threadIdx.x; c < nc; c += blockDim.x) if (lh[c]) atomicAdd(&chist[c], lh[c]);
}
__global__ void __launch_bounds__(1024) s_scan(const uint32_t* __restrict__ in, uint32_t* __restrict__ base, uint32_t* __restrict__ cur, uint32_t nc) {   // nc <= 4096
  __shared__ uint32_t s_part[16];
  const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  uint32_t v[4], sum = 0;
  for (int q = 0; q < 4; q++) { const uint32_t c = 4 * t + q; v[q] = c < nc ? in[c] : 0u; sum += v[q]; }
  uint32_t incl = sum;
  for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d, 64); if ((int)lane >= d) incl += x; }
  if (lane == 63) s_part[wv] = incl;
  __syncthreads();
  uint32_t pre = incl - sum;
  for (uint32_t q = 0; q < wv; q++) pre += s_part[q];
  for (int q = 0; q < 4; q++) { const uint32_t c = 4 * t + q; if (c < nc) { base[c] = pre; cur[c] = pre; } pre += v[q]; if (c + 1 == nc) base[nc] = pre; }
}
template <class T> __global__ void __launch_bounds__(1024) sA_split(const T* __restrict__ tmp, T* __restrict__ mid, uint32_t n, uint32_t fb, uint32_t nc, uint32_t chunk, uint32_t* __restrict__ ccur) {
  extern __shared__ uint32_t lds[];
  uint32_t* lh = lds; uint32_t* lb = lds + nc;
  const uint32_t i0 = blockIdx.x * chunk, i1 = min(i0 + chunk, n);
  if (i0 >= n) return;
  for (uint32_t c = threadIdx.x; c < nc; c += blockDim.x) lh[c] = 0;
  __syncthreads();
  for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) atomicAdd(&lh[tmp[i].key >> fb], 1u);
  __syncthreads();
  for (uint32_t c = threadIdx.x; c < nc; c += blockDim.x) { const uint32_t m = lh[c]; if (m) lb[c] = atomicAdd(&ccur[c], m); lh[c] = 0; }
  __syncthreads();
  for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) { const T t = tmp[i]; const uint32_t c = t.key >> fb; mid[lb[c] + atomicAdd(&lh[c], 1u)] = t; }
}
template <class T, int NF> __global__ void __launch_bounds__(1024) sA_bins(const T* __restrict__ mid, T* __restrict__ srt, uint32_t fb, const uint32_t* __restrict__ cbase) {
  __shared__ uint32_t fh[NF], s_part[16];
  const uint32_t c = blockIdx.x, lo = cbase[c], hi = cbase[c + 1];
  if (lo == hi) return;
  const uint32_t t = threadIdx.x, fm = (1u << fb) - 1u;
  for (uint32_t q = t; q < NF; q += blockDim.x) fh[q] = 0;
  __syncthreads();
  for (uint32_t i = lo + t; i < hi; i += blockDim.x) atomicAdd(&fh[mid[i].key & fm], 1u);
  __syncthreads();
  // exclusive scan of NF counts: NF / 1024 consecutive per thread (NF <= 4096), or one per thread for t < NF
  constexpr int PER = NF > 1024 ? NF / 1024 : 1;
  uint32_t v[PER], sum = 0;
  for (int q = 0; q < PER; q++) { const uint32_t k = PER * t + q; v[q] = k < (uint32_t)NF ? fh[k] : 0u; sum += v[q]; }
  uint32_t incl = sum;
  for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d, 64); if ((int)(t & 63u) >= d) incl += x; }
  if ((t & 63u) == 63u) s_part[t >> 6] = incl;
  __syncthreads();
  uint32_t pre = incl - sum;
  for (uint32_t q = 0; q < (t >> 6); q++) pre += s_part[q];
  for (int q = 0; q < PER; q++) { const uint32_t k = PER * t + q; if (k < (uint32_t)NF) fh[k] = lo + pre; pre += v[q]; }
  __syncthreads();
  for (uint32_t i = lo + t; i < hi; i += blockDim.x) { const T x = mid[i]; const uint32_t p = atomicAdd(&fh[x.key & fm], 1u); srt[p] = x; }
}
// ---- variant B: single-pass split.  The producer of the tuples (k_seed_keys) already counts its tuples per coarse bin in LDS; if every block
// owns a contiguous range of tmp and writes ITS histogram row, a column scan gives every (block, bin) its place and the split reads each
// tuple once.  sB_rows stands in for the producer's counting (timed apart).
template <class T> __global__ void __launch_bounds__(1024) sB_rows(const T* __restrict__ tmp, uint32_t n, uint32_t fb, uint32_t nc, uint32_t chunk, uint32_t* __restrict__ rows) {
  extern __shared__ uint32_t lh[];
  const uint32_t i0 = blockIdx.x * chunk, i1 = min(i0 + chunk, n);
  for (uint32_t c = threadIdx.x; c < nc; c += blockDim.x) lh[c] = 0;
  __syncthreads();
  for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) atomicAdd(&lh[tmp[i].key >> fb], 1u);
  __syncthreads();
  for (uint32_t c = threadIdx.x; c < nc; c += blockDim.x) rows[(size_t)blockIdx.x * nc + c] = lh[c];
}
// column sums -> chist ; then (after s_scan) rows[b][c] = cbase[c] + sum of rows[b'][c], b' < b
__global__ void sB_colsum(const uint32_t* __restrict__ rows, uint32_t nb, uint32_t nc, uint32_t* __restrict__ chist) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; if (c >= nc) return;
  uint32_t s = 0; for (uint32_t b = 0; b < nb; b++) s += rows[(size_t)b * nc + c];
  chist[c] = s;
}
__global__ void sB_colscan(uint32_t* __restrict__ rows, uint32_t nb, uint32_t nc, const uint32_t* __restrict__ cbase) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; if (c >= nc) return;
  uint32_t s = cbase[c]; for (uint32_t b = 0; b < nb; b++) { const uint32_t v = rows[(size_t)b * nc + c]; rows[(size_t)b * nc + c] = s; s += v; }
}
template <class T> __global__ void __launch_bounds__(1024) sB_split(const T* __restrict__ tmp, T* __restrict__ mid, uint32_t n, uint32_t fb, uint32_t nc, uint32_t chunk, const uint32_t* __restrict__ rows) {
  extern __shared__ uint32_t lh[];
  const uint32_t i0 = blockIdx.x * chunk, i1 = min(i0 + chunk, n);
  for (uint32_t c = threadIdx.x; c < nc; c += blockDim.x) lh[c] = rows[(size_t)blockIdx.x * nc + c];
  __syncthreads();
  for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) { const T t = tmp[i]; mid[atomicAdd(&lh[t.key >> fb], 1u)] = t; }
}
// ---- variant C: second pass through LDS.  One block per coarse bin; the bin is taken in pieces of PIECE tuples: a piece is loaded, ranked
// inside its fine bin by LDS atomics, placed in LDS in fine-bin order, and copied out by consecutive lanes: a lane's store lands next to its
// neighbour's (within a fine bin's run of the piece), so the 12-byte stores leave the CU as whole cache-line segments.
template <class T, int NF, int PIECE> __global__ void __launch_bounds__(1024) sC_bins(const T* __restrict__ mid, T* __restrict__ srt, uint32_t fb, const uint32_t* __restrict__ cbase) {
  __shared__ uint32_t fh[NF], fcur[NF], ph[NF], pst[NF], s_part[16];
  __shared__ T stage[PIECE];
  __shared__ uint32_t dst[PIECE];
  const uint32_t c = blockIdx.x, lo = cbase[c], hi = cbase[c + 1];
  if (lo == hi) return;
  const uint32_t t = threadIdx.x, fm = (1u << fb) - 1u;
  for (uint32_t q = t; q < NF; q += blockDim.x) fh[q] = 0;
  __syncthreads();
  for (uint32_t i = lo + t; i < hi; i += blockDim.x) atomicAdd(&fh[mid[i].key & fm], 1u);
  __syncthreads();
  constexpr int PER = NF > 1024 ? NF / 1024 : 1;
  auto scan = [&](uint32_t* arr, uint32_t* out, uint32_t add) {      // out[k] = add + exclusive prefix of arr
    uint32_t v[PER], sum = 0;
    for (int q = 0; q < PER; q++) { const uint32_t k = PER * t + q; v[q] = k < (uint32_t)NF ? arr[k] : 0u; sum += v[q]; }
    uint32_t incl = sum;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d, 64); if ((int)(t & 63u) >= d) incl += x; }
    if ((t & 63u) == 63u) s_part[t >> 6] = incl;
    __syncthreads();
    uint32_t pre = incl - sum;
    for (uint32_t q = 0; q < (t >> 6); q++) pre += s_part[q];
    for (int q = 0; q < PER; q++) { const uint32_t k = PER * t + q; if (k < (uint32_t)NF) out[k] = add + pre; pre += v[q]; }
    __syncthreads();
  };
  scan(fh, fcur, lo);                                   // fcur: next free global place of every fine bin
  for (uint32_t p0 = lo; p0 < hi; p0 += PIECE) {
    const uint32_t p1 = min(p0 + PIECE, hi);
    for (uint32_t q = t; q < NF; q += blockDim.x) ph[q] = 0;
    __syncthreads();
    T mine[PIECE / 1024]; uint32_t rk[PIECE / 1024];
#pragma unroll
    for (int j = 0; j < PIECE / 1024; j++) { const uint32_t i = p0 + j * 1024 + t; if (i < p1) { mine[j] = mid[i]; rk[j] = atomicAdd(&ph[mine[j].key & fm], 1u); } }
    __syncthreads();
    scan(ph, pst, 0);                                   // pst: where a fine bin's run starts inside the piece
#pragma unroll
    for (int j = 0; j < PIECE / 1024; j++) { const uint32_t i = p0 + j * 1024 + t; if (i < p1) { const uint32_t f = mine[j].key & fm, s = pst[f] + rk[j]; stage[s] = mine[j]; dst[s] = fcur[f] + rk[j]; } }
    __syncthreads();
    for (uint32_t s = t; s < p1 - p0; s += blockDim.x) srt[dst[s]] = stage[s];
    __syncthreads();
    for (uint32_t q = t; q < NF; q += blockDim.x) fcur[q] += ph[q];
    __syncthreads();
  }
}
template <class T> __global__ void s_check(const T* __restrict__ srt, uint32_t n, unsigned long long* out /* [0] disorder count, [1] payload sum */) {
  unsigned long long bad = 0, sum = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { if (i + 1 < n && srt[i].key > srt[i + 1].key) bad++; sum += mix32(srt[i].lo) ^ srt[i].key; }
  for (int d = 32; d > 0; d >>= 1) { bad += __shfl_down(bad, d, 64); sum += __shfl_down(sum, d, 64); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], bad); atomicAdd(&out[1], sum); }
}

template <class T> static void sort_suite(uint32_t n) {
  const uint32_t KB = 19, nk = 1u << KB;
  T *tmp, *mid, *srt; uint32_t *chist, *cbase, *ccur, *rows; unsigned long long* chk;
  CK(hipMalloc((void**)&tmp, (size_t)n * sizeof(T))); CK(hipMalloc((void**)&mid, (size_t)n * sizeof(T))); CK(hipMalloc((void**)&srt, (size_t)n * sizeof(T)));
  CK(hipMalloc((void**)&chist, 4097 * 4)); CK(hipMalloc((void**)&cbase, 4097 * 4)); CK(hipMalloc((void**)&ccur, 4097 * 4)); CK(hipMalloc((void**)&rows, (size_t)8192 * 4096 * 4)); CK(hipMalloc((void**)&chk, 16));
  hipLaunchKernelGGL(s_gen<T>, dim3(4096), dim3(256), 0, 0, tmp, n, nk); CK(hipDeviceSynchronize());
  unsigned long long ref[2] = {0, 0};
  { CK(hipMemset(chk, 0, 16)); hipLaunchKernelGGL(s_check<T>, dim3(2048), dim3(256), 0, 0, (const T*)tmp, n, chk); CK(hipMemcpy(ref, chk, 16, hipMemcpyDeviceToHost)); }
  auto verify = [&](const char* name, float a, float b, float c2) {
    unsigned long long got[2]; CK(hipMemset(chk, 0, 16)); hipLaunchKernelGGL(s_check<T>, dim3(2048), dim3(256), 0, 0, (const T*)srt, n, chk); CK(hipMemcpy(got, chk, 16, hipMemcpyDeviceToHost));
    const double gb = (double)n * sizeof(T) / 1e9;
    printf("%-44s pre %7.3f  split %7.3f  bins %7.3f  = %7.3f ms   (%.0f / %.0f GB/s r+w)  %s\n", name, a, b, c2, b + c2, 2 * gb / (b * 1e-3), 2 * gb / (c2 * 1e-3), (got[0] == 0 && got[1] == ref[1]) ? "ok" : "WRONG");
    fflush(stdout);
    CK(hipMemset(srt, 0, (size_t)n * sizeof(T)));
  };
  printf("# %u tuples of %zu bytes, %u keys\n", n, sizeof(T), nk);
  for (int rep = 0; rep < 2; rep++) {
    // A: shipped structure
    const uint32_t cfgA[][2] = {{9, 32768}, {9, 65536}, {9, 131072}, {10, 65536}, {10, 131072}, {11, 131072}, {11, 262144}};
    for (auto& cf : cfgA) {
      const uint32_t fb = cf[0], chunk = cf[1], nc = nk >> fb;
      float t0 = timed([&] { CK(hipMemsetAsync(chist, 0, 4097 * 4, 0)); hipLaunchKernelGGL(sA_chist<T>, dim3(2048), dim3(1024), nc * 4, 0, (const T*)tmp, n, fb, nc, chist); hipLaunchKernelGGL(s_scan, dim3(1), dim3(1024), 0, 0, (const uint32_t*)chist, cbase, ccur, nc); });
      float t1 = timed([&] { hipLaunchKernelGGL(sA_split<T>, dim3((n + chunk - 1) / chunk), dim3(1024), nc * 8, 0, (const T*)tmp, mid, n, fb, nc, chunk, ccur); });
      float t2 = timed([&] {
        if (fb == 9) hipLaunchKernelGGL((sA_bins<T, 512>), dim3(nc), dim3(1024), 0, 0, (const T*)mid, srt, fb, (const uint32_t*)cbase);
        else if (fb == 10) hipLaunchKernelGGL((sA_bins<T, 1024>), dim3(nc), dim3(1024), 0, 0, (const T*)mid, srt, fb, (const uint32_t*)cbase);
        else hipLaunchKernelGGL((sA_bins<T, 2048>), dim3(nc), dim3(1024), 0, 0, (const T*)mid, srt, fb, (const uint32_t*)cbase); });
      char nm[96]; snprintf(nm, sizeof nm, "A two-pass split, nc=%u fine=%u chunk=%u", nc, 1u << fb, chunk); verify(nm, t0, t1, t2);
    }
    // B: single-pass split from per-block rows
    const uint32_t cfgB[][2] = {{9, 32768}, {9, 65536}, {10, 65536}, {10, 131072}};
    for (auto& cf : cfgB) {
      const uint32_t fb = cf[0], chunk = cf[1], nc = nk >> fb, nb = (n + chunk - 1) / chunk;
      float t0 = timed([&] { hipLaunchKernelGGL(sB_rows<T>, dim3(nb), dim3(1024), nc * 4, 0, (const T*)tmp, n, fb, nc, chunk, rows); });
      float t1 = timed([&] {
        hipLaunchKernelGGL(sB_colsum, dim3((nc + 255) / 256), dim3(256), 0, 0, (const uint32_t*)rows, nb, nc, chist);
        hipLaunchKernelGGL(s_scan, dim3(1), dim3(1024), 0, 0, (const uint32_t*)chist, cbase, ccur, nc);
        hipLaunchKernelGGL(sB_colscan, dim3((nc + 255) / 256), dim3(256), 0, 0, rows, nb, nc, (const uint32_t*)cbase);
        hipLaunchKernelGGL(sB_split<T>, dim3(nb), dim3(1024), nc * 4, 0, (const T*)tmp, mid, n, fb, nc, chunk, (const uint32_t*)rows); });
      float t2 = timed([&] {
        if (fb == 9) hipLaunchKernelGGL((sA_bins<T, 512>), dim3(nc), dim3(1024), 0, 0, (const T*)mid, srt, fb, (const uint32_t*)cbase);
        else hipLaunchKernelGGL((sA_bins<T, 1024>), dim3(nc), dim3(1024), 0, 0, (const T*)mid, srt, fb, (const uint32_t*)cbase); });
      char nm[96]; snprintf(nm, sizeof nm, "B one-pass split (rows), nc=%u chunk=%u", nc, chunk); verify(nm, t0, t1, t2);
    }
    // C: A's split + second pass staged through LDS
    {
      const uint32_t fb = 9, chunk = 32768, nc = nk >> fb;
      float t0 = timed([&] { CK(hipMemsetAsync(chist, 0, 4097 * 4, 0)); hipLaunchKernelGGL(sA_chist<T>, dim3(2048), dim3(1024), nc * 4, 0, (const T*)tmp, n, fb, nc, chist); hipLaunchKernelGGL(s_scan, dim3(1), dim3(1024), 0, 0, (const uint32_t*)chist, cbase, ccur, nc); });
      float t1 = timed([&] { hipLaunchKernelGGL(sA_split<T>, dim3((n + chunk - 1) / chunk), dim3(1024), nc * 8, 0, (const T*)tmp, mid, n, fb, nc, chunk, ccur); });
      float t2 = timed([&] { hipLaunchKernelGGL((sC_bins<T, 512, 4096>), dim3(nc), dim3(1024), 0, 0, (const T*)mid, srt, fb, (const uint32_t*)cbase); });
      verify("C split as A + bins staged in LDS, piece 4096", t0, t1, t2);
      t2 = timed([&] { hipLaunchKernelGGL((sC_bins<T, 512, 2048>), dim3(nc), dim3(1024), 0, 0, (const T*)mid, srt, fb, (const uint32_t*)cbase); });
      verify("C split as A + bins staged in LDS, piece 2048", t0, t1, t2);
    }
  }
  CK(hipFree(tmp)); CK(hipFree(mid)); CK(hipFree(srt)); CK(hipFree(chist)); CK(hipFree(cbase)); CK(hipFree(ccur)); CK(hipFree(rows)); CK(hipFree(chk));
}

int main(int argc, char** argv) {
  CK(hipSetDevice(0)); CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
  const std::string mode = argc > 1 ? argv[1] : "";
  if (mode == "calib") return run_calib(argc > 2 ? atof(argv[2]) : 12.0);
  if (mode == "sort") {
    const uint32_t n = (uint32_t)((argc > 2 ? atof(argv[2]) : 60.0) * 1e6);
    sort_suite<Tup>(n);
    sort_suite<Tup8>(n);
    return 0;
  }
  fprintf(stderr, "usage: mb calib [arena_GB] | mb sort [Mtuples]\n");
  return 1;
}
