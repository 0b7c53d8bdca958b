// smr_engine_swseam.hpp -- the Smith-Waterman kernels at the ssw.h seam (included by smr_engine.hip): the device self-check of the packed kernels, smr_ssw_batch, smr_sw_mode,
// smr_walk_rounds.
// (one translation unit: no include guard games -- this file is text of smr_engine.hip, cut out along its stages)

// =================================================================================================
// Device self-check of the packed Smith-Waterman kernel against the 32-bit one (both on the GPU): one wave per case, seeded
// pseudo-random read (1..max_m nt, ~1.5 % N) against either a mutated copy of it with substitutions and indels or a random
// sequence; forward pass and the reverse-direction pass on the prefixes ending in the forward end cell, like k_chain's two calls.
__device__ __forceinline__ uint32_t sc_hash(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu ^ (c + 0x165667B1u) * 0xC2B2AE35u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}
__global__ void __launch_bounds__(64) k_sw_selfcheck(uint32_t n_cases, uint32_t seed, uint32_t max_m, uint32_t lds_m, uint32_t lds_n,
                                                     int match, int mismatch, int scoreN, int go, int ge, int mode_b, unsigned long long* out) {
  SMR_DYN_LDS(unsigned char, lds_raw);
  uint8_t* rdq = lds_raw;
  uint8_t* rfq = rdq + lds_m;
  int* bound = (int*)(rfq + lds_n);
  __shared__ int s_n;
  const int lane = smr::lane_id();
  for (uint32_t cs = blockIdx.x; cs < n_cases; cs += gridDim.x) {
    const uint32_t hm = sc_hash(seed, cs, 1);
    const int m = 1 + (int)(hm % max_m);
    for (int q = lane; q < m; q += 64) { const uint32_t h = sc_hash(seed, cs, 100u + (uint32_t)q); rdq[q] = (h & 63u) == 0 ? 4 : (uint8_t)((h >> 8) & 3u); }
    __syncthreads();
    if (lane == 0) {
      int n = 0;
      const bool homolog = (hm >> 20) & 3u;                       // 3 of 4 cases: a mutated copy (with a random flank), else unrelated
      const int flank = (int)((hm >> 24) & 15u);
      for (int q = 0; q < flank; q++) rfq[n++] = (uint8_t)(sc_hash(seed, cs, 5000u + (uint32_t)q) & 3u);
      for (int q = 0; q < m && n + 2 < (int)lds_n; q++) {
        const uint32_t h = sc_hash(seed, cs, 9000u + (uint32_t)q);
        if (!homolog) { rfq[n++] = (uint8_t)(h & 3u); continue; }
        const uint32_t ev = (h >> 4) & 63u;
        if (ev == 0) continue;                                      // deletion in the reference
        if (ev == 1) rfq[n++] = (uint8_t)((h >> 12) & 3u);          // insertion
        if (ev == 2) { rfq[n++] = 4; continue; }                    // N in the reference
        rfq[n++] = ev < 6 ? (uint8_t)((h >> 16) & 3u) : (rdq[q] == 4 ? (uint8_t)0 : rdq[q]);
      }
      for (int q = 0; q < flank && n + 1 < (int)lds_n; q++) rfq[n++] = (uint8_t)(sc_hash(seed, cs, 7000u + (uint32_t)q) & 3u);
      s_n = n;
    }
    __syncthreads();
    const int n = s_n;
    const smr::SwRes a0 = smr::sw_wave(rdq, m, 0, 1, rfq, n, 0, 1, bound, match, mismatch, scoreN, go, ge, 0);
    __syncthreads();
    const smr::SwRes a1 = smr::sw_wave(rdq, m, 0, 1, rfq, n, 0, 1, bound, match, mismatch, scoreN, go, ge, mode_b);
    __syncthreads();
    bool bad = a0.score != a1.score || a0.end_ref != a1.end_ref || a0.end_read != a1.end_read;
    if (a0.score > 0 && a0.end_ref >= 0) {
      const smr::SwRes b0 = smr::sw_wave(rdq, a0.end_read + 1, a0.end_read, -1, rfq, a0.end_ref + 1, a0.end_ref, -1, bound, match, mismatch, scoreN, go, ge, 0);
      __syncthreads();
      const smr::SwRes b1 = smr::sw_wave(rdq, a0.end_read + 1, a0.end_read, -1, rfq, a0.end_ref + 1, a0.end_ref, -1, bound, match, mismatch, scoreN, go, ge, mode_b);
      __syncthreads();
      bad = bad || b0.score != b1.score || b0.end_ref != b1.end_ref || b0.end_read != b1.end_read;
    }
    if (lane == 0) { atomicAdd(&out[0], 1ull); if (bad) atomicAdd(&out[1], 1ull); atomicAdd(&out[2], (unsigned long long)a0.score); }
    __syncthreads();
  }
}

extern "C" int smr_sw_selfcheck(smr_ctx* c, uint32_t n_cases, uint32_t seed, uint32_t max_len, uint64_t* n_bad) {
  if (!c || !n_bad || max_len == 0 || max_len > 4000) return SMR_ERR_ARG;
  (void)hipSetDevice(c->device);
  DevPool pool;
  IB_GET(d, unsigned long long, 3);
  HIPCHK(c, hipMemsetAsync(d, 0, 3 * 8, c->stream));
  const uint32_t lm = (max_len + 15) & ~15u, ln = (max_len + max_len / 16 + 64 + 15) & ~15u;
  const size_t lds = (size_t)lm + ln + (size_t)2 * ln * 4;
  const int sc[2][3] = {{2, -3, -3}, {5, -4, -4}};
  for (int k = 0; k < 2 && n_cases; k++)
    hipLaunchKernelGGL(k_sw_selfcheck, dim3(std::min<uint32_t>(n_cases, (uint32_t)c->n_cu * 8u)), dim3(64), lds, c->stream, n_cases, seed + 7919u * (uint32_t)k, max_len, lm, ln,
                       sc[k][0], sc[k][1], sc[k][2], 5, 2, std::max(c->sw_mode, 1), d);
  unsigned long long h[3] = {0, 0, 0};
  HIPCHK(c, hipMemcpyAsync(h, d, 3 * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (h[0] != 2ull * n_cases) { set_err(c, "SW self-check did not run all cases"); return SMR_ERR_DEVICE; }
  *n_bad = h[1];
  return SMR_OK;
}


// =================================================================================================
// Batched Smith-Waterman at the ssw.h seam (SURVEY.md 8b "existing C ABI"): what k_chain does per candidate -- ssw_align(prof, ref,
// refLen, gapO, gapE, flag = 2, filters, 0, 0) (ssw.c:834-941) without the CIGAR -- for n independent (read, reference window) pairs,
// one wave per pair.  A unit-test surface for the SW kernels against the reference's own ssw.c (tests/golden/ssw_pairs.json).
// =================================================================================================
template <bool STRIPED>
__global__ void __launch_bounds__(64) k_ssw_batch(uint32_t n_pairs, const uint8_t* __restrict__ reads, const unsigned long long* __restrict__ read_off,
                                                  const uint8_t* __restrict__ refs, const unsigned long long* __restrict__ ref_off, uint32_t lds_m, uint32_t lds_n,
                                                  int match, int mismatch, int scoreN, int go, int ge, uint32_t filters, int mode, int* __restrict__ out, uint16_t* scr_all, uint32_t scr_stride) {
  SMR_DYN_LDS(unsigned char, lds_raw);
  uint8_t* rdq = lds_raw;
  uint8_t* rfq = rdq + lds_m;
  int* bound = (int*)(rfq + lds_n);
  uint16_t* scr = scr_all ? scr_all + (size_t)blockIdx.x * scr_stride : nullptr;      // (mode < 0: the striped slow path)
  const int lane = smr::lane_id();
  for (uint32_t pi = blockIdx.x; pi < n_pairs; pi += gridDim.x) {
    const int m = (int)(read_off[pi + 1] - read_off[pi]), n = (int)(ref_off[pi + 1] - ref_off[pi]);
    for (int q = lane; q < m; q += 64) rdq[q] = reads[read_off[pi] + q];
    for (int q = lane; q < n; q += 64) rfq[q] = refs[ref_off[pi] + q];
    __syncthreads();
    int res[5] = {0, -1, -1, -1, m - 1};          // score1, ref_begin1, ref_end1, read_begin1, read_end1
    if (m > 0 && n > 0) {
      const smr::SwRes fw = smr::sw_wave_t<STRIPED>(rdq, m, 0, 1, rfq, n, 0, 1, bound, match, mismatch, scoreN, go, ge, mode, 0, 0, scr);
      __syncthreads();
      res[0] = fw.score > 65535 ? 65535 : fw.score; res[2] = fw.end_ref; res[4] = fw.end_read;
      if ((uint32_t)res[0] >= filters && fw.score > 0) {
        const smr::SwRes bw = smr::sw_wave_t<STRIPED>(rdq, fw.end_read + 1, fw.end_read, -1, rfq, fw.end_ref + 1, fw.end_ref, -1, bound, match, mismatch, scoreN, go, ge, mode, res[0], fw.word, scr);
        __syncthreads();
        res[1] = fw.end_ref - bw.end_ref; res[3] = fw.end_read - bw.end_read;
      }
    }
    if (lane < 5) out[(size_t)pi * 5 + lane] = res[lane];
    __syncthreads();
  }
}

// the same through the four-problems-per-wave kernel (sw_wave_x4): row g of the wave takes pair 4 b + g; forward pass, then the reverse
// pass of the rows whose score passed the filter (the others idle)
__global__ void __launch_bounds__(64) k_ssw_batch_x4(uint32_t n_pairs, const uint8_t* __restrict__ reads, const unsigned long long* __restrict__ read_off,
                                                     const uint8_t* __restrict__ refs, const unsigned long long* __restrict__ ref_off, uint32_t lds_m, uint32_t lds_n,
                                                     int match, int mismatch, int scoreN, int go, int ge, uint32_t filters, int* __restrict__ out) {
  SMR_DYN_LDS(unsigned char, lds_raw);
  const int lane = smr::lane_id(), g = lane >> 4, gl = lane & 15;
  uint8_t* rdq = lds_raw + (size_t)g * lds_m;
  uint8_t* rfq = lds_raw + (size_t)4 * lds_m + (size_t)g * lds_n;
  for (uint32_t p0 = blockIdx.x * 4; p0 < n_pairs; p0 += gridDim.x * 4) {
    const uint32_t pi = p0 + g;
    const bool have = pi < n_pairs;
    int m = have ? (int)(read_off[pi + 1] - read_off[pi]) : 0, n = have ? (int)(ref_off[pi + 1] - ref_off[pi]) : 0;
    if (n == 0) m = 0;
    __syncthreads();
    bool hasn = false;
    for (int q = gl; q < m; q += 16) rdq[q] = reads[read_off[pi] + q];
    for (int q = gl; q < n; q += 16) { const uint8_t ch = refs[ref_off[pi] + q]; rfq[q] = ch; hasn |= ch == 4; }
    __syncthreads();
    int mm = m;
    for (int d = 32; d > 0; d >>= 1) mm = max(mm, __shfl_xor(mm, d, 64));
    const bool hn = __any(hasn);
    int res[5] = {0, -1, -1, -1, m - 1};
    const smr::SwRes fw = smr::sw_wave_x4(rdq, m, 0, 1, rfq, n, 0, 1, match, mismatch, scoreN, go, ge, mm, hn);
    res[0] = fw.score > 65535 ? 65535 : fw.score; res[2] = fw.end_ref; res[4] = fw.end_read;
    const bool rev = m > 0 && (uint32_t)res[0] >= filters && fw.score > 0;
    const smr::SwRes bw = smr::sw_wave_x4(rdq, rev ? fw.end_read + 1 : 0, fw.end_read, -1, rfq, rev ? fw.end_ref + 1 : 0, fw.end_ref, -1, match, mismatch, scoreN, go, ge, mm, hn);
    if (rev) { res[1] = fw.end_ref - bw.end_ref; res[3] = fw.end_read - bw.end_read; }
    if (have && gl < 5) out[(size_t)pi * 5 + gl] = res[gl];
  }
}

extern "C" int smr_ssw_batch(smr_ctx* c, uint32_t n_pairs, const uint8_t* reads, const uint64_t* read_off, const uint8_t* refs, const uint64_t* ref_off,
                             int match, int mismatch, int score_N, int gap_open, int gap_ext, uint32_t filters, int mode, int32_t* out) {
  if (!c || !read_off || !ref_off || !out || mode < 0 || mode > 4) return SMR_ERR_ARG;
  // (modes 0 - 3 are the fast kernels: only under the schemes whose answers they share with ssw.c; mode 4 = the striped slow path, any scheme)
  if (mode != 4) if (const char* why = scheme_unsupported(mismatch, score_N, gap_open, gap_ext)) { set_err(c, why); return SMR_ERR_ARG; }
  if (n_pairs == 0) return SMR_OK;
  (void)hipSetDevice(c->device);
  uint64_t mx_m = 1, mx_n = 1;
  for (uint32_t i = 0; i < n_pairs; i++) { mx_m = std::max(mx_m, read_off[i + 1] - read_off[i]); mx_n = std::max(mx_n, ref_off[i + 1] - ref_off[i]); }
  const uint32_t lm = (uint32_t)((mx_m + 15) & ~15ull), ln = (uint32_t)((mx_n + 15) & ~15ull);
  const size_t lds = (size_t)lm + ln + (size_t)2 * ln * 4;
  if (lds > 60 * 1024) { set_err(c, "smr_ssw_batch: sequences too long for one LDS tile (read + 9 x reference window <= 60 KB)"); return SMR_ERR_CAPACITY; }
  DevPool pool;
  IB_GET(d_reads, uint8_t, read_off[n_pairs] + 1); IB_GET(d_refs, uint8_t, ref_off[n_pairs] + 1);
  IB_GET(d_ro, unsigned long long, (size_t)n_pairs + 1); IB_GET(d_fo, unsigned long long, (size_t)n_pairs + 1);
  IB_GET(d_out, int, (size_t)n_pairs * 5);
  if (read_off[n_pairs]) HIPCHK(c, hipMemcpyAsync(d_reads, reads, read_off[n_pairs], hipMemcpyHostToDevice, c->stream));
  if (ref_off[n_pairs]) HIPCHK(c, hipMemcpyAsync(d_refs, refs, ref_off[n_pairs], hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_ro, read_off, ((size_t)n_pairs + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_fo, ref_off, ((size_t)n_pairs + 1) * 8, hipMemcpyHostToDevice, c->stream));
  if (mode == 3) {          // four pairs per wave (the kernel k_chain batches candidate windows with): spans up to SW_X4_MAX_ROWS, numbers within the packed range
    for (uint32_t i = 0; i < n_pairs; i++) {
      const uint64_t m = read_off[i + 1] - read_off[i], n = ref_off[i + 1] - ref_off[i];
      if (m > SW_X4_MAX_ROWS || !((long long)m * match + 255 < 32768 && n + 128 <= 8191 && gap_open + mismatch >= 0 && gap_open + score_N >= 0 && match + gap_open <= 255 && score_N + gap_open <= 255)) {
        set_err(c, "smr_ssw_batch mode 3: a pair is outside the range of the four-problem kernel"); return SMR_ERR_ARG;
      }
    }
    hipLaunchKernelGGL(k_ssw_batch_x4, dim3(std::min<uint32_t>((n_pairs + 3) / 4, (uint32_t)c->n_cu * 8u)), dim3(64), (size_t)4 * (lm + ln), c->stream, n_pairs, (const uint8_t*)d_reads,
                       (const unsigned long long*)d_ro, (const uint8_t*)d_refs, (const unsigned long long*)d_fo, lm, ln, match, mismatch, score_N, gap_open, gap_ext, filters, d_out);
  } else
  {
    const uint32_t gb = std::min<uint32_t>(n_pairs, (uint32_t)c->n_cu * 8u), stride = 5u * 16u * ((uint32_t)(mx_m + 7) / 8u + 1u);
    uint16_t* d_scr = nullptr;
    if (mode == 4) { IB_GET(d_scr_, uint16_t, (size_t)gb * stride); d_scr = d_scr_; }
    if (mode == 4) hipLaunchKernelGGL(k_ssw_batch<true>, dim3(gb), dim3(64), lds, c->stream, n_pairs, (const uint8_t*)d_reads,
                       (const unsigned long long*)d_ro, (const uint8_t*)d_refs, (const unsigned long long*)d_fo, lm, ln, match, mismatch, score_N, gap_open, gap_ext, filters, -1, d_out, d_scr, stride);
    else hipLaunchKernelGGL(k_ssw_batch<false>, dim3(gb), dim3(64), lds, c->stream, n_pairs, (const uint8_t*)d_reads,
                       (const unsigned long long*)d_ro, (const uint8_t*)d_refs, (const unsigned long long*)d_fo, lm, ln, match, mismatch, score_N, gap_open, gap_ext, filters, mode, d_out, d_scr, stride);
  }
  HIPCHK(c, hipMemcpyAsync(out, d_out, (size_t)n_pairs * 5 * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return SMR_OK;
}

extern "C" int smr_sw_mode(smr_ctx* c, int set_to) {      // set_to: 0 / 1 = select, anything else = query only; returns the mode in use
  if (!c) return SMR_ERR_ARG;
  if (set_to >= 0 && set_to <= 2) c->sw_mode = set_to;
  return c->sw_mode;
}

// rounds the candidate walk of the next part runs per pass (smr_walk.hpp; adapts to what the previous part needed unless SMR_WALK_ROUNDS fixes it)
extern "C" int smr_walk_rounds(const smr_ctx* c, uint32_t out[3]) {
  if (!c || !out) return SMR_ERR_ARG;
  for (int p = 0; p < 3; p++) out[p] = (!c->walk_rounds_fixed && c->walk_need[p]) ? std::min(c->walk_rounds, c->walk_need[p]) : c->walk_rounds;
  return SMR_OK;
}
