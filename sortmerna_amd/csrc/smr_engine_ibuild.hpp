// smr_engine_ibuild.hpp -- the device index build on the host side (kernels in smr_ibuild.hpp; included by smr_engine.hip): device buffer pool, scans,
// the radix sort driver, ib_part_device, smr_index_build_gpu.
// (one translation unit: no include guard games -- this file is text of smr_engine.hip, cut out along its stages)

namespace {
struct DevPool {                         // device buffers of one build; freed together
  std::vector<void*> ptrs;
  ~DevPool() { for (void* p : ptrs) (void)hipFree(p); }
  template <class T> T* get(smr_ctx* c, size_t count) {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) { set_err(c, "hipMalloc failed in the index build"); return nullptr; }
    ptrs.push_back(p);
    return (T*)p;
  }
};
#define IB_GET(var, type, count) type* var = pool.get<type>(c, (count)); if (!var) return SMR_ERR_DEVICE

template <class T> int dev_scan(smr_ctx* c, DevPool& pool, const T* in, T* out, uint64_t n, T* total) {
  const uint64_t tiles = (n + 2047) / 2048;
  if (n == 0) { if (total) *total = 0; return SMR_OK; }
  IB_GET(sums, T, tiles);
  hipLaunchKernelGGL(smr::k_scan_tile<T>, dim3((uint32_t)tiles), dim3(256), 0, c->stream, in, out, sums, (smr::u64)n);
  if (tiles == 1) {
    if (total) { HIPCHK(c, hipMemcpyAsync(total, sums, sizeof(T), hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
    return SMR_OK;
  }
  IB_GET(pre, T, tiles);
  int rc = dev_scan<T>(c, pool, sums, pre, tiles, total);
  if (rc) return rc;
  hipLaunchKernelGGL(smr::k_scan_add<T>, dim3((uint32_t)tiles), dim3(256), 0, c->stream, out, (const T*)pre, (smr::u64)n);
  return SMR_OK;
}

// stable LSD radix sort of bits [lo, hi) ; the sorted data end up in ka / va (the buffers are swapped as needed)
int dev_radix_sort(smr_ctx* c, DevPool& pool, smr::u64*& ka, smr::u64*& kb, uint32_t*& va, uint32_t*& vb, uint64_t n, int lo, int hi) {
  if (n == 0) return SMR_OK;
  const uint32_t tiles = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
  IB_GET(hist, uint32_t, (size_t)256 * tiles);
  IB_GET(offs, uint32_t, (size_t)256 * tiles);
  for (int shift = lo; shift < hi; shift += 8) {
    hipLaunchKernelGGL(smr::k_rs_hist, dim3(tiles), dim3(64), 0, c->stream, (const smr::u64*)ka, (smr::u64)n, shift, hist, tiles);
    int rc = dev_scan<uint32_t>(c, pool, hist, offs, (uint64_t)256 * tiles, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(smr::k_rs_scatter, dim3(tiles), dim3(64), 0, c->stream, (const smr::u64*)ka, (const uint32_t*)va, kb, vb, (smr::u64)n, shift, (const uint32_t*)offs, tiles);
    std::swap(ka, kb); std::swap(va, vb);
  }
  return SMR_OK;
}

int ib_part_device(void* user, const smr::IBuildInput& in, smr_index& ix, std::string& why) {
  smr_ctx* c = (smr_ctx*)user;
  auto fail = [&](int rc) { why = c->err; return rc; };
  (void)hipSetDevice(c->device);
  DevPool pool;
  const uint32_t L = in.L, P = L / 2, W = L + 1, T = P + 1, NK = 1u << L;
  std::vector<uint64_t> occ_start((size_t)in.n_seqs + 1, 0);
  for (size_t m = 0; m < in.n_seqs; m++) occ_start[m + 1] = occ_start[m] + (in.seq_off[m + 1] - in.seq_off[m] - W + 1);
  const uint64_t N = occ_start.back();
  int occbits = 1; while ((1ull << occbits) < N) occbits++;
  if ((int)(2 * L) + occbits > 64 || N >= 0xFFFFFFF0ull) { why = "part too large for the builder (reduce -m)"; return SMR_ERR_ARG; }
  auto run = [&]() -> int {
    const uint64_t ncodes = in.seq_off[in.n_seqs];
    IB_GET(d_codes, uint8_t, ncodes + 1);
    IB_GET(d_seq_off, smr::u64, (size_t)in.n_seqs + 1);
    IB_GET(d_occ_start, smr::u64, (size_t)in.n_seqs + 1);
    HIPCHK(c, hipMemcpyAsync(d_codes, in.codes, ncodes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_seq_off, in.seq_off, ((size_t)in.n_seqs + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_occ_start, occ_start.data(), ((size_t)in.n_seqs + 1) * 8, hipMemcpyHostToDevice, c->stream));
    smr::IBuildDev B; B.codes = d_codes; B.seq_off = d_seq_off; B.occ_start = d_occ_start; B.n_seqs = in.n_seqs;
    B.L = L; B.P = P; B.W = W; B.T = T; B.occbits = (uint32_t)occbits; B.max_pos = in.max_pos; B.N = N;
    const uint32_t gN = (uint32_t)((N + 255) / 256);
    IB_GET(k0, smr::u64, N); IB_GET(k1, smr::u64, N);
    IB_GET(d_last, uint8_t, N);
    hipLaunchKernelGGL(smr::k_ib_keys, dim3(gN), dim3(256), 0, c->stream, B, k0, d_last);
    uint32_t* nov = nullptr; uint32_t* nov2 = nullptr;
    int rc = dev_radix_sort(c, pool, k0, k1, nov, nov2, N, occbits, occbits + 2 * (int)L);
    if (rc) return rc;
    // ids and groups
    IB_GET(d_flag, uint32_t, N); IB_GET(d_excl, uint32_t, N);
    hipLaunchKernelGGL(smr::k_ib_flags, dim3(gN), dim3(256), 0, c->stream, (const smr::u64*)k0, (smr::u64)N, (uint32_t)occbits, d_flag);
    uint32_t n_ids = 0;
    rc = dev_scan<uint32_t>(c, pool, d_flag, d_excl, N, &n_ids);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    IB_GET(d_gstart, uint32_t, (size_t)n_ids + 1); IB_GET(d_present, uint32_t, (size_t)n_ids + 1);
    HIPCHK(c, hipMemsetAsync(d_present, 0, ((size_t)n_ids + 1) * 4, c->stream));
    const uint32_t N32 = (uint32_t)N;
    HIPCHK(c, hipMemcpyAsync(d_gstart + n_ids, &N32, 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(smr::k_ib_groups, dim3(gN), dim3(256), 0, c->stream, (const smr::u64*)k0, (const uint8_t*)d_last, (const uint32_t*)d_flag, (const uint32_t*)d_excl,
                       (smr::u64)N, (uint32_t)occbits, d_gstart, d_present);
    const uint32_t gI = (n_ids + 255) / 256;
    IB_GET(d_pcount, uint32_t, (size_t)n_ids + 1); IB_GET(d_ecount, uint32_t, (size_t)n_ids + 1);
    IB_GET(d_pos_off, uint32_t, (size_t)n_ids + 1); IB_GET(d_ent_off, uint32_t, (size_t)n_ids + 1);
    HIPCHK(c, hipMemsetAsync(d_pcount + n_ids, 0, 4, c->stream)); HIPCHK(c, hipMemsetAsync(d_ecount + n_ids, 0, 4, c->stream));
    hipLaunchKernelGGL(smr::k_ib_counts, dim3(gI), dim3(256), 0, c->stream, (const uint32_t*)d_gstart, (const uint32_t*)d_present, n_ids, in.max_pos, d_pcount, d_ecount);
    uint32_t n_pos = 0, M = 0;
    rc = dev_scan<uint32_t>(c, pool, d_pcount, d_pos_off, (uint64_t)n_ids + 1, &n_pos); if (rc) return rc;
    rc = dev_scan<uint32_t>(c, pool, d_ecount, d_ent_off, (uint64_t)n_ids + 1, &M); if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    IB_GET(d_pos_arr, uint32_t, (size_t)2 * n_pos);
    hipLaunchKernelGGL(smr::k_ib_positions, dim3(gN), dim3(256), 0, c->stream, B, (const smr::u64*)k0, (const uint32_t*)d_flag, (const uint32_t*)d_excl,
                       (const uint32_t*)d_gstart, (const uint32_t*)d_pcount, (const uint32_t*)d_pos_off, d_pos_arr);
    // entries
    IB_GET(d_fkey, uint32_t, M); IB_GET(d_ftail, smr::u64, M); IB_GET(d_rtail, smr::u64, M);
    IB_GET(r0, smr::u64, M); IB_GET(r1, smr::u64, M); IB_GET(rv0, uint32_t, M); IB_GET(rv1, uint32_t, M);
    hipLaunchKernelGGL(smr::k_ib_entries, dim3(gI), dim3(256), 0, c->stream, B, (const smr::u64*)k0, (const uint32_t*)d_gstart, (const uint32_t*)d_present,
                       (const uint32_t*)d_ent_off, n_ids, d_fkey, d_ftail, r0, rv0);
    rc = dev_radix_sort(c, pool, r0, r1, rv0, rv1, M, 0, 2 * (int)(P + T));
    if (rc) return rc;
    IB_GET(d_cntF, uint32_t, (size_t)NK + 1); IB_GET(d_cntR, uint32_t, (size_t)NK + 1);
    IB_GET(d_fstart, uint32_t, (size_t)NK + 1); IB_GET(d_rstart, uint32_t, (size_t)NK + 1);
    HIPCHK(c, hipMemsetAsync(d_cntF, 0, ((size_t)NK + 1) * 4, c->stream)); HIPCHK(c, hipMemsetAsync(d_cntR, 0, ((size_t)NK + 1) * 4, c->stream));
    hipLaunchKernelGGL(smr::k_ib_rfinal, dim3((M + 255) / 256), dim3(256), 0, c->stream, (const smr::u64*)r0, (const uint32_t*)rv0, M, 2 * T, d_rtail, d_cntR,
                       (const uint32_t*)d_fkey, d_cntF);
    rc = dev_scan<uint32_t>(c, pool, d_cntF, d_fstart, (uint64_t)NK + 1, nullptr); if (rc) return rc;
    rc = dev_scan<uint32_t>(c, pool, d_cntR, d_rstart, (uint64_t)NK + 1, nullptr); if (rc) return rc;
    // mini-tries
    const int burst_depth = (int)(W - P - 3);
    IB_GET(d_size, smr::u64, (size_t)2 * NK + 1); IB_GET(d_toff, smr::u64, (size_t)2 * NK + 1);
    IB_GET(d_nodes, uint32_t, (size_t)2 * NK); IB_GET(d_buckets, uint32_t, (size_t)2 * NK); IB_GET(d_status, uint32_t, 1);
    HIPCHK(c, hipMemsetAsync(d_status, 0, 4, c->stream)); HIPCHK(c, hipMemsetAsync(d_size + 2 * (size_t)NK, 0, 8, c->stream));
    const uint32_t gT = (2 * NK + 255) / 256;
    hipLaunchKernelGGL(smr::k_ib_sizes, dim3(gT), dim3(256), 0, c->stream, (const smr::u64*)d_ftail, (const smr::u64*)d_rtail, (const uint32_t*)d_fstart, (const uint32_t*)d_rstart,
                       NK, (int)T, burst_depth, d_size, d_nodes, d_buckets, d_status);
    smr::u64 words = 0;
    rc = dev_scan<smr::u64>(c, pool, d_size, d_toff, (uint64_t)2 * NK + 1, &words); if (rc) return rc;
    uint32_t status = 0;
    HIPCHK(c, hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (status != smr::TRIE_OK) { set_err(c, status == smr::TRIE_ERR_BUCKET ? "bucket with more than 255 entries" : "mini-trie larger than 2^22 words"); return SMR_ERR_IO; }
    if (words > 0xFFFFFFF0ull) { set_err(c, "trie arena exceeds 2^32 words"); return SMR_ERR_IO; }
    IB_GET(d_trie, uint32_t, words); IB_GET(d_lookup, smr::Lookup, NK);
    hipLaunchKernelGGL(smr::k_ib_emit, dim3(gT), dim3(256), 0, c->stream, (const smr::u64*)d_ftail, (const smr::u64*)d_rtail, (const uint32_t*)d_fstart, (const uint32_t*)d_rstart,
                       NK, (int)T, burst_depth, (const smr::u64*)d_toff, d_trie, d_lookup);
    // back to the host object
    reserve_huge(ix.trie, words); reserve_huge(ix.pos_arr, (size_t)2 * n_pos);      // (2 MB pages for the two GB-sized arrays that the copies below fill)
    ix.lookup.resize(NK); ix.trie.resize(words); ix.pos_off.resize((size_t)n_ids + 1); ix.pos_arr.resize((size_t)2 * n_pos);
    std::vector<uint32_t> hn((size_t)2 * NK), hb((size_t)2 * NK);
    HIPCHK(c, hipMemcpyAsync(ix.lookup.data(), d_lookup, (size_t)NK * sizeof(smr::Lookup), hipMemcpyDeviceToHost, c->stream));
    if (words) HIPCHK(c, hipMemcpyAsync(ix.trie.data(), d_trie, (size_t)words * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(ix.pos_off.data(), d_pos_off, ((size_t)n_ids + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    if (n_pos) HIPCHK(c, hipMemcpyAsync(ix.pos_arr.data(), d_pos_arr, (size_t)2 * n_pos * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hn.data(), d_nodes, hn.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hb.data(), d_buckets, hb.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < hn.size(); i++) { ix.n_nodes += hn[i]; ix.n_buckets += hb[i]; }
    ix.n_entries += 2ull * M;
    return SMR_OK;
  };
  const int rc = run();
  if (rc != SMR_OK) return fail(rc);
  return SMR_OK;
}
}  // namespace

extern "C" int smr_index_build_gpu(smr_ctx* c, const char* ref_fasta, uint32_t L, double max_mb, uint32_t max_pos,
                                   smr_index** parts_out, uint32_t cap_parts, uint32_t* n_parts_out, char* err, size_t errcap) {
  if (!c) return SMR_ERR_ARG;
  return smr_index_build_with(ref_fasta, L, max_mb, max_pos, 0, ib_part_device, c, parts_out, cap_parts, n_parts_out, err, errcap);
}
