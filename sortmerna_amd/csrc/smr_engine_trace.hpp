// smr_engine_trace.hpp -- host side of the traceback stage (included by smr_engine.hip): collecting the alignments that need a CIGAR, the band ladder over
// k_trace_band / k_trace_wide, smr_traceback, smr_cigar_batch.
// (one translation unit: no include guard games -- this file is text of smr_engine.hip, cut out along its stages)

// collect alignments of (index_num, part) that still need a CIGAR
__global__ void k_trace_collect(uint32_t n, uint32_t slots, const RState* __restrict__ saved, const AlignRec* __restrict__ aln, uint32_t index_num, uint32_t part,
                                uint32_t* __restrict__ tasks, unsigned long long* __restrict__ ctr) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool take = false;
  if (i < n * slots) {
    uint32_t r = i / slots, k = i % slots;
    if (k < saved[r].n_align) { const AlignRec& a = aln[i]; take = !(a.has_cigar || a.index_num != index_num || a.part != part); }
  }
  const uint32_t o = block_append(&ctr[C_TRACE_NEXT], take);
  if (take) tasks[o] = i;
}

// CIGARs for every stored alignment of the selected batch that lacks one and belongs to (p->index_num, p->part), whose reference sequences are di's
static int traceback_core(smr_ctx* c, const DevIndex& di, const smr_params* p) {
  int rc;
  ev_drop(c);
  DParams P = make_dparams(c, di, p);
  c->b->fetched = false;
  const uint64_t ntot = (uint64_t)c->b->n * c->b->slots;
  if (c->tasks_cap < ntot) { if ((rc = dev_alloc(c, &c->d_tasks, 2 * ntot))) return rc; c->tasks_cap = ntot; }     // two lists: in / handed on
  if (c->b->cigar_words == 0) {
    c->b->cigar_words = std::max<uint64_t>(ntot * 16, 1u << 20);
    if (const char* e = getenv("SMR_CIGAR_POOL_WORDS")) c->b->cigar_words = std::max<uint64_t>(strtoull(e, nullptr, 10), 16);   // debugging aid: start small, exercise the regrow
    if ((rc = dev_alloc(c, &c->b->d_cigar, c->b->cigar_words))) return rc;
  }
  uint32_t ml, rf; size_t chain_bytes;
  chain_lds(c, P, ml, rf, chain_bytes);                    // ml / rf: the longest read / reference window an alignment can span (edges as k_chain takes them)
  const uint32_t row_pairs = c->b->max_len / 2 + 1;
  std::vector<unsigned long long> h;
  uint32_t* t_in = c->d_tasks; uint32_t* t_out = c->d_tasks + ntot;
  auto before = [&]() -> int {
    HIPCHK(c, hipMemsetAsync(&c->b->d_ctr[C_ERR_CIGAR], 0, 16, c->stream));     // C_ERR_CIGAR, C_ERR_TRACE
    HIPCHK(c, hipMemsetAsync(&c->b->d_ctr[C_TRACE_DEFER], 0, 8, c->stream));
    ev_mark(c, KP_TRACE);
    return SMR_OK;
  };
  // after a kernel: 0 = all done, 1 = tasks were handed on (n_tasks updated), 2 = the CIGAR pool was too small (grown; start over), < 0 = error
  auto after = [&](uint32_t& n_tasks) -> int {
    ev_stop(c);
    HIPCHK(c, hipGetLastError());
    int r2 = read_ctr(c, h); if (r2) return r2;
    ev_collect(c);
    if (h[C_ERR_TRACE]) { set_err(c, "banded traceback left the band for some alignments (internal error)"); return SMR_ERR_CAPACITY; }
    if (h[C_ERR_CIGAR]) {
      // grow the CIGAR pool, keeping what is already there; the failed claims moved the cursor past the end: back to the old capacity
      const uint64_t w = c->b->cigar_words * 2; uint32_t* nw = nullptr;
      if (w > 0xFFFFFFF0ull) { set_err(c, "CIGAR pool exceeds 2^32 words"); return SMR_ERR_CAPACITY; }
      HIPCHK(c, hipMalloc((void**)&nw, w * 4));
      HIPCHK(c, hipMemcpy(nw, c->b->d_cigar, c->b->cigar_words * 4, hipMemcpyDeviceToDevice));
      (void)hipFree(c->b->d_cigar); c->b->d_cigar = nw;
      const unsigned long long cur = std::min<unsigned long long>(h[C_CIGAR_CURSOR], c->b->cigar_words);
      c->b->cigar_words = w;
      HIPCHK(c, hipMemcpy(&c->b->d_ctr[C_CIGAR_CURSOR], &cur, 8, hipMemcpyHostToDevice));
      return 2;
    }
    n_tasks = (uint32_t)h[C_TRACE_DEFER];
    std::swap(t_in, t_out);
    return n_tasks ? 1 : 0;
  };
  for (int attempt = 0; attempt < 40; attempt++) {
    t_in = c->d_tasks; t_out = c->d_tasks + ntot;
    HIPCHK(c, hipMemsetAsync(&c->b->d_ctr[C_TRACE_NEXT], 0, 8, c->stream));
    hipLaunchKernelGGL(k_trace_collect, dim3((uint32_t)((ntot + 1023) / 1024)), dim3(1024), 0, c->stream, c->b->n, c->b->slots, c->b->d_saved, c->b->d_saved_aln, P.index_num, P.part, t_in, c->b->d_ctr);
    if ((rc = read_ctr(c, h))) return rc;
    uint32_t n_tasks = (uint32_t)h[C_TRACE_NEXT];
    if (n_tasks == 0) return SMR_OK;
    const uint32_t pool_words = (uint32_t)std::min<uint64_t>(c->b->cigar_words, 0xFFFFFFF0ull);
    int st = 1;
    // narrow kernels: 8 then 16 lanes per alignment (bands <= 3, <= 7), everything in LDS
    for (int G = 8; G <= 16 && st == 1; G *= 2) {
      const uint32_t ng = 64u / (uint32_t)G;
      const size_t lds = (size_t)row_pairs * 64 + (size_t)ng * (ml + rf) + (size_t)ng * TR_CIG_STAGE * 4;
      if (lds > 64 * 1024) break;
      const uint32_t blocks = std::min<uint32_t>((n_tasks + ng - 1) / ng, (uint32_t)c->n_cu * 16u);
      if ((rc = before())) return rc;
      if (G == 8) hipLaunchKernelGGL(k_trace_band<8>, dim3(blocks), dim3(64), lds, c->stream, dreads(c), dindex(di), P, (const uint32_t*)t_in, n_tasks, c->b->d_saved_aln,
                                     c->b->d_cigar, pool_words, c->b->d_ctr, t_out, ml, rf, row_pairs);
      else hipLaunchKernelGGL(k_trace_band<16>, dim3(blocks), dim3(64), lds, c->stream, dreads(c), dindex(di), P, (const uint32_t*)t_in, n_tasks, c->b->d_saved_aln,
                              c->b->d_cigar, pool_words, c->b->d_ctr, t_out, ml, rf, row_pairs);
      st = after(n_tasks);
    }
    // wide kernel: one wave per alignment, band caps growing level by level up to a band that covers the whole window
    const uint32_t max_band = 2 * std::max(ml, rf);
    const uint32_t level_band[4] = {31u, 255u, 2047u, max_band};
    for (int level = 0; level < 4 && st == 1; level++) {
      if (level > 0 && level_band[level - 1] >= max_band) break;
      const uint32_t band = std::min(level_band[level], max_band);
      const uint32_t wcap = (2 * band + 1 + 63) & ~63u;
      const bool rows_lds = (size_t)wcap * 8 + TR_CIG_STAGE * 4 <= 64 * 1024 && !getenv("SMR_TRACE_GLOBAL_ROWS");     // (the variable: debugging aid, forces the wide-band variant)
      const uint64_t flags_cap = (uint64_t)std::max(c->b->max_len, 1u) * (wcap / 2);
      const uint64_t per_block = flags_cap + (rows_lds ? 0 : (uint64_t)wcap * 8);
      // (measured on 5 kb reads, k_trace per 50 000-read step: 8 blocks per CU 762 ms; 16: 496; 32: 459: the kernel lives on waves in flight, profiles/r4s10_*)
      static const int tw_bpc = getenv("SMR_TRACE_BPC") ? atoi(getenv("SMR_TRACE_BPC")) : 32;
      // (the tiles take at most 16 GiB and at most a quarter of what is free on the device now: with many resident batches and index parts, or on a
      // smaller device, fewer blocks run instead of the allocation failing)
      size_t mem_free = 0, mem_total = 0;
      if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) mem_free = (size_t)16 << 30;
      const uint64_t budget = std::min<uint64_t>(16ull << 30, std::max<uint64_t>(c->trflags_bytes, (uint64_t)mem_free / 4));
      uint32_t blocks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(budget / per_block, 1), (uint64_t)c->n_cu * tw_bpc);
      blocks = std::min(blocks, n_tasks);
      const size_t lds_tw = (size_t)TR_CIG_STAGE * 4 + (rows_lds ? (size_t)wcap * 8 : 0);
      if (c->trflags_bytes < (uint64_t)blocks * flags_cap) { if ((rc = dev_alloc(c, &c->d_trflags, (size_t)blocks * flags_cap))) return rc; c->trflags_bytes = (uint64_t)blocks * flags_cap; }
      if (!rows_lds && c->trrows_ints < (uint64_t)blocks * 2 * wcap) { if ((rc = dev_alloc(c, &c->d_trrows, (size_t)blocks * 2 * wcap))) return rc; c->trrows_ints = (uint64_t)blocks * 2 * wcap; }
      if ((rc = before())) return rc;
      if (rows_lds) hipLaunchKernelGGL(k_trace_wide<true>, dim3(blocks), dim3(64), lds_tw, c->stream, dreads(c), dindex(di), P, (const uint32_t*)t_in, n_tasks,
                                       c->b->d_saved_aln, c->b->d_cigar, pool_words, c->b->d_ctr, t_out, (int)band, c->d_trflags, (unsigned long long)flags_cap, c->d_trrows, wcap);
      else hipLaunchKernelGGL(k_trace_wide<false>, dim3(blocks), dim3(64), lds_tw, c->stream, dreads(c), dindex(di), P, (const uint32_t*)t_in, n_tasks,
                              c->b->d_saved_aln, c->b->d_cigar, pool_words, c->b->d_ctr, t_out, (int)band, c->d_trflags, (unsigned long long)flags_cap, c->d_trrows, wcap);
      st = after(n_tasks);
    }
    if (st < 0) return st;
    if (st == 0) return SMR_OK;
    if (st == 1) { set_err(c, "banded traceback did not reach the alignment score within the widest band (internal error)"); return SMR_ERR_CAPACITY; }
  }
  set_err(c, "CIGAR pool regrow attempts exhausted");
  return SMR_ERR_CAPACITY;
}

extern "C" int smr_traceback(smr_ctx* c, int slot, const smr_params* p) {
  if (!c || slot < 0 || slot >= 64) return SMR_ERR_ARG;
  if (!c->idx[slot].used || !c->b->d_saved) { set_err(c, "index slot empty or no reads uploaded"); return SMR_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  int rc = check_params(c, p); if (rc) return rc;
  if (c->b->n == 0) return SMR_OK;
  return traceback_core(c, c->idx[slot], p);
}

// The traceback kernels at the ssw.h seam: for n independent (read window, reference window, score) triples what banded_sw returns
// (ssw.c:577-773, called from ssw_align :919-926 with band |refLen - readLen| + 1): a throw-away batch / reference set is put on the
// device, with one stored alignment per pair spanning both windows, and goes through the same host logic and kernels as smr_traceback.
extern "C" int smr_cigar_batch(smr_ctx* c, uint32_t n_pairs, const uint8_t* reads, const uint64_t* read_off, const uint8_t* refs, const uint64_t* ref_off,
                               const uint16_t* scores, int match, int mismatch, int score_N, int gap_open, int gap_ext,
                               uint32_t* cigar_out, uint64_t cigar_cap, uint64_t* cigar_off_out) {
  if (!c || !read_off || !ref_off || !scores || !cigar_off_out) return SMR_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  cigar_off_out[0] = 0;
  if (n_pairs == 0) return SMR_OK;
  smr_params p; smr_params_default(&p);
  p.match = match; p.mismatch = mismatch; p.score_N = score_N; p.gap_open = gap_open; p.gap_ext = gap_ext; p.edges = 0;
  Batch* keep = c->b;
  Batch tmp;
  DevIndex di;
  std::vector<uint32_t> words, lens(n_pairs);
  std::vector<uint64_t> rec_off((size_t)n_pairs + 1, 0);
  std::vector<RState> st(n_pairs);
  std::vector<AlignRec> al(n_pairs);
  uint32_t max_len = 1; uint64_t max_ref = 1;
  for (uint32_t i = 0; i < n_pairs; i++) {
    const uint64_t m = read_off[i + 1] - read_off[i], n = ref_off[i + 1] - ref_off[i];
    if (m == 0 || n == 0 || m > 0xFFFFu) { set_err(c, "smr_cigar_batch: empty or oversized pair"); return SMR_ERR_ARG; }
    const uint32_t cw = (uint32_t)((m + 15) >> 4), mw = (uint32_t)((m + 31) >> 5);
    rec_off[i] = words.size();
    words.resize(words.size() + cw + mw, 0u);
    uint32_t* rec = words.data() + rec_off[i];
    for (uint64_t q = 0; q < m; q++) {
      const uint8_t ch = reads[read_off[i] + q];
      if (ch > 3) rec[cw + (q >> 5)] |= 1u << (q & 31); else rec[q >> 4] |= (uint32_t)ch << ((q & 15) * 2);
    }
    lens[i] = (uint32_t)m; max_len = std::max(max_len, (uint32_t)m); max_ref = std::max(max_ref, n);
    memset(&st[i], 0, sizeof(RState)); st[i].n_align = 1; st[i].is_hit = 1;
    memset(&al[i], 0, sizeof(AlignRec));
    al[i].ref_num = i; al[i].ref_begin1 = 0; al[i].ref_end1 = (int32_t)n - 1; al[i].read_begin1 = 0; al[i].read_end1 = (int32_t)m - 1;
    al[i].readlen = (uint32_t)m; al[i].score1 = scores[i]; al[i].strand = 1;
  }
  rec_off[n_pairs] = words.size();
  p.edges = (int32_t)std::min<uint64_t>(max_ref > max_len ? (max_ref - max_len + 1) / 2 : 0, 0x3FFFFFFF);   // so that the LDS window bound covers the longest reference window
  tmp.n = n_pairs; tmp.max_len = max_len; tmp.slots = 1; tmp.used = true;
  int rc = SMR_OK;
  auto run = [&]() -> int {
    int r2;
    c->b = &tmp;
    if ((r2 = dev_alloc(c, &tmp.d_words, words.size() + 4))) return r2;
    if ((r2 = dev_alloc(c, &tmp.d_rec_off, rec_off.size()))) return r2;
    if ((r2 = dev_alloc(c, &tmp.d_len, lens.size()))) return r2;
    if ((r2 = dev_alloc(c, &tmp.d_saved, (size_t)n_pairs))) return r2;
    if ((r2 = dev_alloc(c, &tmp.d_saved_aln, (size_t)n_pairs))) return r2;
    if ((r2 = dev_alloc(c, &tmp.d_ctr, (size_t)C_TOTAL))) return r2;
    if ((r2 = dev_alloc(c, &di.ref_seq, (size_t)ref_off[n_pairs] + 64))) return r2;
    if ((r2 = dev_alloc(c, &di.ref_off, (size_t)n_pairs + 1))) return r2;
    HIPCHK(c, hipMemcpyAsync(tmp.d_words, words.data(), words.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(tmp.d_rec_off, rec_off.data(), rec_off.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(tmp.d_len, lens.data(), lens.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(tmp.d_saved, st.data(), st.size() * sizeof(RState), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(tmp.d_saved_aln, al.data(), al.size() * sizeof(AlignRec), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(tmp.d_ctr, 0, C_TOTAL * 8, c->stream));
    HIPCHK(c, hipMemcpyAsync(di.ref_seq, refs, (size_t)ref_off[n_pairs], hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(di.ref_off, ref_off, ((size_t)n_pairs + 1) * 8, hipMemcpyHostToDevice, c->stream));
    di.n_refs = n_pairs; di.lnwin = 18; di.used = true;
    if ((r2 = check_params(c, &p, false))) return r2;
    if ((r2 = traceback_core(c, di, &p))) return r2;
    std::vector<unsigned long long> h;
    if ((r2 = read_ctr(c, h))) return r2;
    std::vector<uint32_t> pool((size_t)std::min<uint64_t>(h[C_CIGAR_CURSOR], tmp.cigar_words));
    HIPCHK(c, hipMemcpyAsync(al.data(), tmp.d_saved_aln, al.size() * sizeof(AlignRec), hipMemcpyDeviceToHost, c->stream));
    if (!pool.empty()) HIPCHK(c, hipMemcpyAsync(pool.data(), tmp.d_cigar, pool.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint64_t o = 0;
    for (uint32_t i = 0; i < n_pairs; i++) {
      if (!al[i].has_cigar) { set_err(c, "smr_cigar_batch: an alignment was left without a CIGAR"); return SMR_ERR_STATE; }
      for (uint32_t q = 0; q < al[i].cigar_len; q++, o++) if (cigar_out && o < cigar_cap) cigar_out[o] = pool[(size_t)al[i].cigar_off + q];
      cigar_off_out[i + 1] = o;
    }
    return SMR_OK;
  };
  rc = run();
  (void)hipStreamSynchronize(c->stream);
  c->b = keep;
  dev_free(&tmp.d_words); dev_free(&tmp.d_rec_off); dev_free(&tmp.d_len); dev_free(&tmp.d_saved); dev_free(&tmp.d_saved_aln); dev_free(&tmp.d_ctr); dev_free(&tmp.d_cigar);
  dev_free(&di.ref_seq); dev_free(&di.ref_off);
  return rc;
}

