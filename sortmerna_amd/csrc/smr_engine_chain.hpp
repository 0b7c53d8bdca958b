// smr_engine_chain.hpp -- host side of the candidate stage (included by smr_engine.hip inside its anonymous namespace): LDS budgets, scratch of the long-read strips,
// the rounds of the candidate walk and how their number adapts, launch_chain.
// (one translation unit: no include guard games -- this file is text of smr_engine.hip, cut out along its stages)

uint32_t chain_edges(const DParams& P, uint32_t len) { return P.is_as_percent ? (uint32_t)((P.edges / 100.0) * len) + 1 : (uint32_t)std::max(P.edges, 0); }
// ml / rf: the longest read / reference window of the batch (rounded to 16); rq: the longest window of a read of <= SW_X4_MAX_ROWS letters
void chain_lds(const smr_ctx* c, const DParams& P, uint32_t& ml, uint32_t& rf, uint32_t& rq, size_t& bytes) {
  const uint32_t mq_len = std::min<uint32_t>(c->b->max_len, SW_X4_MAX_ROWS);
  ml = (c->b->max_len + 15) & ~15u;
  rf = (c->b->max_len + 2 * chain_edges(P, c->b->max_len) + 16 + 15) & ~15u;
  rq = std::min(rf, (mq_len + 2 * chain_edges(P, mq_len) + 16 + 15) & ~15u);
  bytes = 5 * (size_t)std::min<uint32_t>(ml, SW_X4_MAX_ROWS) + (size_t)9 * rq + (size_t)CH_KEYS_LDS * 8 + (size_t)std::max<uint32_t>(4u * CH_PAIRS_LDS, 2u * c->chain_scap) * 4 +
          (size_t)CH_HITS_LDS * 8 + (size_t)(CH_HITS_LDS + 8) * 4 + (size_t)c->chain_scap * 4;
}
void chain_lds(const smr_ctx* c, const DParams& P, uint32_t& ml, uint32_t& rf, size_t& bytes) { uint32_t rq; chain_lds(c, P, ml, rf, rq, bytes); }
// per block: strip-boundary rows of the Smith-Waterman kernels (2 ints per reference column) and the letters of the read being walked
// (1 byte each) -- only batches with reads of more than one strip
int ensure_bound(smr_ctx* c, uint32_t blocks, uint32_t rf) {
  if (c->b->max_len <= SW_X4_MAX_ROWS) return SMR_OK;
  const size_t need = (size_t)blocks * 2 * rf, need_rd = (size_t)blocks * ((c->b->max_len + 15) & ~15u);
  if (c->bound_cap < need) { int rc = dev_alloc(c, &c->d_bound, need); if (rc) return rc; c->bound_cap = need; }
  if (c->rdq_cap < need_rd) { int rc = dev_alloc(c, &c->d_rdq, need_rd); if (rc) return rc; c->rdq_cap = need_rd; }
  return SMR_OK;
}

__global__ void k_wstat(const unsigned long long* __restrict__ wctr, unsigned long long* __restrict__ out, uint32_t rounds) {
  if (threadIdx.x < 32) out[threadIdx.x] = threadIdx.x < rounds ? wctr[(size_t)threadIdx.x * WC_STRIDE + WC_NLIST] : 0ull;
}
// After a part (the stream is idle): how many rounds its (strand, pass) launches needed -- the last round that listed more reads than the
// closing round takes in its stride, + that closing round; when the closing round itself was that full, two more next time.  Whatever the
// number, the closing round ends every listed read's pass: the records do not depend on it (WALK_VARIANTS of the parity tests).
int adapt_walk_rounds(smr_ctx* c) {
  if (c->walk_rounds_fixed || !c->wstat_n || !c->d_wstat) return SMR_OK;
  unsigned long long h[8 * 32];
  HIPCHK(c, hipMemcpyAsync(h, c->d_wstat, (size_t)c->wstat_n * 32 * 8, hipMemcpyDeviceToHost, c->stream));       // (on the context's stream, like read_ctr: no other stream is waited for)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const unsigned long long few = (unsigned long long)c->n_cu * 8ull;
  uint32_t need[3] = {0, 0, 0};
  for (uint32_t e = 0; e < c->wstat_n; e++) {
    uint32_t want = 2;
    for (uint32_t r = 0; r < c->wstat_rm[e]; r++) if (h[e * 32 + r] > few) want = r + 2 + (r + 1 == c->wstat_rm[e] ? 2u : 0u);
    need[c->wstat_pass[e]] = std::max(need[c->wstat_pass[e]], want);
  }
  // more rounds at once; fewer by half the difference per part (parts of one run differ: eight databases, batches of a mixed sample)
  for (int p = 0; p < 3; p++) if (need[p]) {
    const uint32_t prev = c->walk_need[p] ? c->walk_need[p] : c->walk_rounds;
    c->walk_need[p] = std::min(c->walk_rounds, need[p] >= prev ? need[p] : prev - std::max(1u, (prev - need[p]) / 2u));
  }
  c->wstat_n = 0;
  return SMR_OK;
}

int launch_chain(smr_ctx* c, const DevIndex& di, const DParams& P, int pass, int is_last_strand) {
  uint32_t ml, rf, rq; size_t lds;
  chain_lds(c, P, ml, rf, rq, lds);
  HIPCHK(c, hipMemsetAsync(&c->b->d_ctr[C_WORK_NEXT], 0, 8, c->stream));
  if (lds > 64 * 1024 && lds > c->chain_lds_attr) {     // reads beyond ~5.6 kb: more than the default 64 KB of dynamic LDS per workgroup (gfx950 has 160 KB per CU)
    HIPCHK(c, hipFuncSetAttribute((const void*)k_chain<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_chain<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_chain<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_chain<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_chain<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_chain<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_chain<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_chain<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    c->chain_lds_attr = lds;
  }
  uint32_t blocks = std::min<uint32_t>(c->chain_blocks, std::max(c->b->n, 1u));
  { int rc = ensure_bound(c, c->chain_blocks, rf); if (rc) return rc; }
  int* const gb = c->b->max_len > SW_X4_MAX_ROWS ? c->d_bound : nullptr;
  uint8_t* const grd = c->b->max_len > SW_X4_MAX_ROWS ? c->d_rdq : nullptr;
  if (c->handover) {
    // CAND_REC_WORDS words per read of the batch, one slice per block of k_cand
    const size_t want_w = (size_t)((c->b->n + CAND_BLOCK - 1u) / CAND_BLOCK) * CAND_BLOCK * CAND_REC_WORDS;
    if (c->mrec_cap < c->b->n) { int rc = dev_alloc(c, &c->d_mrec, (size_t)c->b->n); if (rc) return rc; c->mrec_cap = c->b->n; }
    if (c->mpool_words < want_w) { int rc = dev_alloc(c, &c->d_mpool, want_w); if (rc) return rc; c->mpool_words = want_w; }
  }
  uint2* const mrec = c->handover ? c->d_mrec : nullptr;
  // the split path takes the marked reads with a record of k_cand whose Smith-Waterman problems fit the packed kernels
  const uint32_t wmq = std::min<uint32_t>(c->b->max_len, WK_MAX_ROWS), wml = (wmq + 15) & ~15u;
  const bool split = c->walk_split && mrec && P.sw_mode >= 1 && sw_pk_fits((int)wmq, (int)rq, P.match, P.mismatch, P.score_N, P.gap_open);
  const uint32_t RMX = c->walk_rounds, WK = c->walk_k;         // RMX: what d_wctr is laid out for; RM: the rounds of this launch
  const uint32_t RM = (!c->walk_rounds_fixed && c->walk_need[pass]) ? std::min(RMX, c->walk_need[pass]) : RMX;
  if (split) {
    const size_t n = c->b->n;
    if (c->walk_cap < n || c->walk_kcap < WK) {
      for (int q = 0; q < 2; q++) {
        int rc;
        if ((rc = dev_alloc(c, &c->d_wlist[q], n)) || (rc = dev_alloc(c, &c->d_wstate[q], n)) || (rc = dev_alloc(c, &c->d_wtask[q], n * WK)) || (rc = dev_alloc(c, &c->d_wres[q], n * WK))) return rc;
      }
      int rc;
      if ((rc = dev_alloc(c, &c->d_wtidx, 2 * n * WK)) || (rc = dev_alloc(c, &c->d_wslow, n))) return rc;
      c->walk_cap = n; c->walk_kcap = WK;
    }
    if (c->walk_rcap < RMX) { int rc = dev_alloc(c, &c->d_wctr, (size_t)(RMX + 2) * WC_STRIDE); if (rc) return rc; if ((rc = dev_alloc(c, &c->d_wstat, (size_t)8 * 32))) return rc; c->walk_rcap = RMX; }
    HIPCHK(c, hipMemsetAsync(c->d_wctr, 0, (size_t)(RMX + 2) * WC_STRIDE * 8, c->stream));
    const size_t lds_w = (size_t)wml + rq;
    if (lds_w > 64 * 1024 && lds_w > c->walk_lds_attr) { HIPCHK(c, hipFuncSetAttribute((const void*)k_walk<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w)); c->walk_lds_attr = lds_w; }
  }
  const size_t n_tix = (size_t)c->walk_cap * c->walk_kcap;     // the second half of d_wtidx: the score-only tasks
  unsigned long long* const n_slow = split ? c->d_wctr + (size_t)(RMX + 1) * WC_STRIDE : nullptr;
  ev_mark(c, KP_CAND);
  // the reads without any candidate reference end their pass in k_cand; k_chain walks the ones it marks
  hipLaunchKernelGGL(k_cand, dim3((c->b->n + CAND_BLOCK - 1u) / CAND_BLOCK), dim3(256), CAND_LDS_BYTES(c->cand_bloom, c->handover), c->stream, dreads(c), dindex(di), P, pass, is_last_strand, c->b->d_work, c->b->d_rw, (const uint32_t*)c->d_pool, c->b->d_marks, c->cand_bloom,
                     mrec, c->d_mpool, c->mpool_words);
  if (split) {
    // rounds of walk -> Smith-Waterman -> next list (smr_walk.hpp); the last round scores in the walk kernel, so every listed read ends its pass here
    ev_mark(c, KP_WNEXT);
    hipLaunchKernelGGL(k_wlist, dim3((c->b->n + 1023u) / 1024u), dim3(1024), 0, c->stream, dreads(c), c->b->d_marks, (const uint2*)mrec, (uint32_t)WK_MAX_ROWS, c->d_wlist[0], c->d_wslow, c->d_wctr, n_slow, getenv("SMR_WALK_DEBUG") ? n_slow + 8 : (unsigned long long*)nullptr, (P.num_seeds >= 2 && c->walk_gather) ? 1 : 0);
    const int swr = wmq <= 104 ? 13 : wmq <= 152 ? 19 : wmq <= 208 ? 26 : 32;
    const uint32_t walk_blocks = (uint32_t)c->n_cu * 4u * SMR_WALK_WAVES_PER_SIMD, sw_blocks = (uint32_t)c->n_cu * 4u * (uint32_t)SW16_WAVES(swr);
    for (uint32_t rnd = 0; rnd < RM; rnd++) {
      const int cur = (int)(rnd & 1u), prv = cur ^ 1;
      unsigned long long* const wc = c->d_wctr + (size_t)rnd * WC_STRIDE;
      const bool fin = rnd + 1 == RM;
      ev_mark(c, KP_WALK);
#define WALK_ARGS dreads(c), dindex(di), P, pass, is_last_strand, c->b->d_work, c->b->d_work_aln, c->b->d_rw, c->b->d_ctr, (const uint2*)mrec, (const uint32_t*)c->d_mpool, (const uint32_t*)c->d_pool, (const uint2*)c->d_wlist[cur], \
                  (const WState*)c->d_wstate[prv], (const WTask*)c->d_wtask[prv], (const uint2*)c->d_wres[prv], c->d_wstate[cur], c->d_wtask[cur], c->d_wtidx, c->d_wtidx + n_tix, wc, WK, (unsigned long long)n_tix, (int)rnd, wml, rq, c->walk_assume
      if (fin) hipLaunchKernelGGL(k_walk<true>, dim3(walk_blocks * 3u / SMR_WALK_WAVES_PER_SIMD), dim3(64), (size_t)wml + rq, c->stream, WALK_ARGS);
      else {
        hipLaunchKernelGGL(k_walk<false>, dim3(walk_blocks), dim3(64), 0, c->stream, WALK_ARGS);
        ev_mark(c, KP_SW16);
#define SW16_ARGS dreads(c), dindex(di), P, (const WTask*)c->d_wtask[cur], (const uint32_t*)c->d_wtidx, (const uint32_t*)(c->d_wtidx + n_tix), (const unsigned long long*)wc, c->d_wres[cur]
        if (swr == 13) hipLaunchKernelGGL(k_sw16<13>, dim3(sw_blocks), dim3(64), 0, c->stream, SW16_ARGS);
        else if (swr == 19) hipLaunchKernelGGL(k_sw16<19>, dim3(sw_blocks), dim3(64), 0, c->stream, SW16_ARGS);
        else if (swr == 26) hipLaunchKernelGGL(k_sw16<26>, dim3(sw_blocks), dim3(64), 0, c->stream, SW16_ARGS);
        else hipLaunchKernelGGL(k_sw16<32>, dim3(sw_blocks), dim3(64), 0, c->stream, SW16_ARGS);
#undef SW16_ARGS
        ev_mark(c, KP_WNEXT);
        hipLaunchKernelGGL(k_wnext, dim3((uint32_t)c->n_cu * 2u), dim3(1024), 0, c->stream, P, is_last_strand, c->b->d_work, c->b->d_rw, c->b->d_ctr, (const uint2*)c->d_wlist[cur], (const WState*)c->d_wstate[cur],
                           (const uint2*)c->d_wres[cur], c->d_wlist[prv], (const unsigned long long*)wc, wc + WC_STRIDE, WK, (unsigned long long)n_tix, (int)rnd);
      }
#undef WALK_ARGS
    }
    if (!c->walk_rounds_fixed && c->wstat_n < 8) {         // the reads listed per round, kept for adapt_walk_rounds
      hipLaunchKernelGGL(k_wstat, dim3(1), dim3(32), 0, c->stream, (const unsigned long long*)c->d_wctr, c->d_wstat + (size_t)c->wstat_n * 32, RM);
      c->wstat_pass[c->wstat_n] = pass; c->wstat_rm[c->wstat_n] = RM; c->wstat_n++;
    }
    if (getenv("SMR_WALK_DEBUG")) {                         // measurement aid: reads listed and tasks left per round, reads left to k_chain
      std::vector<unsigned long long> h((size_t)(RMX + 2) * WC_STRIDE);
      HIPCHK(c, hipMemcpyAsync(h.data(), c->d_wctr, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      { const unsigned long long* q = &h[(size_t)(RMX + 1) * WC_STRIDE];
        fprintf(stderr, "libsmr_hip: walk rounds (pass %d): slow %llu (positions <= 64 / 128 / 256 / 512 / more / > 64 hits: %llu %llu %llu %llu %llu %llu);", pass, q[0], q[8], q[9], q[10], q[11], q[12], q[13]); }
      for (uint32_t rnd = 0; rnd < RM; rnd++) fprintf(stderr, " %llu/%llu+%llu", h[(size_t)rnd * WC_STRIDE + WC_NLIST], h[(size_t)rnd * WC_STRIDE + WC_NTASK], h[(size_t)rnd * WC_STRIDE + WC_NTASK2]);
      fprintf(stderr, "\n");
    }
  }
  ev_mark(c, KP_CHAIN);
#define CHAIN_ARGS(stab, t2) dreads(c), dindex(di), P, pass, is_last_strand, c->b->d_work, c->b->d_work_aln, c->b->d_rw, c->d_pool, c->b->d_ctr, c->d_tuples, c->d_keys, c->d_pairs, \
                             c->d_lis, c->d_hits, c->keys_cap, c->pairs_cap, c->hits_cap, ml, rf, c->chain_scap, stab, t2, rq, gb, grd, c->b->d_marks, (const uint2*)mrec, (const uint32_t*)c->d_mpool, \
                             (const uint32_t*)(split ? c->d_wslow : nullptr), (const unsigned long long*)n_slow
  // (LONG: the batch has reads of more than one Smith-Waterman strip; the short-read instantiation carries none of their state)
  const bool striped = P.sw_mode < 0;                        // (the slow path that reproduces ssw.c's stripe geometry: instantiations of its own)
  if (striped) {
    if (gb) hipLaunchKernelGGL((k_chain<false, true, true>), dim3(blocks), dim3(64), lds, c->stream, CHAIN_ARGS(c->chain_ext ? c->d_stab : nullptr, c->chain_ext ? c->d_tuples2 : nullptr));
    else hipLaunchKernelGGL((k_chain<false, false, true>), dim3(blocks), dim3(64), lds, c->stream, CHAIN_ARGS(c->chain_ext ? c->d_stab : nullptr, c->chain_ext ? c->d_tuples2 : nullptr));
  } else
  if (gb) hipLaunchKernelGGL((k_chain<false, true>), dim3(blocks), dim3(64), lds, c->stream, CHAIN_ARGS(c->chain_ext ? c->d_stab : nullptr, c->chain_ext ? c->d_tuples2 : nullptr));
  else hipLaunchKernelGGL((k_chain<false, false>), dim3(blocks), dim3(64), lds, c->stream, CHAIN_ARGS(c->chain_ext ? c->d_stab : nullptr, c->chain_ext ? c->d_tuples2 : nullptr));
  if (c->chain_ext) {
    // the reads whose candidate set outgrew the LDS table of the first launch: same walk, set in the block's global table
    HIPCHK(c, hipMemsetAsync(&c->b->d_ctr[C_WORK_NEXT], 0, 8, c->stream));
    if (striped) {
      if (gb) hipLaunchKernelGGL((k_chain<true, true, true>), dim3(blocks), dim3(64), lds, c->stream, CHAIN_ARGS(c->d_stab, c->d_tuples2));
      else hipLaunchKernelGGL((k_chain<true, false, true>), dim3(blocks), dim3(64), lds, c->stream, CHAIN_ARGS(c->d_stab, c->d_tuples2));
    } else
    if (gb) hipLaunchKernelGGL((k_chain<true, true>), dim3(blocks), dim3(64), lds, c->stream, CHAIN_ARGS(c->d_stab, c->d_tuples2));
    else hipLaunchKernelGGL((k_chain<true, false>), dim3(blocks), dim3(64), lds, c->stream, CHAIN_ARGS(c->d_stab, c->d_tuples2));
  }
#undef CHAIN_ARGS
  ev_stop(c);
  HIPCHK(c, hipGetLastError());
  return SMR_OK;
}

// fold the sharded work counters into their base slots (and clear the shards, so the vector can be written back);
// C_POOL_CURSOR becomes the largest shard cursor
void fold_shards(std::vector<unsigned long long>& h) {
  for (int s = 0; s < C_NSHARD; s++)
    for (int k = 0; k < C_SHARD_W; k++) {
      if (k < C_SHARD_X) h[C_WINDOWS + k] += h[C_SHARDS + C_SHARD_W * s + k];
      else if (k < C_SHARD_X + C_SHARD_NX) h[C_TUP_F + k - C_SHARD_X] += h[C_SHARDS + C_SHARD_W * s + k];
      h[C_SHARDS + C_SHARD_W * s + k] = 0;
    }
  unsigned long long mx = 0;
  for (int s = 0; s < C_NSHARD; s++) mx = std::max(mx, h[C_PCUR + s * C_PCUR_STRIDE]);
  h[C_POOL_CURSOR] = mx;
}

int read_ctr(smr_ctx* c, std::vector<unsigned long long>& h) {
  h.resize(C_TOTAL);
  HIPCHK(c, hipMemcpyAsync(h.data(), c->b->d_ctr, C_TOTAL * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  fold_shards(h);
  return SMR_OK;
}

