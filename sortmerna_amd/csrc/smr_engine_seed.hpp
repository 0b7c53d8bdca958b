// smr_engine_seed.hpp -- host side of the seed stage (included by smr_engine.hip inside its anonymous namespace): scratch, the tuple sort of one (strand, pass),
// the shared sort for the index parts of a batch, the searches, launch_seed.
// (one translation unit: no include guard games -- this file is text of smr_engine.hip, cut out along its stages)

int ensure_seed_bufs(smr_ctx* c, const DParams& P) {
  uint32_t mw = 1;
  for (int p = 0; p < 3; p++) mw = std::max(mw, num_windows(c->b->max_len, P.lnwin, P.skip[p]));
  const uint64_t slots = (uint64_t)std::max(c->b->n, 1u) * mw;
  const uint32_t nk = 2u << P.lnwin;                      // 2 x 4^(L/2) bins: forward and reverse keys
  if (2 * slots >= 0xFFFFFF00ull) { set_err(c, "batch too large for the seed stage (reads x windows >= 2^31): use smaller batches"); return SMR_ERR_CAPACITY; }
  int rc;
  if (c->sb_nk < nk) {
    if ((rc = dev_alloc(c, &c->sb.chist, (size_t)4096 + 1))) return rc;
    if ((rc = dev_alloc(c, &c->sb.cbase, (size_t)4096 + 2))) return rc;
    if ((rc = dev_alloc(c, &c->sb.hpre, (size_t)4096 + 2))) return rc;
    if ((rc = dev_alloc(c, &c->sb.hlist, (size_t)4096 + 2))) return rc;
    if (!c->sb.rows && (rc = dev_alloc(c, &c->sb.rows, (size_t)SEED_KEY_BLOCKS * 4096))) return rc;
    if (!c->sb.bcnt && (rc = dev_alloc(c, &c->sb.bcnt, (size_t)SEED_KEY_BLOCKS))) return rc;
    if (!c->sb.redo && (rc = dev_alloc(c, &c->sb.redo, SEED_REDO_CAP))) return rc;
    if (!c->sb.sn && (rc = dev_alloc(c, &c->sb.sn, SN_COUNT))) return rc;
    if ((rc = dev_alloc(c, &c->sb.emap, (size_t)(nk / 2) / 16 + 1))) return rc;
    c->sb_nk = nk;
  }
  if (c->sb_slots < slots) {
    // a forward and a reverse tuple per window; tmp is cut into one region per block of k_seed_keys (the slots of its reads)
    if ((rc = dev_alloc(c, &c->sb.tmp, 2 * slots))) return rc;
    if ((rc = dev_alloc(c, &c->sb.mid, 2 * slots))) return rc;
    if ((rc = dev_alloc(c, &c->sb.srt, 2 * slots))) return rc;
    for (int d = 0; d < 2; d++) {
      if ((rc = dev_alloc(c, &c->sb.wseg[d], slots))) return rc;
      if ((rc = dev_alloc(c, &c->sb.fbits[d], slots / 32 + 2))) return rc;
    }
    if ((rc = dev_alloc(c, &c->sb.wbin, 2 * slots / 64 + 2))) return rc;
    if ((rc = dev_alloc(c, &c->sb.zbits, slots / 32 + 2))) return rc;
    if ((rc = dev_alloc(c, &c->sb.gflag, slots / 2048 + 2))) return rc;
    // skewed batches: a coarse bin of at least SEED_HOT_BIN_MIN tuples in sub-ranges of SEED_HOT_SUB (their fine histograms); the pieces of hot keys (>= 1024 tuples each)
    c->sb.hbin_min = getenv("SMR_SEED_HOT_BIN") ? (uint32_t)std::max(1, atoi(getenv("SMR_SEED_HOT_BIN"))) : SEED_HOT_BIN_MIN;
    c->sb.hsub = getenv("SMR_SEED_HOT_SUB") ? (uint32_t)std::max(1, atoi(getenv("SMR_SEED_HOT_SUB"))) : SEED_HOT_SUB;
    c->sb.cap_hent = (uint32_t)(2 * slots / c->sb.hsub + 2 * slots / c->sb.hbin_min + 2);
    if ((rc = dev_alloc(c, &c->sb.hh, (size_t)c->sb.cap_hent * 512))) return rc;
    c->sb.cap_pieces = (uint32_t)std::min<uint64_t>(2 * slots / std::max(c->hot_min, 64u) + 2 * slots / SEED_DD_PIECE + 16, 1u << 26);
    if ((rc = dev_alloc(c, &c->sb.pieces, (size_t)c->sb.cap_pieces))) return rc;
    c->sb_slots = slots;
  }
  c->sb.hot_min = c->hot_min;
  c->sb.nk = nk; c->sb.nkh = nk / 2;
  c->sb.fb = std::min<uint32_t>(9, P.lnwin); c->sb.nc = nk >> c->sb.fb;       // L <= 20: at most 4096 coarse bins
  c->sb.cb = 2 * P.partialwin; c->sb.kbits = P.lnwin + 1;
  return SMR_OK;
}

// The longest hit list ONE half-seed search can leave: the strings T of pw + 1 chars that lev1_entry (smr_seed.hpp) accepts for a pattern P number
// at most 31 pw - 20 (104, 135, 166 for pw = 4, 5, 6 over every P; 197 ... 290 for pw = 7 ... 10 on the patterns that reach the maximum and on
// sampled ones: tests/test_lev_closed_form.py), each at most one id.  k_seed_pg needs no capacity per search (its lists lie back to back in the
// wave's candidate budget); k_seed_search -- the DFS kernel: overflow redo, exact-counter mode -- has lane-local lists of hcap entries, and its
// reverse search starts from the forward list (twice the bound).
#define SEED_HCAP_BOUND(pw) (31u * (pw) - 20u)
// A list of k_seed_search overflowed: the next size.  4, 8, ... 128, then the bound, then twice the bound, which no search can exceed -- reaching the
// error below would mean the bound is wrong, not that the data is unusual.  (2 x 290 entries x 64 lanes = 145 KB of the 160 KB of LDS.)
bool grow_hcap(smr_ctx* c, uint32_t pw) {
  const uint32_t bound = SEED_HCAP_BOUND(pw);
  if (c->hcap >= 2 * bound) { set_err(c, "a seed search accepted more strings than the LEV(1) bound allows (internal error)"); return false; }
  c->hcap = c->hcap < 128 ? c->hcap * 2 : c->hcap < bound ? bound : 2 * bound;
  return true;
}

// One sort for several index parts (BASELINE configs[3]: eight --ref; any index cut into parts by -m).  The reference loops (index, part) over the same
// reads (processor.cpp:219-277), and the tuples of a (strand, pass) are the same for every part -- but for the reads that are in the pass (per-part state:
// seed_read_active), the keys the part's lookup table has (a missing mini-trie ends a search before it starts) and the value an ambiguous letter reads as,
// which depends on the read's history in the part (Read::flip34, read.cpp:379-401: the reads with such letters keep a small sort of their own per part).  So
// the first part of a batch that is not its last builds SIX sorted arrays (2 strands x 3 passes: every read long enough, every window, both directions)
// and the searches of every part walk them: keys + two sort passes, 4.1 of a stage's 9.2 ms on the eight-reference workload, once instead of eight times.
// Not with minoccur > 0 (that emit filter needs the part's counts), not in the exact-counter mode, not when a shared array has hot keys (k_seed_dedup
// rewrites tuples in place, and which of several equal tuples can stand for the others depends on the part's active reads): the per-part sort runs then.
// k_seed_keys for this batch (reads per wave trip, lanes per read, reads per block) into sb; returns the instantiation
typedef void (*seed_keys_fn)(DReads, DParams, int, SeedBufs, const RWork*, unsigned long long*, int);
seed_keys_fn seed_keys_setup(smr_ctx* c, SeedBufs& sb, size_t& lds_keys) {
  // reads per wave trip (a power of two; their packed records must fit the wave's LDS stage) and lanes per read; reads per block
  const uint32_t rec_words = (c->b->max_len + 15) / 16 + (c->b->max_len + 31) / 32;
  const bool staged = rec_words <= SEED_STAGE_WORDS;
  uint32_t rwr = 64;
  while (rwr > 1 && (uint64_t)rwr * rec_words > SEED_STAGE_WORDS) rwr >>= 1;
  uint32_t gsh = 0;
  while ((64u >> gsh) > rwr) gsh++;
  sb.g_shift = gsh;
  const uint32_t per_trip = SEED_WAVES * rwr;
  const uint32_t trips = std::max<uint32_t>(1u, (uint32_t)(((uint64_t)sb.n + (uint64_t)per_trip * SEED_KEY_BLOCKS - 1) / ((uint64_t)per_trip * SEED_KEY_BLOCKS)));
  sb.rpb = trips * per_trip;
  sb.kb = std::max<uint32_t>(1u, (sb.n + sb.rpb - 1) / sb.rpb);
  const bool mapped = (sb.nkh / 16) * 4 <= 64 * 1024;
  lds_keys = (size_t)4 * (((sb.nc + 3u) & ~3u) + (mapped ? sb.nkh / 16 : 0u) + (staged ? SEED_WAVES * (SEED_STAGE_WORDS + 8u) : 0u));
  seed_keys_fn kf;
  if (gsh == 0) kf = mapped ? k_seed_keys<true, true, true> : k_seed_keys<true, true, false>;
  else if (staged) kf = mapped ? k_seed_keys<false, true, true> : k_seed_keys<false, true, false>;
  else kf = mapped ? k_seed_keys<false, false, true> : k_seed_keys<false, false, false>;
  if (lds_keys > 64 * 1024) (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_keys);
  return kf;
}

// tuples of one (strand, pass) -> key order: k_seed_keys (mode: smr_seed.hpp) + the two-level sort, into sb.srt / sb.wbin / sb.cbase / sb.sn
int seed_sort(smr_ctx* c, const DevIndex& di, const DParams& P, int pass, SeedBufs& sb, int mode) {
  if ((uint64_t)sb.rpb * sb.maxwin >= (1ull << (64 - sb.kbits - sb.cb))) { set_err(c, "seed stage: a block's windows do not fit the tuple format"); return SMR_ERR_CAPACITY; }
  const uint64_t slots = (uint64_t)sb.n * sb.maxwin;
  const uint32_t gw = std::max<uint32_t>(1u, (uint32_t)((2 * slots + 63) / 64));     // wave chunks of 64 tuples the batch can have at most (every kernel checks its range)
  const bool many_bins = sb.nc > 2048u;                    // (their three tables take 48 KB of the 160)
  const size_t lds_split = (size_t)3 * ((sb.nc + 1u) & ~1u) * 4 + (size_t)(many_bins ? SEED_SPLIT_PIECE_MANY_BINS : SEED_SPLIT_PIECE) * sizeof(SeedTup), lds_bins = (size_t)SEED_PIECE * sizeof(SeedTup);
  if (lds_bins > 60 * 1024 && lds_bins > c->bins_lds_attr) { HIPCHK(c, hipFuncSetAttribute((const void*)k_seed_bins, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bins)); HIPCHK(c, hipFuncSetAttribute((const void*)k_seed_hbins_move, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bins)); c->bins_lds_attr = lds_bins; }      // (per context = per device, like split_lds_attr)
  if (lds_split > 64 * 1024 && lds_split > c->split_lds_attr) {
    HIPCHK(c, hipFuncSetAttribute((const void*)k_seed_split<SEED_SPLIT_PIECE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_split));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_seed_split<SEED_SPLIT_PIECE_MANY_BINS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_split));
    c->split_lds_attr = lds_split;
  }
  size_t lds_keys;
  const seed_keys_fn kf = seed_keys_setup(c, sb, lds_keys);
  ev_mark(c, KP_KEYS);
  HIPCHK(c, hipMemsetAsync(sb.sn, 0, SN_COUNT * 4, c->stream));
  if ((mode & 15) != SEED_KEYS_SHARED) hipLaunchKernelGGL(k_seed_emap, dim3((sb.nkh / 16 + 255) / 256), dim3(256), 0, c->stream, (const uint32_t*)di.lkc, sb.nkh, P.minoccur, sb.emap);
  hipLaunchKernelGGL(kf, dim3(sb.kb), dim3(64 * SEED_WAVES), lds_keys, c->stream, dreads(c), P, pass, sb, (const RWork*)c->b->d_rw, c->b->d_ctr, mode);
  // the two-level sort of the stage's forward and reverse tuples (smr_seed.hpp)
  ev_mark(c, KP_SPLIT);                                    // (with the scans of the block histograms in front of it)
  hipLaunchKernelGGL(k_seed_colscan, dim3((sb.nc + 63) / 64), dim3(1024), 0, c->stream, sb);
  hipLaunchKernelGGL(k_seed_cscan, dim3(1), dim3(1024), 0, c->stream, sb, c->b->d_ctr);
  hipLaunchKernelGGL(k_seed_wbin, dim3((gw + 255) / 256), dim3(256), 0, c->stream, sb);
  if (many_bins) hipLaunchKernelGGL(k_seed_split<SEED_SPLIT_PIECE_MANY_BINS>, dim3(sb.kb), dim3(1024), lds_split, c->stream, sb);
  else hipLaunchKernelGGL(k_seed_split<SEED_SPLIT_PIECE>, dim3(sb.kb), dim3(1024), lds_split, c->stream, sb);
  ev_mark(c, KP_BINS);
  hipLaunchKernelGGL(k_seed_bins, dim3(sb.nc), dim3(1024), lds_bins, c->stream, sb);
  // the coarse bins that are far larger than the others, several blocks each (none on evenly spread keys: three empty launches)
  const uint32_t gh = std::min<uint32_t>(sb.cap_hent, (uint32_t)c->n_cu * 2u);       // (grids that loop: an empty launch should cost a launch, not 2 000 blocks)
  hipLaunchKernelGGL(k_seed_hbins_hist, dim3(gh), dim3(1024), 0, c->stream, sb);
  hipLaunchKernelGGL(k_seed_hbins_scan, dim3(std::min<uint32_t>(sb.nc, 128u)), dim3(1024), 0, c->stream, sb);
  hipLaunchKernelGGL(k_seed_hbins_move, dim3(gh), dim3(1024), lds_bins, c->stream, sb);
  return SMR_OK;
}

// the searches of one sorted array: forward (dir 0) or reverse launch, the overflow redo, the repeated seeds' windows
int seed_search(smr_ctx* c, const DevIndex& di, const DParams& P, int pass, const SeedBufs& sb, int dir, uint32_t pool_words, size_t lds, size_t lds_pg) {
  const uint64_t slots = (uint64_t)sb.n * sb.maxwin;
  const uint32_t gw = std::max<uint32_t>(1u, (uint32_t)((2 * slots + 63) / 64));
  const uint32_t gr = std::min<uint32_t>(gw, SEED_REDO_CAP);
  const uint32_t gp = c->pg_grid ? std::min<uint32_t>((gw + 7u) & ~7u, c->pg_grid) : ((gw + 7u) & ~7u);
  const uint32_t gd = std::min<uint32_t>(sb.cap_pieces, (uint32_t)c->n_cu * 8u);
  HIPCHK(c, hipMemsetAsync(&sb.sn[SN_REDO], 0, 4, c->stream));
  if (dir == 0) {
    hipLaunchKernelGGL(k_seed_pg<0>, dim3(gp), dim3(64), lds_pg, c->stream, dindex(di), P, pass, sb, c->ccap, c->d_pool, pool_words, c->b->d_ctr, c->pg_swz);
    hipLaunchKernelGGL(k_seed_search<0>, dim3(gr), dim3(64), lds, c->stream, dindex(di), P, pass, sb, c->hcap, c->d_pool, pool_words, c->b->d_ctr, (const uint32_t*)sb.redo);
    if (sb.hot_min) hipLaunchKernelGGL(k_seed_prop<0>, dim3(gd), dim3(256), 0, c->stream, sb, c->b->d_ctr);
  } else {
    hipLaunchKernelGGL(k_seed_pg<1>, dim3(gp), dim3(64), lds_pg, c->stream, dindex(di), P, pass, sb, c->ccap, c->d_pool, pool_words, c->b->d_ctr, c->pg_swz);
    hipLaunchKernelGGL(k_seed_search<1>, dim3(gr), dim3(64), lds, c->stream, dindex(di), P, pass, sb, c->hcap, c->d_pool, pool_words, c->b->d_ctr, (const uint32_t*)sb.redo);
    if (sb.hot_min) hipLaunchKernelGGL(k_seed_prop<1>, dim3(gd), dim3(256), 0, c->stream, sb, c->b->d_ctr);
  }
  return SMR_OK;
}

// the six shared arrays of the selected batch (see SharedSort); usable = false when a condition above does not hold
int ensure_shared_sort(smr_ctx* c, const DevIndex& di, const DParams& P) {
  SharedSort& S = *c->shared;
  const bool same = S.batch == c->b && S.gen == c->b->gen && S.lnwin == P.lnwin && S.skip[0] == P.skip[0] && S.skip[1] == P.skip[1] && S.skip[2] == P.skip[2] && S.n == c->b->n;
  if (same) return SMR_OK;
  S.batch = c->b; S.gen = c->b->gen; S.lnwin = P.lnwin; S.n = c->b->n; S.max_len = c->b->max_len;
  for (int q = 0; q < 3; q++) S.skip[q] = P.skip[q];
  S.usable = false;
  for (int s = 0; s < 2; s++) for (int p = 0; p < 3; p++) S.set[s][p].built = false;
  const size_t aw = ((size_t)c->b->n + 255) / 256 * 8 + 4;
  if (S.abits_words < aw) { int rc = dev_alloc(c, &S.abits, aw); if (rc) return rc; S.abits_words = aw; }
  DParams Q = P; Q.minoccur = 0;
  for (int p = 0; p < 3; p++) {
    if (p > 0 && P.skip[p] == P.skip[p - 1]) continue;
    const uint32_t mw = num_windows(c->b->max_len, P.lnwin, P.skip[p]);
    const uint64_t cap = 2ull * std::max(c->b->n, 1u) * mw;
    for (int s = 0; s < 2; s++) {
      SharedSet& T = S.set[s][p];
      int rc;
      if (S.cap[p] < cap || !T.srt) {
        if ((rc = dev_alloc(c, &T.srt, (size_t)cap)) || (rc = dev_alloc(c, &T.wbin, (size_t)(cap / 64 + 2))) ||
            (!T.cbase && ((rc = dev_alloc(c, &T.cbase, (size_t)4096 + 2)) || (rc = dev_alloc(c, &T.sn, (size_t)SN_COUNT))))) {
          // no room for the six arrays (17 GB at 8 M reads; several contexts on one device): every part sorts for itself, as without the option
          (void)hipGetLastError();
          for (int s2 = 0; s2 < 2; s2++) for (int p2 = 0; p2 < 3; p2++) { SharedSet& U = S.set[s2][p2]; dev_free(&U.srt); dev_free(&U.wbin); U.built = false; }
          for (int p2 = 0; p2 < 3; p2++) S.cap[p2] = 0;
          c->seed_shared = 0;
          if (getenv("SMR_VERBOSE")) fprintf(stderr, "libsmr_hip: no device memory for the shared seed sort: every index part sorts for itself\n");
          return SMR_OK;
        }
      }
      SeedBufs sb = c->sb;
      sb.maxwin = T.maxwin = mw; sb.cap_tuples = (uint32_t)cap; sb.n = c->b->n; sb.cap_redo = SEED_REDO_CAP;
      sb.srt = T.srt; sb.wbin = T.wbin; sb.cbase = T.cbase; sb.sn = T.sn; sb.abits = nullptr;
      if ((rc = seed_sort(c, di, Q, p, sb, SEED_KEYS_SHARED | (s << 4)))) return rc;
      T.built = true;
    }
    S.cap[p] = std::max(S.cap[p], cap);
  }
  ev_stop(c);
  // a shared array with hot keys: the per-part sort (with k_seed_dedup) serves such a batch better
  uint32_t hot = 0;
  for (int s = 0; s < 2; s++) for (int p = 0; p < 3; p++) if (S.set[s][p].built) {
    uint32_t v = 0;
    HIPCHK(c, hipMemcpyAsync(&v, S.set[s][p].sn + SN_PIECES, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hot += v;
  }
  S.usable = hot == 0 || c->seed_shared >= 2;
  c->n_seed_shared_builds++;
  if (getenv("SMR_VERBOSE")) fprintf(stderr, "libsmr_hip: one seed sort for the parts of this batch: built (%u pieces of hot keys: %s)\n", hot, S.usable ? "in use" : "not used, every part sorts for itself");
  return SMR_OK;
}

// the seed stage of one (strand, pass): the forward and reverse half-seed searches of all windows (smr_seed.hpp)
int launch_seed(smr_ctx* c, const DevIndex& di, const DParams& P, int pass, bool more_parts = false, int strand = -1) {
  int rc = ensure_seed_bufs(c, P); if (rc) return rc;
  if (c->b->max_len > 0xFFFFu) { set_err(c, "seed stage limit: reads <= 65535 nt"); return SMR_ERR_CAPACITY; }
  if ((uint64_t)di.n_ids + di.n_pos >= 0x7FFFFFF0ull) { set_err(c, "seed stage limit: positions + distinct seeds < 2^31 per index part"); return SMR_ERR_CAPACITY; }
  SeedBufs sb = c->sb;
  sb.maxwin = num_windows(c->b->max_len, P.lnwin, P.skip[pass]);
  const uint64_t slots = (uint64_t)c->b->n * sb.maxwin;
  sb.cap_tuples = (uint32_t)(2 * slots);
  sb.n = c->b->n;
  sb.cap_redo = SEED_REDO_CAP;
  sb.abits = nullptr; sb.inv_maxwin = 1.0 / (double)sb.maxwin;
  sb.seg_inline = (c->pool_words <= (1ull << 30) && (uint64_t)di.n_ids + di.n_pos < (1ull << 30) && !(getenv("SMR_SEG_INLINE") && atoi(getenv("SMR_SEG_INLINE")) == 0)) ? 1u : 0u;
  if (c->seed_exact) sb.hot_min = 0;                         // the exact work counters count every window's search
  const size_t lds = (size_t)SEED_LDS_WORDS(c->hcap) * 4, lds_pg1 = (size_t)PG_LDS_WORDS(c->ccap) * 4;
  const size_t lds_pg = lds_pg1 + (getenv("SMR_PG_LDS_PAD") ? (size_t)atoi(getenv("SMR_PG_LDS_PAD")) : 0);      // (the variable: occupancy experiments)
  // lists of more than 128 hits per search (a crafted neighbourhood: SEED_HCAP_MAX) take more than the default 64 KB of dynamic LDS
  if (lds_pg > 64 * 1024 && lds_pg > c->pg_lds_attr) {
    HIPCHK(c, hipFuncSetAttribute((const void*)k_seed_pg<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pg));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_seed_pg<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pg));
    c->pg_lds_attr = lds_pg;
  }
  if (lds > 64 * 1024 && lds > c->search_lds_attr) {
    HIPCHK(c, hipFuncSetAttribute((const void*)k_seed_search<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(c, hipFuncSetAttribute((const void*)k_seed_search<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    c->search_lds_attr = lds;
  }
  const uint32_t pool_words = (uint32_t)std::min<uint64_t>(c->pool_words, 0x7FFFFFF0ull);
  const uint32_t gw = std::max<uint32_t>(1u, (uint32_t)((2 * slots + 63) / 64));     // wave chunks of 64 tuples the batch can have at most (every kernel checks its range)
  // one sort for several parts?  (strand < 0: smr_seed_scan, the test seam of one (strand, pass) -- always the part's own sort)
  bool shared = false;
  if (strand >= 0 && c->seed_shared && !c->seed_exact && P.minoccur == 0 && (more_parts || c->seed_shared >= 2 || (c->shared->usable && c->shared->batch == c->b && c->shared->gen == c->b->gen))) {
    if ((rc = ensure_shared_sort(c, di, P))) return rc;
    shared = c->shared->usable && c->shared->set[strand][pass].built;
  }
  for (int d = 0; d < 2; d++) HIPCHK(c, hipMemsetAsync(sb.fbits[d], 0, (size_t)(slots / 32 + 2) * 4, c->stream));         // no window has a hit segment yet
  HIPCHK(c, hipMemsetAsync(sb.zbits, 0, (size_t)(slots / 32 + 2) * 4, c->stream));
  HIPCHK(c, hipMemsetAsync(sb.gflag, 0, (size_t)(slots / 2048 + 2) * 4, c->stream));
  // the part's own sort: every read of the (strand, pass) -- or, beside the shared arrays, the reads with ambiguous letters on the reverse strand
  const bool own = !shared || strand == 1;
  if (own) {
    if ((rc = seed_sort(c, di, P, pass, sb, shared ? SEED_KEYS_AMB : SEED_KEYS_ALL))) return rc;
    const uint32_t gd = std::min<uint32_t>(sb.cap_pieces, (uint32_t)c->n_cu * 8u);
    if (sb.hot_min) hipLaunchKernelGGL(k_seed_dedup, dim3(gd), dim3(256), 0, c->stream, sb);
  }
  SeedBufs sh = sb;                                          // the shared array of this (strand, pass), filtered by the reads that are in the launch
  if (shared) {
    const SharedSet& T = c->shared->set[strand][pass];
    sh.srt = T.srt; sh.wbin = T.wbin; sh.cbase = T.cbase; sh.sn = T.sn; sh.cap_tuples = (uint32_t)std::min<uint64_t>(c->shared->cap[pass], 0xFFFFFFFFull);
    sh.abits = c->shared->abits; sh.hot_min = 0;
    c->n_seed_shared++;
    hipLaunchKernelGGL(k_seed_active, dim3((c->b->n + 255u) / 256u), dim3(256), 0, c->stream, c->b->n, pass, strand == 0 ? 1 : 0, (const RWork*)c->b->d_rw, c->shared->abits);
  }
  const uint32_t* no_redo = nullptr;
  if (c->seed_exact) {
    ev_mark(c, KP_PG0);
    hipLaunchKernelGGL(k_seed_search<0>, dim3(gw), dim3(64), lds, c->stream, dindex(di), P, pass, sb, c->hcap, c->d_pool, pool_words, c->b->d_ctr, no_redo);
    ev_mark(c, KP_PG1);
    hipLaunchKernelGGL(k_seed_search<1>, dim3(gw), dim3(64), lds, c->stream, dindex(di), P, pass, sb, c->hcap, c->d_pool, pool_words, c->b->d_ctr, no_redo);
  } else {
    // pigeonhole search; the (rare) waves whose candidate pool overflowed are searched again by the DFS kernel
    for (int dir = 0; dir < 2; dir++) {
      ev_mark(c, dir ? KP_PG1 : KP_PG0);
      if (shared && (rc = seed_search(c, di, P, pass, sh, dir, pool_words, lds, lds_pg))) return rc;
      if (own && (rc = seed_search(c, di, P, pass, sb, dir, pool_words, lds, lds_pg))) return rc;
    }
  }
  if (getenv("SMR_SEED_DEBUG")) {                            // (debug aid: synchronises)
    uint32_t sn[SN_COUNT], hent = 0, sn2[SN_COUNT] = {0};
    HIPCHK(c, hipMemcpyAsync(sn, sb.sn, sizeof sn, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&hent, sb.hpre + sb.nc, 4, hipMemcpyDeviceToHost, c->stream));
    if (shared) HIPCHK(c, hipMemcpyAsync(sn2, sh.sn, sizeof sn2, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    fprintf(stderr, "libsmr_hip: seed stage pass %d: %u tuples (%u forward), %u sub-ranges of large coarse bins, %u pieces of hot keys (from %u tuples per key), %u redo waves; shared sort %s (%u tuples)\n",
            pass, sn[SN_TUPLES], sn[SN_FWD], hent, sn[SN_PIECES], sb.hot_min, sn[SN_REDO], shared ? "in use" : "no", sn2[SN_TUPLES]);
  }
  ev_mark(c, KP_FINISH);
  hipLaunchKernelGGL(k_seed_finish, dim3((c->b->n + 255) / 256), dim3(256), 0, c->stream, dreads(c), P, pass, sb, c->b->d_work, c->b->d_rw, c->d_pool, pool_words, c->b->d_ctr);
  ev_stop(c);
  HIPCHK(c, hipGetLastError());
  return SMR_OK;
}

