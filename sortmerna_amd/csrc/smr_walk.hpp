// smr_walk.hpp -- part of the HIP kernels of libsmr_hip (included by smr_kernels.hpp after smr_chain.hpp).
//
// compute_lis_alignment (alignment.cpp:100-509) for the reads k_cand marks, taken apart into ROUNDS of three kernels instead of one persistent
// kernel that walks and scores (k_chain, smr_chain.hpp: 168 VGPRs, 3 waves per SIMD, the walk's scalar control flow and the Smith-Waterman
// systolic arrays in one register allocation).  The candidate walk is a resumable generator of Smith-Waterman tasks, so:
//
//   k_wlist   the marked reads whose positions k_cand left as a record (at most CAND_REC_MAX = 64 of them: 99 % of the marked reads) and that
//             the packed SW kernel takes become the round-0 list; every other marked read stays with k_chain (the `slow` list)
//   k_walk    one wave per listed read, no Smith-Waterman registers: candidate set, candidate order and every candidate's sorted pairs by ONE
//             bitonic sort of the (reference, position, window) triples across the 64 lanes; the walk consumes the results of the tasks the
//             read left in the previous round and, at the first task it has no result for, leaves up to K tasks (that one + the ones the walk
//             reaches next under a prediction) and its Walk state
//   k_sw16    nothing but Smith-Waterman over the dense task list: SIXTEEN problems per wave (a quad of lanes = 8 virtual lanes x R rows,
//             hand-over by quad_perm DPP), no LDS, reference letters fetched one period of four columns ahead
//   k_wnext   reads whose look-ahead ended inside their batch with nothing aligned end their pass here (what the sequential walk does with
//             those results); the others form the next round's list
//
// The last round (k_walk<true>) scores what it still meets in the kernel itself (one problem per wave), so the number of rounds is fixed and no
// host synchronisation sits between them.  Results equal the sequential walk's by construction -- a task's result is looked up by its geometry,
// predictions only decide WHICH windows are scored ahead -- and by test (every parity test runs through this path; SMR_WALK_SPLIT=0 = k_chain only).
#pragma once

namespace smr {

#define WK_MAX 15u                    // most tasks a read leaves per round
// The task arrays of a round hold `cap` tasks (K0 per read of the batch).  A round with few listed reads lets each of them leave more tasks: a
// read that meets a family of a hundred references (real 16S data) gets through its candidates in a handful of rounds, and round 0, where
// most listed reads have one spurious candidate, stays at K0.  Every kernel of a round derives the same number from the round's list length
// (round 0: K0 whatever the list length).
__host__ __device__ __forceinline__ uint32_t walk_tasks_per_read(uint32_t nlist, unsigned long long cap, uint32_t k0, int round) {
  if (round == 0) return k0;
  const unsigned long long q = nlist ? cap / nlist : (unsigned long long)WK_MAX;
  return (uint32_t)(q < k0 ? k0 : (q > WK_MAX ? WK_MAX : q));
}
#define WK_MAX_ROWS 256u              // longest read span k_sw16 takes (8 virtual lanes x 32 rows)
#define WK_MAX_POS 128u               // most positions (seed hits x their occurrences) of a read the round kernels take: two per lane
#ifndef WK_CLAIM
#define WK_CLAIM 24u                                      // list entries a wave claims at a time, at most (24 x WK_MAX task slots = 7 424 bytes of LDS for the kernel; with 32 it was 8 160 and the kernel 6 % slower, profiles/r5s33_*: fewer than the 20 waves per CU its registers allow)
#endif
// per-round counters (u64 words): every hot one on a 128-byte line of its own
enum { WC_NLIST = 0, WC_CLAIM = 16, WC_NTASK = 32, WC_NTASK2 = 48, WC_STRIDE = 64 };

struct WTask {                        // one Smith-Waterman task: read span x reference window (alignment.cpp:271-357)
  uint32_t r, max_ref;
  uint64_t rf_start;                  // where the window starts in ix.ref_seq
  uint32_t ars, head;                 // align_ref_start, head (what the walk needs besides the window when it resumes at this task)
  uint16_t aq, m, nref;               // align_que_start, read span, window length
  uint16_t flags;                     // bit 0: the read is walked on its reverse-complement strand; bit 1: rows and columns run backwards from aq / rf_start (a begin-cell task)
};
struct WState {                       // the walk of a read standing AT the first task it had no result for (= task 0 of the tasks it left)
  uint32_t k, it, ms_lo, ms_hi, begin_ref, begin_read;
  int32_t best;                       // Walk::best there (also what the read ends with when nothing of its tasks aligns)
  uint32_t bits;                      // 0 is_aligned, 1 go_on, 2 started, 3 pending_pop, 4 search, 5 look-ahead ended under "nothing aligns", 6 last result aligned, 8..11 tasks left
  uint32_t cells;                     // DP cells of the tasks left
};
#define WS_NK(bits) (((bits) >> 8) & 15u)

// ------------------------------------------------------------------------------------------------
// k_wlist: marks -> the round-0 list of the split path (marks cleared) and the list of the reads that stay with k_chain
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_wlist(DReads rd, uint8_t* __restrict__ marks, const uint2* __restrict__ mrec, uint32_t max_rows, uint2* __restrict__ list0,
                                                uint32_t* __restrict__ slow, unsigned long long* __restrict__ wc0, unsigned long long* __restrict__ n_slow, unsigned long long* census, int gather) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool mk = i < rd.n && marks[i] == 1;
  // (with a record of k_cand, or with few enough positions for k_walk to gather them itself; num_seeds < 2 makes every reference a candidate: records only)
  const bool fast = mk && mrec && (mrec[i].x != NONE || (mrec[i].y - 1u < WK_MAX_POS && gather)) && rd.len[i] <= max_rows;
  const uint32_t o = block_append(&wc0[WC_NLIST], fast);
  if (fast) { list0[o] = make_uint2(i, NONE); marks[i] = 0; }
  const uint32_t o2 = block_append(n_slow, mk && !fast);
  if (mk && !fast) slow[o2] = i;
  if (census) {                                            // (measurement aid: why the reads left to k_chain have no record -- positions <= 64 but no room in the block's slice, <= 128, <= 256, <= 512, more, too many hits)
    const uint32_t np = (mk && !fast && mrec) ? mrec[i].y : 0u;
    const bool s = mk && !fast;
    const int cls = !s ? -1 : np == 0xFFFFFFFFu ? 5 : np <= 64u ? 0 : np <= 128u ? 1 : np <= 256u ? 2 : np <= 512u ? 3 : 4;
    for (int q = 0; q < 6; q++) { const unsigned long long m = __ballot(cls == q); if (m && (threadIdx.x & 63u) == 0) atomicAdd(&census[q], (unsigned long long)__popcll(m)); }
  }
}

// ------------------------------------------------------------------------------------------------
// Sixteen Smith-Waterman problems per wave.  A QUAD of lanes is one systolic array of 8 virtual lanes (low halves = virtual lanes 0..3, high
// halves 4..7) with R consecutive read rows each: n + 7 steps for n columns instead of the n + 31 of the four-problem kernel, and the ~14
// instructions of hand-over per step are shared by 8 R cells.  Recurrence, representation (Y = H - gap_open, one v_perm_b32 score lookup per
// cell pair) and end-cell rule are those of sw_wave_pk_x4 (smr_sw_pk.hpp), whose results it must equal bit for bit.  No LDS: a lane builds the
// score tables of its rows from the packed read record, and the reference letters enter at lane 0 of the quad from a register that holds the
// four columns of the current period (quad_perm broadcast) while the next period's letters are on their way from memory.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int dpp_quad_ror1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x93 /* quad_perm:[3,0,1,2] */, 0xF, 0xF, false); }
template <int U> __device__ __forceinline__ int dpp_quad_bcast(int v) { return __builtin_amdgcn_update_dpp(v, v, U * 0x55 /* quad_perm:[U,U,U,U] */, 0xF, 0xF, false); }
__device__ __forceinline__ unsigned long long quad_max_u64(unsigned long long v) {
  for (int d = 2; d > 0; d >>= 1) { const unsigned long long o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
  return v;
}

struct __attribute__((aligned(4))) Sw16W3 { uint32_t a, b, c; };       // (4-byte aligned: global_load_dwordx3 / x2 take that)
struct __attribute__((aligned(4))) Sw16W2 { uint32_t a, b; };
// KEYS = false: the score only (no running-maximum key per row: ten instead of thirteen instructions per cell pair, R registers less) -- for
// the tasks of reads that are not expected to align (their result is "score <= minimal_score", nothing else is asked of it; when one aligns
// after all, its end cell is found by k_begins)
template <int R, bool HASN, bool KEYS>
__device__ __forceinline__ SwRes sw_quad16(const uint32_t* __restrict__ rec, uint32_t len, uint32_t reversed, int m, int aq, const uint8_t* __restrict__ ref, int n,
                                           int match, int mismatch, int scoreN, int go, int ge, int dir) {      // dir = 1, or -1: rows and columns run backwards from aq / ref (ssw_align's reverse pass)
  const int gl = lane_id() & 3;
  const pk16 GE = pk_splat(ge), GO = pk_splat(go), ZERO = pk_splat(0);
  const uint32_t TN = (((uint32_t)(scoreN + go)) & 0xFFFFu) * 0x00010001u;
  uint32_t tlo[R], thi[R];
  pk16 Y[R], E[R];
  uint32_t key[KEYS ? R : 1];
  pk16 hmax = ZERO;
  const uint32_t t_mm = ((uint32_t)(mismatch + go) & 0xFFu) * 0x01010101u, t_x = ((uint32_t)(mismatch + go) ^ (uint32_t)(match + go)) & 0xFFu,
                 t_n = ((uint32_t)(scoreN + go) & 0xFFu) * 0x01010101u;
  // The letters of a lane's rows are two runs of R consecutive read positions (low half: rows gl R .., high half: rows (gl + 4) R ..): at most three
  // words of 2-bit codes and two of ambiguity bits per run (R <= 32), and all ten are asked for before the first is looked at -- read_nt per row
  // was two loads under `if (row < m)` each, which the compiler turned into 2 R waits for memory one after the other.
  const uint32_t cw = (len + 15u) >> 4, cwm = max(cw, 1u) - 1u, awm = max((len + 31u) >> 5, 1u) - 1u;
  const int sgn = reversed ? -dir : dir;                  // a row further = this many physical positions further (read.cpp:350-357: the reverse strand is read from the end)
  uint32_t cwd[2][3], awd[2][2];
  int pa[2], cw0[2], aw0[2];
#pragma unroll
  for (int hf = 0; hf < 2; hf++) {
    const int row0 = (gl + 4 * hf) * R, nv = max(1, min(R, m - row0));
    const int k0 = aq + dir * row0;
    pa[hf] = reversed ? (int)len - 1 - k0 : k0;
    const int pmin = max(0, min(pa[hf], pa[hf] + sgn * (nv - 1)));      // the lowest position of the run's rows < m (they lie inside the read)
    cw0[hf] = (int)min((uint32_t)pmin >> 4, cwm); aw0[hf] = (int)min((uint32_t)pmin >> 5, awm);
    // (one 12-byte and one 8-byte load per run: the words behind a record's last one are the next record's or the slack the upload leaves behind the batch)
    const Sw16W3 c3 = *reinterpret_cast<const Sw16W3*>(rec + cw0[hf]);
    const Sw16W2 a2 = *reinterpret_cast<const Sw16W2*>(rec + cw + aw0[hf]);
    cwd[hf][0] = c3.a; cwd[hf][1] = c3.b; cwd[hf][2] = c3.c; awd[hf][0] = a2.a; awd[hf][1] = a2.b;
  }
#pragma unroll
  for (int j = 0; j < R; j++) {
    uint32_t t2[2];
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
      const int row = (gl + 4 * hf) * R + j;
      uint32_t t = 0;
      if (row < m) {
        const int ph = pa[hf] + sgn * j;
        const int wi = (ph >> 4) - cw0[hf], ai = (ph >> 5) - aw0[hf];
        const uint32_t wd = wi == 0 ? cwd[hf][0] : wi == 1 ? cwd[hf][1] : cwd[hf][2];
        const uint32_t ad = ai == 0 ? awd[hf][0] : awd[hf][1];
        uint32_t c = (wd >> ((ph & 15) * 2)) & 3u;
        if (reversed) c = 3u - c;
        if ((ad >> (ph & 31)) & 1u) c = 4u;               // (Read::flip34, read.cpp:379-401: ambiguous letters are 4 for Smith-Waterman)
        t = c == 4u ? t_n : t_mm ^ (t_x << (8u * c));
      }
      t2[hf] = t;
    }
    tlo[j] = t2[0]; thi[j] = t2[1];
    Y[j] = pk_splat(-go); E[j] = ZERO; if (KEYS) key[j] = 0;
  }
  int steps = m > 0 ? n + (m + R - 1) / R - 1 : 0;
  for (int d = 32; d > 0; d >>= 1) steps = max(steps, __shfl_xor(steps, d, 64));
#ifdef SMR_SW16_NOSTEPS                                   // what-if build: set-up and result only (how much of the kernel is not the step loop)
  steps = min(steps, 4);
#endif
  uint32_t lastY = pk_bits(pk_splat(-go)), lastF = 0, selcur = PK_SEL_NONE * 0x00010001u;
  pk16 diag0 = pk_splat(-go);
  const uint32_t in_y = (uint32_t)(-go) & 0xFFFFu;
  uint32_t xlo = ((uint32_t)(0x3FFF + gl) << 1) | 1u;
  const bool first = gl == 0;
  uint32_t wnext = gl < n ? pk_sel_of(ref[dir * gl]) : PK_SEL_NONE;
#define SW16_STEP(U)                                                                                                        \
  {                                                                                                                         \
    const uint32_t win = (uint32_t)dpp_quad_bcast<U>((int)wcur);                                                            \
    const uint32_t rY = (uint32_t)dpp_quad_ror1((int)lastY), rF = (uint32_t)dpp_quad_ror1((int)lastF), rS = (uint32_t)dpp_quad_ror1((int)selcur); \
    const pk16 upY = pk_from(first ? ((rY << 16) | in_y) : rY);                                                             \
    const pk16 upF = pk_from(first ? (rF << 16) : rF);                                                                      \
    selcur = first ? ((rS << 16) | win | 0x00040000u) : rS;                                                                 \
    uint32_t nmask = 0;                                                                                                     \
    if (HASN) nmask = ((selcur >> 8) & 0x00010001u) * 0xFFFFu;                                                              \
    const uint32_t xhi = xlo + 7u;                                                                                          \
    pk16 diag = diag0, uy = upY, uf = upF;                                                                                  \
    _Pragma("unroll") for (int j = 0; j < R; j++) {                                                                         \
      uint32_t T = perm_b32(thi[j], tlo[j], selcur);                                                                        \
      if (HASN) T = (T & ~nmask) | (TN & nmask);                                                                            \
      const pk16 a = pk_add(diag, pk_from(T));                                                                              \
      const pk16 e = pk_max(pk_subs_u(E[j], GE), Y[j]);   /* E is kept clamped at 0 (saturating subtract): e >= 0, so */    \
      const pk16 f = pk_max(pk_sub(uf, GE), uy);                                                                            \
      const pk16 h = pk_max(pk_max(a, e), f);             /* h = max(a, e, f, 0) without the fourth operand */               \
      diag = Y[j];                                                                                                          \
      const pk16 y = pk_sub(h, GO);                                                                                         \
      Y[j] = y; E[j] = e;                                                                                                   \
      if (KEYS) { const uint32_t hu = pk_bits(h); key[j] = max(key[j], max((hu << 16) | xlo, (hu & 0xFFFF0000u) | xhi)); }  \
      else hmax = pk_max(hmax, h);                                                                                          \
      uy = y; uf = f;                                                                                                       \
    }                                                                                                                       \
    diag0 = upY;                                                                                                            \
    lastY = pk_bits(uy); lastF = pk_bits(uf);                                                                               \
    xlo -= 2u;                                                                                                              \
  }
  // (steps beyond a problem's own n + lanes - 1, up to the wave's maximum rounded to a period, meet the selector NONE: values that never reach a maximum)
  for (int t = 0; t < steps; t += 4) {
    const uint32_t wcur = wnext;
    const int cq = t + 4 + gl;
    wnext = cq < n ? pk_sel_of(ref[dir * cq]) : PK_SEL_NONE;
    SW16_STEP(0) SW16_STEP(1) SW16_STEP(2) SW16_STEP(3)
  }
#undef SW16_STEP
  if (!KEYS) {
    const uint32_t hb = pk_bits(hmax);
    int sc = max((int)(hb & 0xFFFFu), (int)(hb >> 16));
    for (int d = 2; d > 0; d >>= 1) sc = max(sc, __shfl_xor(sc, d, 64));
    SwRes rs; rs.score = sc; rs.end_ref = -2; rs.end_read = 0xFFFF;       // (end cell not computed: wres_pack makes y = 0xFFFFFFFF of it)
    return rs;
  }
  int bestH = 0, bestcol = 0x1FFFFF, bestrow = 0x1FFFFF;
#pragma unroll
  for (int j = 0; j < R; j++) {
    const int h = (int)(key[j] >> 16);
    const int hf = (key[j] & 1u) ? 0 : 1;
    const int col = 0x3FFF - (int)((key[j] >> 1) & 0x7FFFu);
    const int row = (gl + 4 * hf) * R + j;
    if (h > bestH || (h == bestH && h > 0 && (col < bestcol || (col == bestcol && row < bestrow)))) { bestH = h; bestcol = col; bestrow = row; }
  }
  unsigned long long k64 = bestH > 0 ? (((unsigned long long)bestH << 42) | ((unsigned long long)(0x1FFFFF - bestcol) << 21) |
                                        (unsigned long long)(0x1FFFFF - bestrow)) : 0ull;
  k64 = quad_max_u64(k64);
  SwRes rr;
  if (k64 == 0) { rr.score = 0; rr.end_ref = -1; rr.end_read = m - 1; return rr; }
  rr.score = (int)(k64 >> 42);
  rr.end_ref = 0x1FFFFF - (int)((k64 >> 21) & 0x1FFFFF);
  rr.end_read = 0x1FFFFF - (int)(k64 & 0x1FFFFF);
  return rr;
}

// a result: x = score, y = (end_ref + 1) << 16 | end_read
__device__ __forceinline__ uint2 wres_pack(const SwRes& s) { return make_uint2((uint32_t)s.score, s.end_ref < -1 ? 0xFFFFFFFFu : ((uint32_t)(s.end_ref + 1) << 16) | ((uint32_t)s.end_read & 0xFFFFu)); }
__device__ __forceinline__ SwRes wres_unpack(const uint2 v) {
  SwRes s; s.score = (int)v.x;
  if (v.y == 0xFFFFFFFFu) { s.end_ref = -2; s.end_read = 0; }
  else { s.end_ref = (int)(v.y >> 16) - 1; s.end_read = (int)(v.y & 0xFFFFu); }
  return s;
}

// R rows per virtual lane = read spans up to 8 R letters; the host picks the instantiation from the longest read of the batch (13: <= 104
// letters, 19: <= 152, 26: <= 208, 32: <= 256).  Registers: five per row (two score tables, Y, E, the running-maximum key) + ~30
#ifndef SW16_WAVES
#define SW16_WAVES(R) ((R) <= 13 ? 4 : (R) <= 19 ? 3 : 2)
#endif
template <int R>
__global__ void __launch_bounds__(64, SW16_WAVES(R)) k_sw16(DReads rd, DIndex ix, DParams P, const WTask* __restrict__ tk, const uint32_t* __restrict__ tidx1,
                                                           const uint32_t* __restrict__ tidx2, const unsigned long long* __restrict__ wc, uint2* __restrict__ res) {
  const int lane = lane_id(), g = lane >> 2, gl = lane & 3;
  // two lists: tidx[0, ntA) = tasks scored with their end cells, tidx2[0, ntB) = tasks of which only the score is asked
  const uint32_t ntA = (uint32_t)wc[WC_NTASK], ntB = (uint32_t)wc[WC_NTASK2];
  const uint32_t npassA = (ntA + 15u) / 16u, npass = npassA + (ntB + 15u) / 16u;
  // (the slot of a pass's task is asked for one pass ahead: the first of the four dependent loads -- slot, task, read, letters -- in front of a pass's work)
  auto slot_of = [&](uint32_t p) -> uint32_t {
    const bool keys = p < npassA;
    const uint32_t ti = (keys ? p : p - npassA) * 16u + (uint32_t)g;
    return (p < npass && ti < (keys ? ntA : ntB)) ? (keys ? tidx1 : tidx2)[ti] : NONE;
  };
  uint32_t slot_next = slot_of(blockIdx.x);
  for (uint32_t p = blockIdx.x; p < npass; p += gridDim.x) {
    const bool keys = p < npassA;
    const uint32_t ti = (keys ? p : p - npassA) * 16u + (uint32_t)g;
    const uint32_t* const tidx = keys ? tidx1 : tidx2;
    const uint32_t slot = slot_next;
    const bool have = slot != NONE;
    int m = 0, n = 0, aq = 0, dir = 1;
    uint32_t len = 0, reversed = 0;
    const uint32_t* rec = rd.words;
    const uint8_t* ref = ix.ref_seq;
    if (have) {
      const WTask t = tk[slot];
      m = t.m; n = t.nref; aq = t.aq; reversed = t.flags & 1u; dir = (t.flags & 2u) ? -1 : 1;
      len = rd.len[t.r]; rec = rd.words + rd.rec_off[t.r];
      ref = ix.ref_seq + t.rf_start;
    }
    slot_next = slot_of(p + gridDim.x);
    int nn = n;
    for (int d = 32; d > 0; d >>= 1) nn = max(nn, __shfl_xor(nn, d, 64));
    // does the window hold an ambiguous letter?  Only asked of a reference DB that has one at all (DIndex::ref_any_n), eight letters per lane at a time
    bool hn = false;
    if (ix.ref_any_n) {
      const int nm = max(n, 1) - 1;
      for (int q0 = gl; q0 < nn; q0 += 32) {
        uint8_t b[8];
#pragma unroll
        for (int i = 0; i < 8; i++) b[i] = ref[dir * min(q0 + 4 * i, nm)];
#pragma unroll
        for (int i = 0; i < 8; i++) hn |= (q0 + 4 * i < n) && b[i] == 4;
      }
    }
    SwRes s;
    const bool hasn = __any(hn);
#define SW16_ARGS rec, len, reversed, m, aq, ref, n, P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext, dir
    if (keys) { if (hasn) s = sw_quad16<R, true, true>(SW16_ARGS); else s = sw_quad16<R, false, true>(SW16_ARGS); }
    else { if (hasn) s = sw_quad16<R, true, false>(SW16_ARGS); else s = sw_quad16<R, false, false>(SW16_ARGS); }
#undef SW16_ARGS
    // (the task's slot is asked for again rather than kept across the Smith-Waterman loop, whose rows take every register the occupancy allows)
    if (have && gl == 0) res[tidx[ti]] = wres_pack(s);
  }
}

// ------------------------------------------------------------------------------------------------
// k_walk<FINAL>: one wave per listed read and round.
// ------------------------------------------------------------------------------------------------
#ifndef SMR_WALK_WAVES_PER_SIMD
#define SMR_WALK_WAVES_PER_SIMD 5
#endif
template <bool FINAL>
__global__ void __launch_bounds__(64, FINAL ? 3 : SMR_WALK_WAVES_PER_SIMD)
k_walk(DReads rd, DIndex ix, DParams P, int pass, int is_last_strand, RState* __restrict__ work, AlignRec* __restrict__ work_aln, RWork* __restrict__ rw,
       unsigned long long* __restrict__ ctr, const uint2* __restrict__ mrec, const uint32_t* __restrict__ mpool, const uint32_t* __restrict__ pool,
       const uint2* __restrict__ list, const WState* __restrict__ ws_prev, const WTask* __restrict__ tk_prev, const uint2* __restrict__ res_prev,
       WState* __restrict__ ws_cur, WTask* __restrict__ tk_cur, uint32_t* __restrict__ tidx, uint32_t* __restrict__ tidx2, unsigned long long* __restrict__ wc,
       uint32_t K0, unsigned long long task_cap, int round, uint32_t lds_ml, uint32_t lds_rf, uint32_t assume_min) {
  SMR_DYN_LDS(unsigned char, lds_raw);                  // FINAL: read letters (lds_ml) | reference window (lds_rf)
  __shared__ unsigned long long l_pairs[WK_MAX_POS];    // (reference position << 32 | window position) of the sorted triples of the candidates
  __shared__ uint2 l_cand[64];                          // candidates in walk order: {reference, count | first triple << 8}
  __shared__ unsigned long long l_cref[64];             // ... where their reference sequences start,
  __shared__ uint32_t l_clen[64];                       // ... and their lengths
  __shared__ uint32_t l_hkey[256], l_hcnt[256];         // the hash table that groups the triples by reference
  __shared__ unsigned long long l_stage[WK_MAX_POS];    // the member triples as sort keys (before that: the hits of a read that gathers its positions itself; after: LIS arrays)
  __shared__ WTask s_ctk[WK_MAX];                       // the tasks the read left in the previous round ...
  __shared__ uint2 s_cres[WK_MAX];                      // ... and their results
  __shared__ uint32_t s_tix[WK_CLAIM * WK_MAX];         // task slots of the chunk being worked on (appended to tidx / tidx2 with one atomic per chunk):
                                                        // the tasks to be scored with their end cells from the front, the score-only ones from the back
  __shared__ uint32_t s_next, s_tbase;
  const int lane = lane_id();
  const uint32_t nlist = (uint32_t)wc[WC_NLIST];
  const uint32_t K = walk_tasks_per_read(nlist, task_cap, K0, round);                                                // tasks per read of this round ...
  const uint32_t Kp = round > 0 ? walk_tasks_per_read((uint32_t)(wc - WC_STRIDE)[WC_NLIST], task_cap, K0, round - 1) : K0;      // ... and of the previous one (where this round's reads left theirs)
  unsigned long long n_fwd = 0, n_cells = 0, n_spec = 0, n_spec_used = 0, n_newhit = 0;
#ifdef SMR_WALK_PHASES                                    // where a wave's cycles go (build with -DSMR_WALK_PHASES, run with SMR_DEBUG_PHASES=1): claim + loads, sort + candidates, advance, results + bookkeeping, look-ahead + tasks, write-back, task list
  unsigned long long wph[7] = {0, 0, 0, 0, 0, 0, 0}, wlast = clock64();
#define WPH(i) { const unsigned long long tn_ = clock64(); wph[i] += tn_ - wlast; wlast = tn_; }
#else
#define WPH(i)
#endif
  const uint32_t claim = max(1u, min(WK_CLAIM, nlist / (gridDim.x * 4u)));
  for (;;) {
    __syncthreads();
    if (lane == 0) s_next = (uint32_t)atomicAdd(&wc[WC_CLAIM], (unsigned long long)claim);
    __syncthreads();
    const uint32_t chunk_base = s_next;
    if (chunk_base >= nlist) break;
    const uint32_t chunk_n = min(claim, nlist - chunk_base);
    uint32_t ntix = 0, ntix2 = 0;
    // the chunk's entries at once: lane i asks for entry chunk_base + i and its read's record and length (one round trip per chunk, not two per read)
    uint32_t c_r = 0, c_prev = NONE, c_len = 0;
    uint2 c_mr = make_uint2(NONE, 0u);
    if ((uint32_t)lane < chunk_n) { const uint2 le = list[chunk_base + lane]; c_r = le.x; c_prev = le.y; c_mr = mrec[c_r]; c_len = rd.len[c_r]; }
    for (uint32_t ci = 0; ci < chunk_n; ci++) {
      const uint32_t e = chunk_base + ci;
      const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)c_r, (int)ci), prev = (uint32_t)__builtin_amdgcn_readlane((int)c_prev, (int)ci);
      const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)c_len, (int)ci);
      const uint2 mr = make_uint2((uint32_t)__builtin_amdgcn_readlane((int)c_mr.x, (int)ci), (uint32_t)__builtin_amdgcn_readlane((int)c_mr.y, (int)ci));
      RWork w = rw[r];
      RState st = work[r];
      const uint32_t* rec = FINAL ? rd.words + rd.rec_off[r] : nullptr;
      WState ps; ps.bits = 0;
      if (prev != NONE) ps = ws_prev[prev];
      uint32_t live_bits = 0;                             // what ws_cur[e].bits becomes: 0 = the read is finished
      if (w.strand_active && w.search && w.pass_n == (uint32_t)pass && mr.y - 1u < WK_MAX_POS) {
        WPH(0)
        // ---- candidate references (alignment.cpp:117-148) and their (reference position, window position) pairs in walk order ----
        // The read's triples (reference, reference position, window position), up to two per lane: from k_cand's record, or -- a read whose
        // record found no room, or that has 65..128 positions -- gathered here through hits -> list bounds -> positions.  Equal references are
        // found by an exact hash table in LDS; the triples of the references with enough seeds are moved to the front and sorted by (slot,
        // reference position, window position) as ONE 64-bit key each: a candidate's pairs are a run, already in the order the walk wants
        // them, and a read with one spurious candidate of two seeds sorts two lanes, not all of them.
        const uint32_t npos = mr.y;
        uint32_t seq[2] = {0, 0}, pos[2] = {0, 0}, win[2] = {0, 0};
        const bool valid[2] = {(uint32_t)lane < npos, (uint32_t)lane + 64u < npos};
        __syncthreads();
        if (mr.x != NONE) {
          const uint32_t* rp = mpool + mr.x;
          if (valid[0]) { seq[0] = rp[lane]; pos[0] = rp[npos + lane]; win[0] = rp[2u * npos + lane]; }      // (a record has at most 64 positions)
        } else {
          uint32_t* const g_hp = (uint32_t*)l_stage;            // [65] first position of every hit | [64] its list start | [64] its window position
          uint32_t* const g_lo = g_hp + 65, * const g_wn = g_lo + 64;
          const uint32_t nh = w.hit_total;                      // (<= CAND_HITS = 64: k_cand only scans such reads)
          uint32_t lo = 0, ln = 0;
          if ((uint32_t)lane < nh) {
            const uint32_t h = (uint32_t)lane, c0 = w.blk_cnt[0], c1 = w.blk_cnt[1];
            const uint32_t at = h < c0 ? w.blk_off[0] + 2 * h : h - c0 < c1 ? w.blk_off[1] + 2 * (h - c0) : w.blk_off[2] + 2 * (h - c0 - c1);
            const uint32_t id = pool[at];
            g_wn[lane] = pool[at + 1];
            lo = id + 1u; ln = ix.pos_arr[id].x;               // (the id is the place of the list's header word)
          }
          uint32_t tot;
          const uint32_t ex = wave_excl_scan_u32(ln, tot);
          if ((uint32_t)lane < nh) { g_hp[lane] = ex; g_lo[lane] = lo; }
          if (lane == 0) g_hp[nh] = tot;
          __syncthreads();
#pragma unroll
          for (int q = 0; q < 2; q++) if (valid[q]) {
            const uint32_t p = (uint32_t)lane + 64u * q;
            uint32_t h = 0;
            for (uint32_t step = 32; step > 0; step >>= 1) { const uint32_t t = h + step; if (t < nh && g_hp[t] <= p) h = t; }
            const uint2 pa = ix.pos_arr[g_lo[h] + (p - g_hp[h])];
            seq[q] = pa.y; pos[q] = pa.x; win[q] = g_wn[h];
          }
          __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { l_hkey[lane + 64 * q] = 0xFFFFFFFFu; l_hcnt[lane + 64 * q] = 0; }
        __syncthreads();
        uint32_t slot[2] = {0, 0};
#pragma unroll
        for (int q = 0; q < 2; q++) if (valid[q]) {
          uint32_t sl = (seq[q] * 0x9E3779B1u) >> 24;
          for (;;) { const uint32_t o = atomicCAS(&l_hkey[sl], 0xFFFFFFFFu, seq[q]); if (o == 0xFFFFFFFFu || o == seq[q]) break; sl = (sl + 1u) & 255u; }
          atomicAdd(&l_hcnt[sl], 1u);
          slot[q] = sl;
        }
        __threadfence_block();
        __syncthreads();
        const bool mem0 = valid[0] && (int)l_hcnt[slot[0]] >= P.num_seeds, mem1 = valid[1] && (int)l_hcnt[slot[1]] >= P.num_seeds;
        const unsigned long long mm0 = __ballot(mem0), mm1 = __ballot(mem1);
        const uint32_t tot0 = (uint32_t)__popcll(mm0), total = tot0 + (uint32_t)__popcll(mm1);
        const unsigned long long ltm = (1ull << lane) - 1ull;
        if (mem0) l_stage[__popcll(mm0 & ltm)] = ((unsigned long long)slot[0] << 48) | ((unsigned long long)pos[0] << 16) | (unsigned long long)(win[0] & 0xFFFFu);
        if (mem1) l_stage[tot0 + (uint32_t)__popcll(mm1 & ltm)] = ((unsigned long long)slot[1] << 48) | ((unsigned long long)pos[1] << 16) | (unsigned long long)(win[1] & 0xFFFFu);
        __syncthreads();
        if (total > 64) wave_sort_u64(l_stage, total);          // (in LDS; rare: more than 64 triples belong to candidates)
        else if (total > 1) {
          unsigned long long key = (uint32_t)lane < total ? l_stage[lane] : ~0ull;
          uint32_t np2 = 2; while (np2 < total) np2 <<= 1;
          for (uint32_t kk = 2; kk <= np2; kk <<= 1)
            for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
              const unsigned long long o = __shfl_xor(key, (int)j, 64);
              const bool want_min = (((uint32_t)lane & j) == 0) == (((uint32_t)lane & kk) == 0);
              if (want_min ? o < key : key < o) key = o;
            }
          if ((uint32_t)lane < total) l_stage[lane] = key;
        }
        __syncthreads();
        // runs of equal slots in the sorted keys = the candidates; element i of the sorted array is looked at by lane i & 63 (two per lane)
        unsigned long long key2[2]; bool head[2]; uint32_t sl2[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const uint32_t i = (uint32_t)lane + 64u * q;
          key2[q] = i < total ? l_stage[i] : ~0ull;
          sl2[q] = (uint32_t)(key2[q] >> 48) & 255u;
          head[q] = i < total && (i == 0 || sl2[q] != ((uint32_t)(l_stage[i - (i ? 1u : 0u)] >> 48) & 255u));
        }
        const unsigned long long h0 = __ballot(head[0]), h1 = __ballot(head[1]);
        const uint32_t ncand = (uint32_t)__popcll(h0) + (uint32_t)__popcll(h1);
        uint32_t count[2], hseq[2];
        {
          const unsigned long long ab0 = lane < 63 ? h0 >> (lane + 1) : 0ull, ab1 = lane < 63 ? h1 >> (lane + 1) : 0ull;
          const uint32_t e0 = ab0 ? (uint32_t)lane + (uint32_t)__ffsll((long long)ab0) : h1 ? 63u + (uint32_t)__ffsll((long long)h1) : total;
          const uint32_t e1 = ab1 ? 64u + (uint32_t)lane + (uint32_t)__ffsll((long long)ab1) : total;
          count[0] = e0 - (uint32_t)lane; count[1] = e1 - (64u + (uint32_t)lane);      // (of a head: the length of its run)
          hseq[0] = head[0] ? l_hkey[sl2[0]] : 0u; hseq[1] = head[1] ? l_hkey[sl2[1]] : 0u;
        }
        // walk order: count descending, reference ascending (alignment.cpp:134-148)
        const unsigned long long my0 = ((unsigned long long)(0xFFFFFFFFu - count[0]) << 32) | hseq[0], my1 = ((unsigned long long)(0xFFFFFFFFu - count[1]) << 32) | hseq[1];
        uint32_t rank0 = 0, rank1 = 0;
        for (unsigned long long mq = h0; mq; mq &= mq - 1) {
          const int c = __ffsll((long long)mq) - 1;
          const unsigned long long key_c = ((unsigned long long)(0xFFFFFFFFu - (uint32_t)__builtin_amdgcn_readlane((int)count[0], c)) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)hseq[0], c);
          rank0 += key_c < my0 ? 1u : 0u; rank1 += key_c < my1 ? 1u : 0u;
        }
        for (unsigned long long mq = h1; mq; mq &= mq - 1) {
          const int c = __ffsll((long long)mq) - 1;
          const unsigned long long key_c = ((unsigned long long)(0xFFFFFFFFu - (uint32_t)__builtin_amdgcn_readlane((int)count[1], c)) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)hseq[1], c);
          rank0 += key_c < my0 ? 1u : 0u; rank1 += key_c < my1 ? 1u : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const uint32_t i = (uint32_t)lane + 64u * q;
          if (i < total) l_pairs[i] = ((key2[q] >> 16) << 32) | (key2[q] & 0xFFFFull);
          if (head[q]) {
            // (where the candidate's reference sequence lies: asked for by all candidates at once, not one round trip per candidate as the walk reaches it)
            const uint32_t rk = q ? rank1 : rank0;
            const uint64_t r0_ = ix.ref_off[hseq[q]], r1_ = ix.ref_off[hseq[q] + 1];
            l_cand[rk] = make_uint2(hseq[q], count[q] | (i << 8));
            l_cref[rk] = r0_; l_clen[rk] = (uint32_t)(r1_ - r0_);
          }
        }
        // the tasks of the previous round and their results
        const uint32_t n_prev = prev != NONE ? WS_NK(ps.bits) : 0u;
        if ((uint32_t)lane < n_prev) { s_ctk[lane] = tk_prev[(size_t)prev * Kp + lane]; s_cres[lane] = res_prev[(size_t)prev * Kp + lane]; }
        __syncthreads();
        WPH(1)

        // ONE walk state per read: the real walk consumes results until it meets a task without one; from there the same state runs on as
        // the look-ahead (the read is suspended: what it resumes from is written first).  What a candidate's run of pairs, its reference and
        // that reference's place are is read from LDS where it is needed, not carried along.
        struct Walk { uint32_t k, it, ms_lo, ms_hi, begin_ref, begin_read; int is_aligned, best, go_on, started, pending_pop; };
        const uint64_t rlen = len;
        // candidate wk.k: termination rules (:156-169).  1 = loaded, 0 = the candidate loop ends here
        auto load_candidate = [&](Walk& wk) -> int {
          if (wk.k >= ncand || !wk.go_on) return 0;
          const uint32_t cy = l_cand[wk.k].y;
          const uint32_t max_occur = cy & 0xFFu;
          if (max_occur < (uint32_t)P.num_seeds) return 0;
          if (wk.is_aligned && P.min_lis > 0 && wk.k > 0 && max_occur < (l_cand[wk.k - 1].y & 0xFFu)) {   // :165-169
            --wk.best;
            if (wk.best < 1) return 0;
          }
          const unsigned long long p0 = l_pairs[cy >> 8];
          wk.it = 0; wk.ms_lo = 0; wk.ms_hi = 0;
          wk.begin_ref = (uint32_t)(p0 >> 32); wk.begin_read = (uint32_t)p0;
          wk.pending_pop = 0;
          return 1;
        };
        // the sliding window of read length along candidate wk.k (:203-506), up to its next window that calls for ssw_align
        auto next_task = [&](Walk& wk, SwTask& tk) -> bool {
          const uint2 ce = l_cand[wk.k];
          const unsigned long long* pairs = l_pairs + (ce.y >> 8);
          const uint32_t np = ce.y & 0xFFu;
          while (wk.it != np && wk.go_on) {
            if (!wk.pending_pop) {
              wk.pending_pop = 1;
              const uint64_t end_ref_max = (uint64_t)wk.begin_ref + len - wk.begin_read - P.lnwin + 1;
              int push = 0;
              for (;;) {                                            // (the pairs are sorted by reference position: the ones to push are a prefix of what is left)
                const uint32_t pi = wk.it + (uint32_t)lane;
                const bool okp = pi < np && (uint64_t)(uint32_t)(pairs[min(pi, np - 1)] >> 32) <= end_ref_max;
                const unsigned long long pm = __ballot(okp);
                const uint32_t pc = pm == ~0ull ? 64u : (uint32_t)__ffsll((long long)~pm) - 1u;
                if (pc) { wk.it += pc; wk.ms_hi = wk.it; push = 1; }
                if (pc < 64u) break;
              }
              int skip_to_pop = 0;
              if (!push && wk.is_aligned) skip_to_pop = 1;        // heuristic 1 (:243-246)
              else wk.is_aligned = 0;
              if (!skip_to_pop && (wk.ms_hi - wk.ms_lo) >= (uint32_t)P.num_seeds) {
                uint32_t lis0;
                const uint32_t nw = wk.ms_hi - wk.ms_lo;
                uint32_t* const lisb = (uint32_t*)l_stage;           // (free once the candidates are laid out: 2 x WK_MAX_POS words)
                const uint32_t nl = nw <= 64 ? wave_lis_first(pairs + wk.ms_lo, nw, lis0) : serial_lis_first(pairs + wk.ms_lo, nw, lisb, lisb + WK_MAX_POS, lis0);
                if (nl >= (uint32_t)P.min_lis) {
                  const unsigned long long pl = pairs[wk.ms_lo + lis0];
                  const uint32_t lcs_ref_start = (uint32_t)(pl >> 32), lcs_que_start = (uint32_t)pl;
                  const uint64_t reflen = l_clen[wk.k];
                  uint64_t hd = 0, tail = 0, align_ref_start = 0, align_que_start = 0, align_length = 0;
                  uint32_t edges;
                  if (P.is_as_percent) edges = (uint32_t)((P.edges / 100.0) * (double)rlen);
                  else edges = (uint32_t)P.edges;
                  if (lcs_ref_start < lcs_que_start) {                         // :287-325
                    align_ref_start = 0; align_que_start = lcs_que_start - lcs_ref_start; hd = 0;
                    if (reflen < rlen) {
                      tail = 0;
                      if (align_que_start > (rlen - reflen)) align_length = reflen - (align_que_start - (rlen - reflen));
                      else align_length = reflen;
                    } else {
                      tail = reflen - align_ref_start - rlen;
                      if (tail > (uint64_t)(uint32_t)(edges - 1)) tail = edges;
                      align_length = rlen + hd + tail - align_que_start;
                    }
                  } else {                                                     // :326-357
                    align_ref_start = lcs_ref_start - lcs_que_start; align_que_start = 0;
                    if (align_ref_start > (uint64_t)(uint32_t)(edges - 1)) hd = edges;
                    if (align_ref_start + rlen > reflen) { tail = 0; align_length = reflen - align_ref_start - hd; }
                    else {
                      tail = reflen - align_ref_start - rlen;
                      if (tail > (uint64_t)(uint32_t)(edges - 1)) tail = edges;
                      align_length = rlen + hd + tail;
                    }
                  }
                  tk.max_ref = ce.x; tk.align_ref_start = align_ref_start; tk.head = hd; tk.align_que_start = align_que_start;
                  tk.m = (int)(align_length - hd - tail); tk.nref = (int)align_length;
                  tk.rf_start = l_cref[wk.k] + align_ref_start - hd;
                  return true;
                }
              }
            }
            // pop (:486-506)
            wk.pending_pop = 0;
            if (wk.ms_hi > wk.ms_lo) wk.ms_lo++;
            if (wk.ms_hi == wk.ms_lo) {
              if (wk.it != np) { const unsigned long long pn = pairs[wk.it]; wk.begin_ref = (uint32_t)(pn >> 32); wk.begin_read = (uint32_t)pn; }
              else break;
            } else { const unsigned long long pn = pairs[wk.ms_lo]; wk.begin_ref = (uint32_t)(pn >> 32); wk.begin_read = (uint32_t)pn; }
          }
          return false;
        };
        // 1 = the walk stands at a task, 0 = the walk is over
        auto advance = [&](Walk& wk, SwTask& tk) -> int {
          for (;;) {
            if (!wk.started) {
              if (load_candidate(wk) != 1) return 0;
              wk.started = 1;
            }
            if (next_task(wk, tk)) return 1;
            wk.k++; wk.started = 0;
          }
        };
        auto put_task = [&](uint32_t j, const SwTask& t) {
          if (lane == 0) {
            WTask o; o.r = r; o.max_ref = t.max_ref; o.rf_start = t.rf_start; o.ars = (uint32_t)t.align_ref_start; o.head = (uint32_t)t.head; o.aq = (uint16_t)t.align_que_start;
            o.m = (uint16_t)t.m; o.nref = (uint16_t)t.nref; o.flags = w.reversed ? 1 : 0;
            tk_cur[(size_t)e * K + j] = o;
          }
        };
        auto task_ok = [&](const SwTask& t) -> bool { return t.m > 0 && t.nref > 0 && (uint32_t)t.m <= lds_ml && (uint32_t)t.nref <= lds_rf && (uint32_t)t.nref <= 0xFFFFu; };

        Walk R;
        int search = 1, last_aligned = 0;
        SwTask tk;
        bool have_tk = false;                               // a resumed read stands at the first task it left: no advance, the task is in the cache
        if (prev == NONE) {
          R.k = 0; R.it = 0; R.ms_lo = 0; R.ms_hi = 0; R.begin_ref = 0; R.begin_read = 0;
          R.is_aligned = 0; R.best = w.best; R.go_on = 1; R.started = 0; R.pending_pop = 0;
        } else {
          R.k = ps.k; R.it = ps.it; R.ms_lo = ps.ms_lo; R.ms_hi = ps.ms_hi; R.begin_ref = ps.begin_ref; R.begin_read = ps.begin_read; R.best = ps.best;
          R.is_aligned = (int)(ps.bits & 1u); R.go_on = (int)((ps.bits >> 1) & 1u); R.started = (int)((ps.bits >> 2) & 1u); R.pending_pop = (int)((ps.bits >> 3) & 1u);
          search = (int)((ps.bits >> 4) & 1u); last_aligned = (int)((ps.bits >> 6) & 1u);
          const WTask c0 = s_ctk[0];
          tk.max_ref = c0.max_ref; tk.rf_start = c0.rf_start; tk.align_ref_start = c0.ars; tk.head = c0.head; tk.align_que_start = c0.aq; tk.m = c0.m; tk.nref = c0.nref;
          have_tk = true;
        }
        // what the look-ahead assumes of the tasks it runs past: in round 0 "aligns" for a read whose best candidate has many seeds (a read sampled
        // from the DB meets a family of references, one accepted alignment each), "does not" otherwise (a spurious candidate of a background read);
        // later what the read's last result was.  Only which windows get scored ahead depends on it, never a result.
        const uint32_t max_SW_score = len * (uint32_t)P.match;
        bool live = false, rdq_staged = false, dirty = false;      // dirty: st / w changed by an accepted alignment (flip34 alone is redone where it matters)
        for (;;) {
          WPH(3)
          if (!have_tk) { if (advance(R, tk) != 1) break; }
          WPH(2)
          if (w.has_amb && !w.is04) { w.is04 = 1; w.aval = 4; }           // read.flip34() to the 0..4 alphabet before SSW (:360-361)
          const int m = tk.m, nref = tk.nref;
          const bool sw_ok = task_ok(tk);
          if (!sw_ok && (m > 0 && nref > 0)) { if (lane == 0) atomicAdd(&ctr[C_ERR_PAIRS], 1ull); }
          SwRes fw; fw.score = 0; fw.end_ref = -1; fw.end_read = m - 1;
          if (sw_ok) {
            int ce = -1;
            if (have_tk) ce = 0;
            else for (uint32_t q = 0; q < n_prev; q++)
              if (s_ctk[q].max_ref == tk.max_ref && s_ctk[q].rf_start == tk.rf_start && (uint32_t)s_ctk[q].aq == (uint32_t)tk.align_que_start && (int)s_ctk[q].m == m && (int)s_ctk[q].nref == nref) { ce = (int)q; break; }
            if (ce >= 0) { fw = wres_unpack(s_cres[ce]); if (ce > 0) n_spec_used++; }
            else if (!FINAL) {
              // the walk needs a result it does not have: this task and the ones the walk reaches next under the prediction become the read's
              // tasks of this round; the state the read resumes from (standing at this task) is written first, then the same walk runs on as the look-ahead
              const int assume = prev == NONE ? ((ncand > 0 && (l_cand[0].y & 0xFFu) >= assume_min) ? 1 : 0) : last_aligned;
              WState o; o.k = R.k; o.it = R.it; o.ms_lo = R.ms_lo; o.ms_hi = R.ms_hi; o.begin_ref = R.begin_ref; o.begin_read = R.begin_read; o.best = R.best;
              const uint32_t bits0 = (uint32_t)(R.is_aligned & 1) | ((uint32_t)(R.go_on & 1) << 1) | ((uint32_t)(R.started & 1) << 2) | ((uint32_t)(R.pending_pop & 1) << 3) | ((uint32_t)(search & 1) << 4) |
                                     ((uint32_t)(last_aligned & 1) << 6);
              uint32_t nk = 1, cells = (uint32_t)m * (uint32_t)nref;
              put_task(0, tk);
              R.is_aligned = assume;
              bool ended = false;
              while (nk < K) {
                SwTask t2;
                if (advance(R, t2) != 1) { ended = true; break; }
                if (!task_ok(t2)) break;
                put_task(nk, t2); nk++; cells += (uint32_t)t2.m * (uint32_t)t2.nref;
                R.is_aligned = assume;
              }
              n_spec += nk - 1;
              o.bits = bits0 | ((ended && !assume) ? 32u : 0u) | (nk << 8); o.cells = cells;
              if (lane == 0) ws_cur[e] = o;
              live_bits = nk << 8;
              // (a read that is expected to align gets its end cells with the scores; of the others only the score is asked)
              if (assume) { for (uint32_t q = lane; q < nk; q += 64) s_tix[ntix + q] = e * K + q; ntix += nk; }
              else { for (uint32_t q = lane; q < nk; q += 64) s_tix[WK_CLAIM * WK_MAX - 1u - (ntix2 + q)] = e * K + q; ntix2 += nk; }
              live = true;
              WPH(4)
              break;
            } else {
              // last round: score it here, one problem per wave
              uint8_t* const rdq = lds_raw; uint8_t* const rfq = lds_raw + lds_ml;
              __syncthreads();
              if (!rdq_staged) { for (uint32_t q = lane; q < len; q += 64) rdq[q] = (uint8_t)read_nt(rec, len, q, w.reversed, 4u); rdq_staged = true; }
              for (int q = lane; q < nref; q += 64) rfq[q] = ix.ref_seq[tk.rf_start + q];
              __syncthreads();
              fw = sw_wave(rdq, m, (int)tk.align_que_start, 1, rfq, nref, 0, 1, nullptr, P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext, P.sw_mode);
              __syncthreads();
            }
            n_fwd++; n_cells += (unsigned long long)m * nref;
          }
          have_tk = false;
          const int score1 = fw.score > 65535 ? 65535 : fw.score;
          // (the begin cell is k_begins' business, as with k_chain: window start in ref_begin1 / read_begin1, has_cigar = 2)
          R.is_aligned = (sw_ok && (uint32_t)score1 > P.minimal_score);     // strict (:388)
          last_aligned = R.is_aligned;
          if (R.is_aligned) {
            dirty = true;
            if ((uint32_t)score1 == max_SW_score) ++st.max_SW_count;
            AlignRec al;
            al.ref_begin1 = (int32_t)(tk.align_ref_start - tk.head);
            // (a task of which only the score was asked: the END of its window stands in for the end cell, has_cigar = 3, k_begins finds both cells)
            const bool no_end = fw.end_ref < -1;
            al.ref_end1 = (no_end ? nref - 1 : fw.end_ref) + (int32_t)(tk.align_ref_start - tk.head);
            al.read_begin1 = (int32_t)tk.align_que_start;
            al.read_end1 = (no_end ? m - 1 : fw.end_read) + (int32_t)tk.align_que_start;
            al.readlen = len; al.ref_num = tk.max_ref;
            al.index_num = (uint16_t)P.index_num; al.part = (uint16_t)P.part;
            al.strand = (uint8_t)!w.reversed; al.score1 = (uint16_t)score1;
            al.has_cigar = no_end ? 3 : 2; al.cigar_off = 0; al.cigar_len = 0;
            AlignRec* slots = work_aln + (size_t)r * P.slots;
            if (!st.is_hit) {                                              // :411-416
              st.is_hit = 1;
              n_newhit++;                                                  // (counted per wave and added once: 400 000 reads become hits in one launch, and atomics on one line retire at 83 per microsecond)
            }
            if (P.num_alignments == 0 || !P.is_best || (P.is_best && st.n_align < P.num_alignments)) {
              if (st.n_align < P.slots) { if (lane == 0) slots[st.n_align] = al; st.n_align++; w.is_new_hit = 1; }
              else { if (lane == 0) atomicAdd(&ctr[C_ERR_SLOTS], 1ull); }
            } else if (P.is_best && st.n_align == P.num_alignments) {
              __syncthreads();
              if (slots[st.min_index].score1 < (uint16_t)score1) {         // :425-459
                if (P.num_alignments > 1 && st.max_index == 0 && st.min_index == 0) {
                  uint32_t mn = 0, mx = 0;
                  for (uint32_t q = 1; q < st.n_align; q++) { if (slots[q].score1 < slots[mn].score1) mn = q; if (slots[q].score1 > slots[mx].score1) mx = q; }
                  st.min_index = mn; st.max_index = mx;
                }
                const uint32_t mn = st.min_index, mx = st.max_index;
                const uint16_t mx_score = slots[mx].score1;
                __syncthreads();
                if (lane == 0) slots[mn] = al;
                __threadfence_block();
                __syncthreads();
                w.is_new_hit = 1;
                if ((uint16_t)score1 > (mn == mx ? (uint16_t)score1 : mx_score) && st.n_align > 1) {
                  st.max_index = mn;
                  uint32_t m2 = 0;
                  for (uint32_t q = 1; q < st.n_align; q++) if (slots[q].score1 < slots[m2].score1) m2 = q;
                  st.min_index = m2;
                }
              }
            }
            __syncthreads();
            if (P.num_alignments > 0) {                                    // :462-469
              if (P.is_best) { if (P.num_alignments == st.max_SW_count) R.go_on = 0; }
              else if (P.num_alignments == st.n_align) R.go_on = 0;
            }
            search = 0;
          }
        }
        if (live) { if (dirty && lane == 0) { work[r] = st; rw[r] = w; } }
        else { w.best = R.best; chain_finish_read(P, is_last_strand, r, st, w, search, lane == 0, work, rw); }
        WPH(5)
      }
      if (!live_bits && lane == 0) ws_cur[e].bits = 0;
    }
    // the chunk's tasks go to the dense list with one atomic
    if (!FINAL && ntix) {
      __syncthreads();
      if (lane == 0) s_tbase = (uint32_t)atomicAdd(&wc[WC_NTASK], (unsigned long long)ntix);
      __syncthreads();
      const uint32_t tb = s_tbase;
      for (uint32_t q = lane; q < ntix; q += 64) tidx[tb + q] = s_tix[q];
    }
    if (!FINAL && ntix2) {
      __syncthreads();
      if (lane == 0) s_tbase = (uint32_t)atomicAdd(&wc[WC_NTASK2], (unsigned long long)ntix2);
      __syncthreads();
      const uint32_t tb = s_tbase;
      for (uint32_t q = lane; q < ntix2; q += 64) tidx2[tb + q] = s_tix[WK_CLAIM * WK_MAX - 1u - q];
    }
    WPH(6)
  }
  if (lane == 0) {
    if (n_fwd) ctr_add(ctr, C_SW_FWD, n_fwd);
    if (n_cells) ctr_add(ctr, C_SW_CELLS, n_cells);
    if (n_spec) atomicAdd(&ctr[C_SW_SPEC], n_spec);
    if (n_spec_used) atomicAdd(&ctr[C_SW_SPEC_USED], n_spec_used);
    if (n_newhit) { atomicAdd(&ctr[C_NUM_ALIGNED], n_newhit); atomicAdd(&ctr[C_PER_DB + P.index_num], n_newhit); }
#ifdef SMR_WALK_PHASES
    for (int q = 0; q < 7; q++) if (wph[q]) atomicAdd(&ctr[C_SHARDS + (blockIdx.x & (C_NSHARD - 1)) * C_SHARD_W + C_SHARD_PH + q], wph[q]);
#endif
  }
#undef WPH
}

// ------------------------------------------------------------------------------------------------
// k_wnext: after the round's tasks are scored.  A read whose look-ahead reached the end of its walk under "nothing aligns" and whose results
// are all "no alignment" ends exactly as the sequential walk ends it: those ssw_align calls, nothing recorded, pass control.  Every other
// read that left tasks goes on the next round's list, with the place of its state, tasks and results.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_wnext(DParams P, int is_last_strand, RState* __restrict__ work, RWork* __restrict__ rw, unsigned long long* __restrict__ ctr,
                                                const uint2* __restrict__ list, const WState* __restrict__ ws, const uint2* __restrict__ res, uint2* __restrict__ list_next,
                                                const unsigned long long* __restrict__ wc, unsigned long long* __restrict__ wc_next, uint32_t K0, unsigned long long task_cap, int round) {
  const uint32_t n = (uint32_t)wc[WC_NLIST];
  const uint32_t K = walk_tasks_per_read(n, task_cap, K0, round);
  unsigned long long n_fwd = 0, n_cells = 0, n_used = 0;
  for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) {
    const uint32_t e = base + threadIdx.x;
    bool live = false;
    uint32_t r = 0;
    if (e < n) {
      const WState s = ws[e];
      const uint32_t nk = WS_NK(s.bits);
      if (nk > 0) {
        live = true;
        r = list[e].x;
        if (s.bits & 32u) {
          bool any = false;
          for (uint32_t j = 0; j < nk; j++) { const uint32_t sc = res[(size_t)e * K + j].x; any |= (sc > 65535u ? 65535u : sc) > P.minimal_score; }
          if (!any) {
            RWork w = rw[r];
            RState st = work[r];
            if (w.has_amb && !w.is04) { w.is04 = 1; w.aval = 4; }
            w.best = s.best;
            n_fwd += nk; n_cells += s.cells; n_used += nk - 1;
            chain_finish_read(P, is_last_strand, r, st, w, (int)((s.bits >> 4) & 1u), true, work, rw);
            live = false;
          }
        }
      }
    }
    const uint32_t o = block_append(&wc_next[WC_NLIST], live);
    if (live) list_next[o] = make_uint2(r, e);
  }
  for (int d = 32; d > 0; d >>= 1) { n_fwd += __shfl_xor(n_fwd, d, 64); n_cells += __shfl_xor(n_cells, d, 64); n_used += __shfl_xor(n_used, d, 64); }
  if (lane_id() == 0) {
    if (n_fwd) ctr_add(ctr, C_SW_FWD, n_fwd);
    if (n_cells) ctr_add(ctr, C_SW_CELLS, n_cells);
    if (n_used) atomicAdd(&ctr[C_SW_SPEC_USED], n_used);
  }
}

// ------------------------------------------------------------------------------------------------
// The begin cells of the alignments that are still stored when the part is done (ssw_align's reverse pass, ssw.c:900-918; see k_begins in
// smr_chain.hpp), sixteen per wave: k_begins_prep turns the pending alignments k_begins_collect listed into tasks of k_sw16 -- stage 0: the
// forward pass over the window of an alignment stored end-pending (has_cigar = 3), stage 1: the reverse pass from the end cell of every
// pending alignment --, k_begins_apply puts the cells into the records.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_begins_prep(DIndex ix, uint32_t slots, const uint32_t* __restrict__ tasks, const unsigned long long* __restrict__ n_tasks_p, const AlignRec* __restrict__ work_aln,
                                                      int stage, WTask* __restrict__ tk, uint32_t* __restrict__ tidx, unsigned long long* __restrict__ wc) {
  const uint32_t n_tasks = (uint32_t)*n_tasks_p;
  for (uint32_t base = blockIdx.x * blockDim.x; base < n_tasks; base += gridDim.x * blockDim.x) {
    const uint32_t t = base + threadIdx.x;
    bool take = false;
    if (t < n_tasks) {
      const AlignRec al = work_aln[tasks[t]];
      take = stage == 0 ? al.has_cigar == 3 : al.has_cigar >= 2;
      if (take) {
        WTask o;
        o.r = tasks[t] / slots; o.max_ref = al.ref_num; o.ars = 0; o.head = 0;
        o.m = (uint16_t)(al.read_end1 - al.read_begin1 + 1); o.nref = (uint16_t)(al.ref_end1 - al.ref_begin1 + 1);
        if (stage == 0) { o.aq = (uint16_t)al.read_begin1; o.rf_start = ix.ref_off[al.ref_num] + (uint64_t)al.ref_begin1; o.flags = al.strand ? 0 : 1; }
        else { o.aq = (uint16_t)al.read_end1; o.rf_start = ix.ref_off[al.ref_num] + (uint64_t)al.ref_end1; o.flags = (al.strand ? 0 : 1) | 2; }
        tk[t] = o;
      }
    }
    const uint32_t p = block_append(&wc[WC_NTASK], take);
    if (take) tidx[p] = t;
  }
}
__global__ void __launch_bounds__(256) k_begins_apply(const uint32_t* __restrict__ tasks, const unsigned long long* __restrict__ n_tasks_p, AlignRec* __restrict__ work_aln, int stage,
                                                      const WTask* __restrict__ tk, const uint2* __restrict__ res, unsigned long long* __restrict__ ctr) {
  const uint32_t n_tasks = (uint32_t)*n_tasks_p;
  unsigned long long n_rev = 0, n_cells = 0;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_tasks; t += gridDim.x * blockDim.x) {
    AlignRec* const al = work_aln + tasks[t];
    const uint32_t hc = al->has_cigar;
    if (stage == 0 ? hc != 3 : hc < 2) continue;
    const SwRes s = wres_unpack(res[t]);
    if (stage == 0) { al->ref_end1 = al->ref_begin1 + s.end_ref; al->read_end1 = al->read_begin1 + s.end_read; al->has_cigar = 2; }
    else { al->ref_begin1 = al->ref_end1 - s.end_ref; al->read_begin1 = al->read_end1 - s.end_read; al->has_cigar = 0; n_rev++; n_cells += (unsigned long long)tk[t].m * tk[t].nref; }
  }
  for (int d = 32; d > 0; d >>= 1) { n_rev += __shfl_xor(n_rev, d, 64); n_cells += __shfl_xor(n_cells, d, 64); }
  if (lane_id() == 0) { if (n_rev) ctr_add(ctr, C_SW_REV, n_rev); if (n_cells) ctr_add(ctr, C_SW_CELLS, n_cells); }
}

}  // namespace smr
