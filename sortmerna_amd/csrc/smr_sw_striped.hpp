// smr_sw_striped.hpp -- the slow path of the Smith-Waterman stage: ssw.c's stripe geometry on a wave (included by smr_chain.hpp).
#pragma once

namespace smr {

// The reference scores with Farrar's striped kernels (ssw.c:150-575): the read is cut into 16 (8-bit kernel) or 8 (16-bit kernel) SEGMENTS of
// segLen = ceil(len / lanes) rows, SIMD lane k works on segment k, and the F dependency that runs down a column across segment boundaries is
// repaired afterwards by a "lazy F" loop.  Two properties of that code are not the affine recurrence H = max(0, diag + s, E, F) that the fast
// kernels here compute (smr_sw_pk.hpp, smr_walk.hpp, sw_wave_r):
//   * E of a cell is stored before the lazy-F loop has raised the cell's H (ssw.c:267,496): a gap in one sequence directly behind a gap in the
//     other that crosses a segment boundary is not seen -- optimal only when 2 * gap < |mismatch|;
//   * the 16-bit kernel leaves its lazy-F loop as soon as no lane has F - gap_ext > H - gap_open (ssw.c:496-507): with gap_open <= gap_ext that
//     is the case one cell behind a segment boundary, and a longer gap across the boundary is lost.
// Under the schemes where either can happen (scheme_unsupported in smr_engine.hip; the reference's command line accepts them) the reference's
// answer is a deterministic function of the stripe geometry, and this file reproduces it: a 16-lane row of the wave IS the SSE2 register -- lane
// k holds the k-th element of every vector, `_mm_slli_si128` is a lane shift, the per-segment arrays (H of the column being stored and of the one
// before, E, the H column of the best score) live in a per-wave scratch row in global memory (each lane only ever touches its own elements, so no
// barrier is needed), `_mm_movemask_epi8` is a ballot.  One problem per wave, ~30 instructions per cell row of 16: an order of magnitude slower
// than the packed kernels, and only selected when the scheme asks for it (DParams::sw_mode < 0).  ssw_align's flow around it (ssw.c:834-918):
// the 8-bit kernel first; when its score saturates (255) the 16-bit kernel; the reverse pass for the begin cell with the kernel the forward pass
// ended with, stopping at the column that reaches the forward score (`terminate`).
//
// rdq / rfq: read / reference letters 0..4 (any address space the caller has them in), taken at rd0 + rdstep * p and rf0 + rfstep * t as in the
// other kernels (reverse pass: the reversed prefixes, steps -1).  scr: 5 * 16 * ceil(m / 8) uint16 of scratch of this wave.
// Returns what sw_sse2_byte / sw_sse2_word return: score, end column t (in processing order), end row p (smallest read position in the best
// column's H that holds the score, ssw.c:305-336).
struct StripedEnd { int score, end_t, end_p; };

template <bool WORD>
__device__ __forceinline__ StripedEnd sw_striped_pass(const uint8_t* rdq, int m, int rd0, int rdstep, const uint8_t* rfq, int n, int rf0, int rfstep,
                                                      int match, int mismatch, int scoreN, int go, int ge, int bias, int terminate, uint16_t* scr) {
  constexpr int NL = WORD ? 8 : 16;
  const int lane = lane_id();
  const bool act = lane < NL;
  const int segLen = (m + NL - 1) / NL;
  const size_t vb = (size_t)segLen * 16;
  uint16_t* Q = scr;                                       // the lane's read letters, segment by segment (5 = beyond the read)
  uint16_t* HA = scr + vb; uint16_t* HB = scr + 2 * vb; uint16_t* E = scr + 3 * vb; uint16_t* HM = scr + 4 * vb;
  if (act) for (int j = 0; j < segLen; j++) {
    const int p = j + lane * segLen;
    Q[(size_t)j * 16 + lane] = p < m ? rdq[rd0 + rdstep * p] : 5;
    HA[(size_t)j * 16 + lane] = 0; HB[(size_t)j * 16 + lane] = 0; E[(size_t)j * 16 + lane] = 0; HM[(size_t)j * 16 + lane] = 0;
  }
  uint16_t* HStore = HA; uint16_t* HLoad = HB;
  // saturating arithmetic of the two element types on ints: unsigned 8-bit / signed 16-bit adds, unsigned subtracts, (signed) maxima
  auto adds = [](int a, int b) -> int { const int s = a + b; return WORD ? (s > 32767 ? 32767 : (s < -32768 ? -32768 : s)) : (s > 255 ? 255 : s); };
  auto subs = [](int a, int b) -> int { return a > b ? a - b : 0; };
  auto lane_max = [&](int v, int idle) -> int {            // maximum over the NL lanes of the vector
    v = act ? v : idle;
    for (int d = NL / 2; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return __shfl(v, 0, 64);
  };
  auto shift1 = [&](int v) -> int { const int u = __shfl_up(v, 1, 64); return lane == 0 ? 0 : u; };      // _mm_slli_si128 by one element
  int vMaxScore = 0, vMaxMark = 0, best = 0;
  int end_i = WORD ? 0 : -1, end_p = m - 1;
  bool saturated = false;
  for (int t = 0; t < n; t++) {
    const int r = rfq[rf0 + rfstep * t];
    int vF = 0, vMaxColumn = 0;
    int vH = shift1(act ? (int)(WORD ? (int16_t)HStore[(size_t)(segLen - 1) * 16 + lane] : HStore[(size_t)(segLen - 1) * 16 + lane]) : 0);
    { uint16_t* pv = HLoad; HLoad = HStore; HStore = pv; }
    if (act) for (int j = 0; j < segLen; j++) {
      const size_t o = (size_t)j * 16 + lane;
      const int q = Q[o];
      const int sc = q == 5 ? 0 : ((q == 4 || r == 4) ? scoreN : (q == r ? match : mismatch));      // Read::initScoringMatrix (read.cpp:274-288); 0 beyond the read (qP_*: ssw.c:116-141, 375-397)
      int h = WORD ? adds(vH, sc) : subs(adds(vH, sc + bias), bias);
      int e = E[o];
      h = max(h, e); h = max(h, vF);
      vMaxColumn = max(vMaxColumn, h);
      HStore[o] = (uint16_t)h;
      h = subs(h, go);
      e = max(subs(e, ge), h);
      E[o] = (uint16_t)e;
      vF = max(subs(vF, ge), h);
      vH = WORD ? (int)(int16_t)HLoad[o] : (int)HLoad[o];
    }
    if (!WORD) {
      // lazy F of the 8-bit kernel (ssw.c:267-299): around the segments until no lane's F can still raise an H
      int j = 0;
      vF = shift1(vF);
      for (;;) {
        const int h0 = act ? (int)HStore[(size_t)j * 16 + lane] : 0;
        if (!__any(act && subs(vF, subs(h0, go)) != 0)) break;
        if (act) { const int h = max(h0, vF); vMaxColumn = max(vMaxColumn, h); HStore[(size_t)j * 16 + lane] = (uint16_t)h; vF = subs(vF, ge); }
        if (++j >= segLen) { j = 0; vF = shift1(vF); }
      }
    } else {
      // ... of the 16-bit kernel (ssw.c:496-507): at most once around per lane, and out as soon as no lane has F - gap_ext > H - gap_open
      bool done = false;
      for (int kk = 0; kk < 8 && !done; kk++) {
        vF = shift1(vF);
        for (int j = 0; j < segLen; j++) {
          bool more = false;
          if (act) {
            const size_t o = (size_t)j * 16 + lane;
            int h = max((int)(int16_t)HStore[o], vF);
            HStore[o] = (uint16_t)h;
            h = subs(h, go);
            vF = subs(vF, ge);
            more = vF > h;
          }
          if (!__any(more)) { done = true; break; }
        }
      }
    }
    vMaxScore = max(vMaxScore, vMaxColumn);
    if (__any(act && vMaxScore != vMaxMark)) {
      vMaxMark = vMaxScore;
      const int temp = lane_max(vMaxScore, WORD ? -32768 : 0);
      if ((WORD ? (int)(uint16_t)temp : temp) > best) {
        best = WORD ? (int)(uint16_t)temp : temp;
        if (!WORD && best + bias >= 255) { saturated = true; break; }
        end_i = t;
        if (act) for (int j = 0; j < segLen; j++) HM[(size_t)j * 16 + lane] = HStore[(size_t)j * 16 + lane];
      }
    }
    const int mc = lane_max(vMaxColumn, WORD ? -32768 : 0);
    if ((WORD ? (int)(uint16_t)mc : mc) == terminate) break;
  }
  // the smallest read position of the best column that holds the score
  int mine = m - 1;
  if (act) for (int j = 0; j < segLen; j++) if ((int)HM[(size_t)j * 16 + lane] == best) { const int p = j + lane * segLen; if (p < mine) mine = p; }
  mine = act ? mine : m - 1;
  for (int d = NL / 2; d > 0; d >>= 1) mine = min(mine, __shfl_xor(mine, d, 64));
  end_p = __shfl(mine, 0, 64);
  StripedEnd out;
  out.score = !WORD && saturated ? 255 : best;
  out.end_t = end_i; out.end_p = end_p;
  return out;
}

// forward pass (rdstep > 0): ssw_align's 8-bit kernel, then the 16-bit one if that saturated -- `word` tells which produced the result; reverse
// pass (rdstep < 0): the kernel the forward pass ended with (`word` in), stopping where the column maximum reaches `terminate` = the forward score
__device__ __attribute__((noinline)) SwRes sw_wave_striped(const uint8_t* rdq, int m, int rd0, int rdstep, const uint8_t* rfq, int n, int rf0, int rfstep,
                                                            int match, int mismatch, int scoreN, int go, int ge, int terminate, int& word, uint16_t* scr) {
  int bias = 0;                                            // the most negative entry of the 5 x 5 matrix (ssw_init, ssw.c:790-800)
  bias = min(bias, min(mismatch, min(scoreN, match)));
  bias = -bias;
  StripedEnd e;
  if (rdstep > 0) {
    e = sw_striped_pass<false>(rdq, m, rd0, rdstep, rfq, n, rf0, rfstep, match, mismatch, scoreN, go, ge, bias, 255, scr);
    word = 0;
    if (e.score == 255) { e = sw_striped_pass<true>(rdq, m, rd0, rdstep, rfq, n, rf0, rfstep, match, mismatch, scoreN, go, ge, bias, 0xFFFF, scr); word = 1; }
  } else {
    if (word) e = sw_striped_pass<true>(rdq, m, rd0, rdstep, rfq, n, rf0, rfstep, match, mismatch, scoreN, go, ge, bias, terminate & 0xFFFF, scr);
    else e = sw_striped_pass<false>(rdq, m, rd0, rdstep, rfq, n, rf0, rfstep, match, mismatch, scoreN, go, ge, bias, terminate & 0xFF, scr);
  }
  SwRes r;
  r.score = e.score; r.end_ref = e.end_t; r.end_read = e.end_p;
  return r;
}

}  // namespace smr
