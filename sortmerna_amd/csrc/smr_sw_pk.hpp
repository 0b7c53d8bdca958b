// smr_sw_pk.hpp -- Smith-Waterman score + end cell on packed 16-bit lanes (v_pk_*_i16): the anti-diagonal systolic
// array of smr_chain.hpp with 128 VIRTUAL lanes per wave.  The low half of every 32-bit register belongs to virtual
// lane l, the high half to virtual lane l + 64; virtual lane v owns R consecutive read rows and works on reference
// column t - v at step t, so the high half of lane 0 continues where the low half of lane 63 stopped (one readlane
// per stream and step), and one v_pk op advances two DP cells.  Same recurrence and the same end-cell rule as
// sw_wave_r (ssw.c:150-373, 305-336); results are bit-identical (tests: the packed and the scalar kernel are compared
// with the oracle and with each other, and smr_create checks one against the other on the device).
//
// Representation (all signed 16 bit):  Y = H - gap_open (so e = max(E - ge, Y), f = max(F_up - ge, Y_up)), score table
// T = score + gap_open >= 0 selected per step by ONE v_perm_b32 for both halves (byte 0..3 of the low row's table,
// 4..7 of the high row's; selector 0x0C = constant 0 = "score -gap_open", used for the columns before 0 and after n-1
// and for the rows >= m, which therefore never reach the maximum: every value there is strictly smaller than a valid
// cell's, or 0).  h = max(Y_diag + T, e, f, 0).  Per cell pair: perm, add, 2 x (sub, max), 3 x max, sub = 10 VALU ops
// + 3 for the running maximum key (h << 16 | first column | half), instead of ~20 per single cell.  The inputs of virtual
// lane 0 (reference letters; the boundary row of the previous strip) are loaded 64 columns at a time, one per lane, and
// read back with v_readlane, so the step loop has no memory access.
// Preconditions (checked by the caller): m * match + 255 < 32768, n + 128 <= 8191, gap_open + min(score) >= 0,
// match + gap_open <= 255.
#pragma once

namespace smr {

// (pk16, pk_from / pk_bits / pk_add / pk_sub / pk_max and perm_b32: <smr_device_ops.hpp>)
__device__ __forceinline__ pk16 pk_splat(int v) { return pk_from(((uint32_t)v & 0xFFFFu) * 0x00010001u); }

#define PK_SEL_NONE 0x0C0Cu          // half selector: both bytes constant 0
#define PK_SEL_N 0x0D0Cu             // reference letter N: marker in the high byte (the result is replaced by T_N)

// selector half of reference letter c for the low half; +4 moves it to the high half's table
__device__ __forceinline__ uint32_t pk_sel_of(int c) { return c < 4 ? (0x0C00u | (uint32_t)c) : PK_SEL_N; }
__device__ __forceinline__ uint32_t pk_sel_to_hi(uint32_t s) { return (s & 0xFFu) < 4u ? s + 4u : s; }

// wave_ror:1 -- lane l reads lane l-1, lane 0 reads lane 63 (all lanes have a source, `old` is never used)
__device__ __forceinline__ int dpp_ror1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x13C /* wave_ror:1 */, 0xF, 0xF, false); }

// ROR = true (sw mode 2): the hand-over from lane 63's low half to lane 0's high half stays in the vector unit (wave_ror + one
// select per stream) instead of v_readlane -> SALU -> v_mov in front of a wave_shr; same values, shorter dependent chain per step.
template <int R, bool HASN, bool ROR>
__device__ __forceinline__ SwRes sw_wave_pk_r(const uint8_t* rdq, int m, int rd0, int rdstep, const uint8_t* rfq, int n, int rf0, int rfstep,
                                              int* bound, int match, int mismatch, int scoreN, int go, int ge) {
  const int lane = lane_id();
  int bestH = 0, bestcol = 0x1FFFFF, bestrow = 0x1FFFFF;
  const int rps = 128 * R;
  const int nstrips = (m + rps - 1) / rps;
  const pk16 GE = pk_splat(ge), GO = pk_splat(go), ZERO = pk_splat(0);
  const uint32_t TN = (((uint32_t)(scoreN + go)) & 0xFFFFu) * 0x00010001u;
  for (int s = 0; s < nstrips; s++) {
    const int row0 = s * rps;
    // score tables: byte b of tlo[j] / thi[j] = score(read letter of the row, reference letter b) + gap_open; rows >= m: 0
    uint32_t tlo[R], thi[R];
    pk16 Y[R], E[R];
    uint32_t key[R];
#pragma unroll
    for (int j = 0; j < R; j++) {
      uint32_t t2[2];
      for (int hf = 0; hf < 2; hf++) {
        const int row = row0 + (lane + 64 * hf) * R + j;
        uint32_t t = 0;
        if (row < m) {
          const int c = rdq[rd0 + rdstep * row];
          for (int b = 0; b < 4; b++) t |= (uint32_t)((c == 4 ? scoreN : (c == b ? match : mismatch)) + go) << (8 * b);
        }
        t2[hf] = t;
      }
      tlo[j] = t2[0]; thi[j] = t2[1];
      Y[j] = pk_splat(-go); E[j] = ZERO; key[j] = 0;
    }
    const bool has_prev = s > 0, has_next = s + 1 < nstrips;
    const int vused = has_next ? 128 : (m - row0 + R - 1) / R;       // virtual lanes that own a valid row
    const int steps = n + vused - 1;
    // streams entering a virtual lane from the one above it: Y and F of its last row, and the selector (= reference letter)
    uint32_t lastY = pk_bits(pk_splat(-go)), lastF = 0, selcur = PK_SEL_NONE * 0x00010001u;
    pk16 diag0 = pk_splat(-go);
    // running-maximum key: h << 16 | (0x3FFF - column) << 1 | (1 for the low half): larger = higher score, then earlier column, then
    // the low half (smaller row); 15 bits hold the column term for every column -127 .. 8191, so nothing spills into h
    uint32_t xlo = ((uint32_t)(0x3FFF + lane) << 1) | 1u;
    // the inputs of virtual lane 0 (reference letter; boundary row of the previous strip) for 64 steps = columns t0 .. t0+63, one per lane, read
    // back with v_readlane; the next 64 are fetched before the current 64 are worked on (the letters of a long read's window and the boundary
    // rows come from global memory)
    auto fetch_in = [&](int t0, uint32_t& cS, int& cY, int& cF) {
      const int cq = t0 + lane;
      cS = PK_SEL_NONE; cY = -go; cF = 0;
      if (cq < n) {
        cS = pk_sel_of(rfq[rf0 + rfstep * cq]);
        if (has_prev) { cY = bound[2 * cq] - go; cF = bound[2 * cq + 1]; }
      }
    };
    uint32_t nxS; int nxY, nxF;
    fetch_in(0, nxS, nxY, nxF);
    for (int t0 = 0; t0 < steps; t0 += 64) {
      const uint32_t chS = nxS;
      const int chY = nxY, chF = nxF;
      // (the strip that writes bound[] for the next one runs 127 columns behind its own reads: what it fetches ahead is never what it has yet to write)
      if (t0 + 64 < steps) fetch_in(t0 + 64, nxS, nxY, nxF);
      const int tend = min(64, steps - t0);
      for (int tt = 0; tt < tend; tt++) {
        const uint32_t inS = (uint32_t)__builtin_amdgcn_readlane((int)chS, tt);
        const uint32_t inY = (uint32_t)__builtin_amdgcn_readlane(chY, tt), inF = (uint32_t)__builtin_amdgcn_readlane(chF, tt);
        // virtual lane 64 (high half of lane 0) continues what lane 63's low half produced in the previous step
        pk16 upY, upF;
        if (ROR) {
          // lane 0: low half = the new input, high half = lane 63's low half (for the selector: moved to the high row's table, | 4)
          const uint32_t rY = (uint32_t)dpp_ror1((int)lastY), rF = (uint32_t)dpp_ror1((int)lastF), rS = (uint32_t)dpp_ror1((int)selcur);
          const bool first = lane == 0;
          upY = pk_from(first ? ((rY << 16) | (inY & 0xFFFFu)) : rY);
          upF = pk_from(first ? ((rF << 16) | (inF & 0xFFFFu)) : rF);
          selcur = first ? ((rS << 16) | inS | 0x00040000u) : rS;
        } else {
          const uint32_t y63 = (uint32_t)__builtin_amdgcn_readlane((int)lastY, 63), f63 = (uint32_t)__builtin_amdgcn_readlane((int)lastF, 63),
                         s63 = (uint32_t)__builtin_amdgcn_readlane((int)selcur, 63);
          const uint32_t injY = (y63 << 16) | (inY & 0xFFFFu), injF = (f63 << 16) | (inF & 0xFFFFu),
                         injS = (pk_sel_to_hi(s63 & 0xFFFFu) << 16) | inS;
          upY = pk_from((uint32_t)dpp_shr1((int)injY, (int)lastY));
          upF = pk_from((uint32_t)dpp_shr1((int)injF, (int)lastF));
          selcur = (uint32_t)dpp_shr1((int)injS, (int)selcur);
        }
        uint32_t nmask = 0;
        if (HASN) nmask = ((selcur >> 8) & 0x00010001u) * 0xFFFFu;               // 0xFFFF in the halves whose letter is N
        const uint32_t xhi = xlo + 127u;                                           // the high half is 64 columns behind, flag 0
        pk16 diag = diag0, uy = upY, uf = upF;
#pragma unroll
        for (int j = 0; j < R; j++) {
          uint32_t T = perm_b32(thi[j], tlo[j], selcur);
          if (HASN) T = (T & ~nmask) | (TN & nmask);
          const pk16 a = pk_add(diag, pk_from(T));
          const pk16 e = pk_max(pk_sub(E[j], GE), Y[j]);
          const pk16 f = pk_max(pk_sub(uf, GE), uy);
          const pk16 h = pk_max(pk_max(a, e), pk_max(f, ZERO));
          diag = Y[j];                                      // Y(row, col-1): the diagonal of the next row
          const pk16 y = pk_sub(h, GO);
          Y[j] = y; E[j] = e;
          const uint32_t hu = pk_bits(h);
          key[j] = max(key[j], max((hu << 16) | xlo, (hu & 0xFFFF0000u) | xhi));
          uy = y; uf = f;
        }
        diag0 = upY;                                         // Y(row0 - 1, col): diagonal of the first row at the next column
        lastY = pk_bits(uy); lastF = pk_bits(uf);
        xlo -= 2u;
        if (has_next && lane == 63) {                        // virtual lane 127 hands its last row to the next strip
          const int c127 = t0 + tt - 127;
          if (c127 >= 0 && c127 < n) { bound[2 * c127] = (int)(int16_t)(lastY >> 16) + go; bound[2 * c127 + 1] = (int)(int16_t)(lastF >> 16); }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < R; j++) {
      const int h = (int)(key[j] >> 16);
      const int hf = (key[j] & 1u) ? 0 : 1;
      const int col = 0x3FFF - (int)((key[j] >> 1) & 0x7FFFu);      // the winning half's own column (xlo / xhi above)
      const int row = row0 + (lane + 64 * hf) * R + j;
      if (h > bestH || (h == bestH && h > 0 && (col < bestcol || (col == bestcol && row < bestrow)))) { bestH = h; bestcol = col; bestrow = row; }
    }
    __syncthreads();
  }
  unsigned long long k64 = bestH > 0 ? (((unsigned long long)bestH << 42) | ((unsigned long long)(0x1FFFFF - bestcol) << 21) |
                                        (unsigned long long)(0x1FFFFF - bestrow)) : 0ull;
  k64 = wave_max_u64(k64);
  SwRes rr;
  if (k64 == 0) { rr.score = 0; rr.end_ref = -1; rr.end_read = m - 1; return rr; }
  rr.score = (int)(k64 >> 42);
  rr.end_ref = 0x1FFFFF - (int)((k64 >> 21) & 0x1FFFFF);
  rr.end_read = 0x1FFFFF - (int)(k64 & 0x1FFFFF);
  return rr;
}

// ------------------------------------------------------------------------------------------------
// Four independent problems per wave: each DPP row of 16 lanes is its own systolic array of 32 virtual lanes (low halves =
// virtual lanes 0..15, high halves 16..31, R consecutive read rows each: reads up to 32 R rows, one strip).  A single 150-nt
// problem keeps 75 of the 128 virtual lanes of sw_wave_pk_r busy for n + 74 steps; four of them side by side fill 120 virtual
// lanes for n + 29 steps, i.e. ~2.7 x fewer wave instructions per problem.  Same recurrence, same representation, same end-cell
// rule as sw_wave_pk_r (whose results it must equal bit for bit).  Differences: the hand-over between neighbouring virtual lanes is
// row_ror:1 (lane 0 of a row takes lane 15's low half into its high half), and the reference letters of a row's problem enter at
// its lane 0 from a 16-column register window that moves down by one lane per step (row_shl:1) and is reloaded every 16 steps.
// Every lane passes ITS row's problem: read rdq[rd0 + rdstep * row], m rows; reference rfq[rf0 + rfstep * col], n columns
// (m = 0: the row idles).  The result of a row's problem is returned in all 16 lanes of that row.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int dpp_row_ror1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x121 /* row_ror:1 */, 0xF, 0xF, false); }
__device__ __forceinline__ int dpp_row_shl1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x101 /* row_shl:1 */, 0xF, 0xF, false); }

template <int R, bool HASN>
__device__ __forceinline__ SwRes sw_wave_pk_x4(const uint8_t* rdq, int m, int rd0, int rdstep, const uint8_t* rfq, int n, int rf0, int rfstep,
                                               int match, int mismatch, int scoreN, int go, int ge) {
  const int gl = lane_id() & 15;
  const pk16 GE = pk_splat(ge), GO = pk_splat(go), ZERO = pk_splat(0);
  const uint32_t TN = (((uint32_t)(scoreN + go)) & 0xFFFFu) * 0x00010001u;
  uint32_t tlo[R], thi[R];
  pk16 Y[R], E[R];
  uint32_t key[R];
#pragma unroll
  for (int j = 0; j < R; j++) {
    uint32_t t2[2];
    for (int hf = 0; hf < 2; hf++) {
      const int row = (gl + 16 * hf) * R + j;
      uint32_t t = 0;
      if (row < m) {
        const int c = rdq[rd0 + rdstep * row];
        for (int b = 0; b < 4; b++) t |= (uint32_t)((c == 4 ? scoreN : (c == b ? match : mismatch)) + go) << (8 * b);
      }
      t2[hf] = t;
    }
    tlo[j] = t2[0]; thi[j] = t2[1];
    Y[j] = pk_splat(-go); E[j] = ZERO; key[j] = 0;
  }
  int steps = m > 0 ? n + (m + R - 1) / R - 1 : 0;
  for (int d = 32; d > 0; d >>= 1) steps = max(steps, __shfl_xor(steps, d, 64));
  uint32_t lastY = pk_bits(pk_splat(-go)), lastF = 0, selcur = PK_SEL_NONE * 0x00010001u;
  pk16 diag0 = pk_splat(-go);
  const uint32_t in_y = (uint32_t)(-go) & 0xFFFFu;                  // what enters virtual lane 0 from above: H = 0 (Y = -gap_open), F = 0
  uint32_t xlo = ((uint32_t)(0x3FFF + gl) << 1) | 1u;               // column term of the running-maximum key, as in sw_wave_pk_r
  uint32_t win = PK_SEL_NONE;
  const bool first = gl == 0;
  for (int t = 0; t < steps; t++) {
    if ((t & 15) == 0) { const int cq = t + gl; win = cq < n ? pk_sel_of(rfq[rf0 + rfstep * cq]) : PK_SEL_NONE; }
    const uint32_t rY = (uint32_t)dpp_row_ror1((int)lastY), rF = (uint32_t)dpp_row_ror1((int)lastF), rS = (uint32_t)dpp_row_ror1((int)selcur);
    const pk16 upY = pk_from(first ? ((rY << 16) | in_y) : rY);
    const pk16 upF = pk_from(first ? (rF << 16) : rF);
    selcur = first ? ((rS << 16) | win | 0x00040000u) : rS;
    win = (uint32_t)dpp_row_shl1((int)win);
    uint32_t nmask = 0;
    if (HASN) nmask = ((selcur >> 8) & 0x00010001u) * 0xFFFFu;
    const uint32_t xhi = xlo + 31u;                                 // the high half is 16 columns behind, flag 0
    pk16 diag = diag0, uy = upY, uf = upF;
#pragma unroll
    for (int j = 0; j < R; j++) {
      uint32_t T = perm_b32(thi[j], tlo[j], selcur);
      if (HASN) T = (T & ~nmask) | (TN & nmask);
      const pk16 a = pk_add(diag, pk_from(T));
      const pk16 e = pk_max(pk_sub(E[j], GE), Y[j]);
      const pk16 f = pk_max(pk_sub(uf, GE), uy);
      const pk16 h = pk_max(pk_max(a, e), pk_max(f, ZERO));
      diag = Y[j];
      const pk16 y = pk_sub(h, GO);
      Y[j] = y; E[j] = e;
      const uint32_t hu = pk_bits(h);
      key[j] = max(key[j], max((hu << 16) | xlo, (hu & 0xFFFF0000u) | xhi));
      uy = y; uf = f;
    }
    diag0 = upY;
    lastY = pk_bits(uy); lastF = pk_bits(uf);
    xlo -= 2u;
  }
  int bestH = 0, bestcol = 0x1FFFFF, bestrow = 0x1FFFFF;
#pragma unroll
  for (int j = 0; j < R; j++) {
    const int h = (int)(key[j] >> 16);
    const int hf = (key[j] & 1u) ? 0 : 1;
    const int col = 0x3FFF - (int)((key[j] >> 1) & 0x7FFFu);
    const int row = (gl + 16 * hf) * R + j;
    if (h > bestH || (h == bestH && h > 0 && (col < bestcol || (col == bestcol && row < bestrow)))) { bestH = h; bestcol = col; bestrow = row; }
  }
  unsigned long long k64 = bestH > 0 ? (((unsigned long long)bestH << 42) | ((unsigned long long)(0x1FFFFF - bestcol) << 21) |
                                        (unsigned long long)(0x1FFFFF - bestrow)) : 0ull;
  for (int d = 8; d > 0; d >>= 1) { const unsigned long long o = __shfl_xor(k64, d, 16); k64 = o > k64 ? o : k64; }
  SwRes rr;
  if (k64 == 0) { rr.score = 0; rr.end_ref = -1; rr.end_read = m - 1; return rr; }
  rr.score = (int)(k64 >> 42);
  rr.end_ref = 0x1FFFFF - (int)((k64 >> 21) & 0x1FFFFF);
  rr.end_read = 0x1FFFFF - (int)(k64 & 0x1FFFFF);
  return rr;
}

// the largest read span the four-problem kernel takes, and whether a problem's numbers fit the packed representation at all
#define SW_X4_MAX_ROWS 256
#ifndef SW_X4_INLINE
#define SW_X4_INLINE __attribute__((noinline))
#endif
__host__ __device__ __forceinline__ bool sw_pk_fits(int m, int n, int match, int mismatch, int scoreN, int go) {
  return (long long)m * match + 255 < 32768 && n + 128 <= 8191 && go + mismatch >= 0 && go + scoreN >= 0 && match + go <= 255 && scoreN + go <= 255;
}
// max_m: the longest read span among the wave's problems (wave-uniform), hasn: some reference window holds an N (wave-uniform)
__device__ SW_X4_INLINE SwRes sw_wave_x4(const uint8_t* rdq, int m, int rd0, int rdstep, const uint8_t* rfq, int n, int rf0, int rfstep,
                                                      int match, int mismatch, int scoreN, int go, int ge, int max_m, bool hasn) {
#define X4_ARGS rdq, m, rd0, rdstep, rfq, n, rf0, rfstep, match, mismatch, scoreN, go, ge
  if (hasn) {
    if (max_m <= 96) return sw_wave_pk_x4<3, true>(X4_ARGS);
    if (max_m <= 160) return sw_wave_pk_x4<5, true>(X4_ARGS);
    return sw_wave_pk_x4<8, true>(X4_ARGS);
  }
  if (max_m <= 96) return sw_wave_pk_x4<3, false>(X4_ARGS);
  if (max_m <= 160) return sw_wave_pk_x4<5, false>(X4_ARGS);
  return sw_wave_pk_x4<8, false>(X4_ARGS);
#undef X4_ARGS
}

}  // namespace smr
