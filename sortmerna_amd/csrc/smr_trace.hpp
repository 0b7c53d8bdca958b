// smr_trace.hpp -- part of the HIP kernels of libsmr_hip (included by smr_kernels.hpp).
//
// CIGARs of the stored alignments.  What has to come out is what the reference's banded_sw
// (/root/reference/src/sortmerna/ssw.c:577-773) produces for the same (reference window, read window, score): an affine-gap DP
// restricted to the diagonals |j - i| <= band around the window's main diagonal, band doubled until the DP reaches the known score,
// then a walk from the window's last cell back to read row 0, run-length encoded.  How it is computed here is not how ssw.c does it:
//
//  * Lanes lie ACROSS the band: lane k owns diagonal k (reference column j = i + k - band of read row i), so the three
//    neighbours of a cell are the own lane's previous value (diagonal), the lane to the right in the previous row (gap in the
//    read, E) and the lane to the left in the same row (gap in the reference, F).  One DP row is one step.
//  * The serial F chain of a row is a max-plus prefix scan: with G(k) = F(k) + k*gap_ext,
//        G(k) = max( G_init, max_{k' < k} ( H(k') - gap_open + (k'+1)*gap_ext ) ),
//    an (exclusive) prefix maximum over the lanes -- DPP row shifts inside 16 lanes, row broadcasts across them.  For
//    gap_open >= gap_ext the scan over H' = max(E, diagonal) already yields the exact F (an H that came from F never opens a
//    better gap than extending that F); otherwise the scan is repeated on the updated H until nothing changes.
//  * Direction flags are 4 bits per cell (2: where H came from; 1: E opened or extended; 1: F opened or extended), kept in LDS
//    by the narrow kernel and in a compact global tile by the wide one; one lane walks them back.
//  * k_trace_band<G>: 64/G alignments per wave, G = 8 or 16 lanes each (bands up to 3 / 7: practically every short-read
//    alignment), DP state in registers.  k_trace_wide: one wave per alignment, the band in strips of 64 diagonals with the scan
//    carried from strip to strip, DP rows in LDS (or in global memory when the band is too wide for it).
//  * Alignments whose band outgrows a kernel are handed to the next one through a task list, with the band to start from.
//
// Behaviour that is the reference's and is reproduced on purpose (each checked against its records by the golden tests):
//  out-of-band neighbours count as H = E = F = 0, not -inf (ssw.c:627,642-650); the cell above the window's last reference column
//  is also read as 0 while the band still starts at column 0 and already reaches past the window's end (the `edge` slot of
//  ssw.c:626-627 then coincides with that column; it only matters when the walk starts with a gap in the reference); ties prefer the diagonal, then E over F only when strictly larger
//  (ssw.c:655-664); the walk stops at read row 0 wherever it is and counts that cell as one more M (ssw.c:729-747), and a walk
//  that starts with a gap leaves a leading 0M run.
#pragma once

namespace smr {

#define TR_NEG (-(1 << 29))
#define TR_CIG_STAGE 24          // CIGAR runs staged in LDS per alignment before pool space is claimed (longer CIGARs are walked twice)

template <int CTRL> __device__ __forceinline__ int tr_dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, 0xF, false); }

// inclusive prefix maximum inside groups of G (8 or 16) lanes: row_shr:1/2/4/8
template <int G> __device__ __forceinline__ int tr_group_scan_max(int x, int gl) {
  int v;
  v = tr_dpp<0x111>(TR_NEG, x); if (gl >= 1) x = max(x, v);
  v = tr_dpp<0x112>(TR_NEG, x); if (gl >= 2) x = max(x, v);
  v = tr_dpp<0x114>(TR_NEG, x); if (gl >= 4) x = max(x, v);
  if (G > 8) { v = tr_dpp<0x118>(TR_NEG, x); if (gl >= 8) x = max(x, v); }
  return x;
}
// ... over the 64 lanes: in-row shifts, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3
__device__ __forceinline__ int tr_wave_scan_max(int x) {
  x = max(x, tr_dpp<0x111>(TR_NEG, x));
  x = max(x, tr_dpp<0x112>(TR_NEG, x));
  x = max(x, tr_dpp<0x114>(TR_NEG, x));
  x = max(x, tr_dpp<0x118>(TR_NEG, x));
  x = max(x, __builtin_amdgcn_update_dpp(TR_NEG, x, 0x142, 0xA, 0xF, false));
  x = max(x, __builtin_amdgcn_update_dpp(TR_NEG, x, 0x143, 0xC, 0xF, false));
  return x;
}

// What one band cell needs from its three neighbours, and what it leaves behind.  hup/eup: H and E of the cell above, hd: H of the
// diagonal neighbour, sc: substitution score.  Sets e (E of this cell), de (E opened from H), e1 = max(E, 0), t2 = diagonal path.
struct TrCell { int e, de, e1, t2, hq; };
__device__ __forceinline__ TrCell tr_cell_open(int hup, int eup, int hd, int sc, int go, int ge) {
  TrCell c;
  const int eo = hup - go, ee = eup - ge;
  c.e = max(eo, ee); c.de = eo > ee ? 1 : 0;
  c.e1 = max(c.e, 0); c.t2 = hd + sc;
  c.hq = max(c.e1, c.t2);                      // H without the F path
  return c;
}
// flag nibble: bits 0-1 where H came from (0 diagonal, 1 E, 2 F), bit 2 E opened, bit 3 F opened
__device__ __forceinline__ int tr_cell_flags(const TrCell& c, int f1, int df) {
  const int t1 = max(c.e1, f1);
  const int hsel = t1 <= c.t2 ? 0 : (c.e1 > f1 ? 1 : 2);
  return hsel | c.de << 2 | df << 3;
}

// The walk back from the window's last cell over the flags (get(i, k) -> nibble), emitting the runs in walk order (the CIGAR is
// their reverse).  Returns the number of runs, or -1 when the path leaves the band or the window.
template <class Get, class Emit>
__device__ __forceinline__ int tr_walk(int readLen, int refLen, int bw, Get get, Emit emit) {
  int i = readLen - 1, k = refLen - readLen + bw, state = 0, run_op = 0, run = 0, n = 0, last = 0;
  while (i > 0) {
    if (k < 0 || k > 2 * bw || i + k - bw < 0) return -1;
    const int nib = get(i, k);
    const int sel = state ? state : (nib & 3);
    int op;
    if (sel == 0) { op = 0; --i; state = 0; }
    else if (sel == 1) { op = 1; state = (nib & 4) ? 0 : 1; --i; ++k; }
    else { op = 2; state = (nib & 8) ? 0 : 2; --k; }
    if (op == run_op) ++run;
    else { emit(n++, (uint32_t)run << 4 | (uint32_t)run_op); run_op = op; run = 1; }
    last = op;
  }
  if (last == 0) emit(n++, (uint32_t)(run + 1) << 4);
  else { emit(n++, (uint32_t)run << 4 | (uint32_t)last); emit(n++, 16u); }
  return n;
}

// claim pool space for n runs, write them reversed; staged runs when they fit, else a second walk writes in place
template <class Get>
__device__ __forceinline__ bool tr_finish(int readLen, int refLen, int bw, Get get, uint32_t* stage, AlignRec& al, uint32_t* __restrict__ cigar_pool,
                                          uint32_t cigar_words, unsigned long long* __restrict__ ctr) {
  const int n = tr_walk(readLen, refLen, bw, get, [&](int q, uint32_t v) { if (q < TR_CIG_STAGE) stage[q] = v; });
  if (n <= 0) { atomicAdd(&ctr[C_ERR_TRACE], 1ull); return false; }
  const unsigned long long old = atomicAdd(&ctr[C_CIGAR_CURSOR], (unsigned long long)n);
  if (old + (unsigned long long)n > cigar_words) { atomicAdd(&ctr[C_ERR_CIGAR], 1ull); return false; }
  uint32_t* out = cigar_pool + old;
  if (n <= TR_CIG_STAGE) { for (int q = 0; q < n; q++) out[n - 1 - q] = stage[q]; }
  else tr_walk(readLen, refLen, bw, get, [&](int q, uint32_t v) { out[n - 1 - q] = v; });
  al.has_cigar = 1; al.cigar_off = (uint32_t)old; al.cigar_len = (uint32_t)n;
  return true;
}

// ------------------------------------------------------------------------------------------------
// k_trace_band<G>: 64/G alignments per wave.  Dynamic LDS: flags[row_pairs][64] bytes (byte [p][lane] = rows 2p | 2p+1 << 4 of the
// lane's diagonal) | read windows [64/G][lds_ml] | reference windows [64/G][lds_rf] | staged runs [64/G][TR_CIG_STAGE] u32
// ------------------------------------------------------------------------------------------------
template <int G>
__global__ void __launch_bounds__(64) k_trace_band(DReads rd, DIndex ix, DParams P, const uint32_t* __restrict__ tasks, uint32_t n_tasks,
                                                   AlignRec* __restrict__ aln, uint32_t* __restrict__ cigar_pool, uint32_t cigar_words,
                                                   unsigned long long* __restrict__ ctr, uint32_t* __restrict__ tasks_out,
                                                   uint32_t lds_ml, uint32_t lds_rf, uint32_t row_pairs) {
  constexpr int NG = 64 / G;
  SMR_DYN_LDS(unsigned char, lds_raw);
  uint8_t* flags = lds_raw;
  const int lane = lane_id(), g = lane / G, gl = lane % G;
  uint8_t* rdq = flags + (size_t)row_pairs * 64 + (size_t)g * lds_ml;
  uint8_t* rfq = flags + (size_t)row_pairs * 64 + (size_t)NG * lds_ml + (size_t)g * lds_rf;
  uint32_t* stage = (uint32_t*)(flags + (size_t)row_pairs * 64 + (size_t)NG * (lds_ml + lds_rf)) + g * TR_CIG_STAGE;
  const int go = P.gap_open, ge = P.gap_ext;
  const int f_init = max(-go, -ge);
  for (uint32_t tb = blockIdx.x * NG; tb < n_tasks; tb += gridDim.x * NG) {
    const uint32_t t = tb + g;
    const bool have = t < n_tasks;
    uint32_t slot = 0;
    AlignRec al;
    int refLen = 0, readLen = 0, score = 0, bw = 1, state = 2;        // state 0: band DP to run, 1: score reached, 2: not for this kernel
    __syncthreads();
    if (have) {
      slot = tasks[t]; al = aln[slot];
      refLen = al.ref_end1 - al.ref_begin1 + 1; readLen = al.read_end1 - al.read_begin1 + 1; score = al.score1;
      bw = max(abs(refLen - readLen) + 1, (int)al.cigar_len);           // cigar_len of an alignment without CIGAR: the band an earlier kernel stopped at
      if ((uint32_t)readLen <= lds_ml && (uint32_t)refLen <= lds_rf && (uint32_t)(readLen + 1) / 2 <= row_pairs) {
        state = 0;
        const uint32_t r = slot / P.slots, len = rd.len[r];
        const uint32_t* rec = rd.words + rd.rec_off[r];
        const uint8_t* ref = ix.ref_seq + ix.ref_off[al.ref_num] + al.ref_begin1;
        for (int q = gl; q < readLen; q += G) rdq[q] = (uint8_t)read_nt(rec, len, (uint32_t)(al.read_begin1 + q), al.strand ? 0u : 1u, 4u);
        for (int q = gl; q < refLen; q += G) rfq[q] = ref[q];
      }
    }
    __syncthreads();
    int mx = 0;
    for (;;) {
      if (state == 0 && 2 * bw + 1 > G) state = 2;
      const bool run = state == 0;
      if (!__any(run)) break;
      int rows = run ? readLen : 0;
      for (int d = 32; d > 0; d >>= 1) rows = max(rows, __shfl_xor(rows, d, 64));
      int hp = 0, ep = 0, nib_even = 0;                                 // H, E of the lane's diagonal in the previous row
      for (int i = 0; i < rows; i++) {
        const int j = i + gl - bw;
        const bool cell = run && i < readLen && gl <= 2 * bw && j >= 0 && j < refLen;
        int hup = tr_dpp<0x101>(0, hp), eup = tr_dpp<0x101>(0, ep);     // row_shl:1 -- diagonal k+1 of the previous row is the cell above
        const bool edge_col = i <= bw + 1 && refLen - 1 < i + bw && j == refLen - 1;
        if (gl >= 2 * bw || i == 0 || edge_col) { hup = 0; eup = 0; }
        const int hd = (i > 0 && j > 0) ? hp : 0;
        const int rnt = cell ? rdq[i] : 4, fnt = cell ? rfq[j] : 4;
        const int sc = (rnt == 4 || fnt == 4) ? P.score_N : (rnt == fnt ? P.match : P.mismatch);
        const TrCell c = tr_cell_open(hup, eup, hd, sc, go, ge);
        const int klo = max(0, bw - i);                                 // first diagonal of the row inside the window
        int hc = c.hq, f, f1;
        for (;;) {
          const int hl = tr_dpp<0x111>(0, hc);                          // row_shr:1 -- the left neighbour
          int x = !cell ? TR_NEG : (gl == klo ? f_init + klo * ge : hl - go + gl * ge);
          x = tr_group_scan_max<G>(x, gl);
          f = x - gl * ge; f1 = max(f, 0);
          const int hn = max(c.hq, f1);
          const bool changed = cell && hn != hc;
          hc = hn;
          if (go >= ge || !__any(changed)) break;
        }
        const int hl = tr_dpp<0x111>(0, hc), fl = tr_dpp<0x111>(0, f);
        const int df = gl == klo ? (-go > -ge ? 1 : 0) : (hl - go > fl - ge ? 1 : 0);
        const int nib = tr_cell_flags(c, f1, df);
        if (cell) mx = max(mx, hc);
        hp = cell ? hc : 0; ep = cell ? c.e : 0;
        if (i & 1) { if (run) flags[(size_t)(i >> 1) * 64 + lane] = (uint8_t)(nib_even | nib << 4); }
        else nib_even = nib;
      }
      if ((rows & 1) && run) flags[(size_t)(rows >> 1) * 64 + lane] = (uint8_t)nib_even;
      int gm = mx;
      for (int d = G / 2; d > 0; d >>= 1) gm = max(gm, __shfl_xor(gm, d, G));
      if (run) { if (gm >= score) state = 1; else bw *= 2; }
      __syncthreads();
    }
    if (gl == 0 && have) {
      if (state == 1) {
        const uint8_t* fg = flags + g * G;
        auto get = [&](int i, int k) { return (int)(fg[(size_t)(i >> 1) * 64 + k] >> ((i & 1) * 4)) & 15; };
        if (tr_finish(readLen, refLen, bw, get, stage, al, cigar_pool, cigar_words, ctr)) aln[slot] = al;
      } else {
        aln[slot].cigar_len = (uint32_t)bw;
        tasks_out[atomicAdd(&ctr[C_TRACE_DEFER], 1ull)] = slot;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_trace_wide: one wave per alignment, bands up to band_cap.  DP rows (H and E of the previous row, updated in place strip by
// strip, left to right: a strip reads its right neighbour's first diagonal before that strip overwrites it): 2 x wcap ints, in
// dynamic LDS (ROWS_LDS) or at g_rows + block * 2 * wcap.  Flags: g_flags + block * flags_cap bytes, row i at i * (wp / 2), two
// neighbouring diagonals per byte.  The kernel lives on waves in flight (a letter of the read per DP row and a reference letter per lane and
// strip come from global memory inside the row loop): 32 blocks per CU, 64 VGPRs; staging both sequences in LDS first (19 KB per block: 8 blocks
// per CU) was slower than leaving that latency to the other waves (round 4, profiles/r4s10_*).
// ------------------------------------------------------------------------------------------------
template <bool ROWS_LDS>
__global__ void __launch_bounds__(64, 8) k_trace_wide(DReads rd, DIndex ix, DParams P, const uint32_t* __restrict__ tasks, uint32_t n_tasks,
                                                   AlignRec* __restrict__ aln, uint32_t* __restrict__ cigar_pool, uint32_t cigar_words,
                                                   unsigned long long* __restrict__ ctr, uint32_t* __restrict__ tasks_out, int band_cap,
                                                   uint8_t* __restrict__ g_flags, unsigned long long flags_cap, int* g_rows, uint32_t wcap) {
  SMR_DYN_LDS(unsigned char, lds_raw);
  uint32_t* stage = (uint32_t*)lds_raw;
  int* rowH = ROWS_LDS ? (int*)(lds_raw + TR_CIG_STAGE * 4) : g_rows + (size_t)blockIdx.x * 2 * wcap;
  int* rowE = rowH + wcap;
  uint8_t* fl = g_flags + (size_t)blockIdx.x * flags_cap;
  const int lane = lane_id();
  const int go = P.gap_open, ge = P.gap_ext;
  const int f_init = max(-go, -ge);
  for (uint32_t t = blockIdx.x; t < n_tasks; t += gridDim.x) {
    const uint32_t slot = tasks[t];
    AlignRec al = aln[slot];
    const int refLen = al.ref_end1 - al.ref_begin1 + 1, readLen = al.read_end1 - al.read_begin1 + 1, score = al.score1;
    const uint32_t r = slot / P.slots, len = rd.len[r];
    const uint32_t* rec = rd.words + rd.rec_off[r];
    const uint32_t reversed = al.strand ? 0u : 1u;
    const uint8_t* ref = ix.ref_seq + ix.ref_off[al.ref_num] + al.ref_begin1;
    int bw = max(abs(refLen - readLen) + 1, (int)al.cigar_len);
    int mx = 0, state = 0, wp = 64;
    while (state == 0) {
      wp = (2 * bw + 1 + 63) & ~63;
      if (bw > band_cap || (uint32_t)wp > wcap || (unsigned long long)readLen * (wp / 2) > flags_cap) { state = 2; break; }
      __syncthreads();
      for (int k = lane; k < wp; k += 64) { rowH[k] = 0; rowE[k] = 0; }
      __syncthreads();
      for (int i = 0; i < readLen; i++) {
        const int rnt = (int)read_nt(rec, len, (uint32_t)(al.read_begin1 + i), reversed, 4u);
        const int klo = max(0, bw - i), khi = min(2 * bw, refLen - 1 - i + bw);
        const bool edge_row = i <= bw + 1 && refLen - 1 < i + bw;
        int carry = TR_NEG, last_h = 0, last_f = 0;                     // scan prefix and exact H, F of the previous strip's last diagonal
        uint8_t* frow = fl + (size_t)i * (wp / 2);
        for (int k0 = klo & ~63; k0 <= khi; k0 += 64) {
          const int k = k0 + lane, j = i + k - bw;
          const bool cell = k >= klo && k <= khi;
          const int hown = rowH[k];
          int hup = k + 1 < wp ? rowH[k + 1] : 0, eup = k + 1 < wp ? rowE[k + 1] : 0;
          if (k >= 2 * bw || i == 0 || (edge_row && j == refLen - 1)) { hup = 0; eup = 0; }
          const int hd = (i > 0 && j > 0) ? hown : 0;
          const int fnt = cell ? ref[j] : 4;
          const int sc = (rnt == 4 || fnt == 4) ? P.score_N : (rnt == fnt ? P.match : P.mismatch);
          const TrCell c = tr_cell_open(hup, eup, hd, sc, go, ge);
          int hc = c.hq, f, f1, x;
          for (;;) {
            const int hl = tr_dpp<0x138>(last_h, hc);                   // wave_shr:1, lane 0 takes the previous strip's last H
            x = !cell ? TR_NEG : (k == klo ? f_init + klo * ge : hl - go + k * ge);
            x = max(tr_wave_scan_max(x), carry);
            f = x - k * ge; f1 = max(f, 0);
            const int hn = max(c.hq, f1);
            const bool changed = cell && hn != hc;
            hc = hn;
            if (go >= ge || !__any(changed)) break;
          }
          const int hl = tr_dpp<0x138>(last_h, hc), fleft = tr_dpp<0x138>(last_f, f);
          const int df = k == klo ? (-go > -ge ? 1 : 0) : (hl - go > fleft - ge ? 1 : 0);
          const int nib = tr_cell_flags(c, f1, df);
          if (cell) mx = max(mx, hc);
          __syncthreads();                                              // every lane has read the old row values of this strip
          rowH[k] = cell ? hc : 0; rowE[k] = cell ? c.e : 0;
          const int nb = nib | tr_dpp<0x130>(0, nib) << 4;              // wave_shl:1 -- pack two neighbouring diagonals
          if (!(lane & 1)) frow[k >> 1] = (uint8_t)nb;
          carry = __builtin_amdgcn_readlane(x, 63); last_h = __builtin_amdgcn_readlane(hc, 63); last_f = __builtin_amdgcn_readlane(f, 63);
        }
        __syncthreads();
      }
      for (int d = 32; d > 0; d >>= 1) mx = max(mx, __shfl_xor(mx, d, 64));
      if (mx >= score) state = 1; else bw *= 2;
    }
    __threadfence_block();
    __syncthreads();
    if (lane == 0) {
      if (state == 1) {
        const int hw = wp / 2;
        auto get = [&](int i, int k) { return (int)(fl[(size_t)i * hw + (k >> 1)] >> ((k & 1) * 4)) & 15; };
        if (tr_finish(readLen, refLen, bw, get, stage, al, cigar_pool, cigar_words, ctr)) aln[slot] = al;
      } else {
        aln[slot].cigar_len = (uint32_t)bw;
        tasks_out[atomicAdd(&ctr[C_TRACE_DEFER], 1ull)] = slot;
      }
    }
    __syncthreads();
  }
}

}  // namespace smr
