// smr_trace.hpp -- part of the HIP kernels of libsmr_hip (included by smr_kernels.hpp).
#pragma once

namespace smr {

// ------------------------------------------------------------------------------------------------
// k_trace: banded_sw (ssw.c:577-773), one thread per stored alignment.  Scratch per wave slot:
//   dir  : bytes, element e of lane l at dir[e*64 + l]
//   hbuf : ints,  3 arrays (h_b, e_b, h_c) of `wcap` ints, element e of lane l at (a*wcap + e)*64 + l
// ------------------------------------------------------------------------------------------------
#define TR_SET_U(u, w, i, j) { int x_ = (i) - (w); x_ = x_ > 0 ? x_ : 0; (u) = (j) - x_ + 1; }
#define TR_SET_D(u, w, i, j, p) { int x_ = (i) - (w); x_ = x_ > 0 ? x_ : 0; x_ = (j) - x_; (u) = x_ * 3 + (p); }

__global__ void __launch_bounds__(64) k_trace(DReads rd, DIndex ix, DParams P, const uint32_t* __restrict__ tasks, uint32_t n_tasks,
                                              AlignRec* __restrict__ aln, uint32_t* __restrict__ cigar_pool, uint32_t cigar_words,
                                              unsigned long long* __restrict__ ctr, int8_t* g_dir, int* g_hbuf, uint32_t* g_cig,
                                              uint64_t dir_cap, uint32_t wcap, uint32_t cig_cap) {
  const int lane = lane_id();
  int8_t* dir = g_dir + (size_t)blockIdx.x * dir_cap * 64;
  int* hb = g_hbuf + (size_t)blockIdx.x * 3 * wcap * 64;
  uint32_t* cg = g_cig + (size_t)blockIdx.x * cig_cap * 64;
#define DIRX(e) dir[(size_t)(e) * 64 + lane]
#define HB(a, e) hb[((size_t)(a) * wcap + (e)) * 64 + lane]
#define CG(e) cg[(size_t)(e) * 64 + lane]
  for (uint32_t tb = blockIdx.x * 64; tb < n_tasks; tb += gridDim.x * 64) {
    uint32_t t = tb + lane;
    if (t >= n_tasks) continue;
    const uint32_t slot = tasks[t];
    AlignRec al = aln[slot];
    const uint32_t r = slot / P.slots;
    const uint32_t len = rd.len[r];
    const uint32_t* rec = rd.words + rd.rec_off[r];
    const uint32_t reversed = al.strand ? 0u : 1u;
    const uint8_t* ref = ix.ref_seq + ix.ref_off[al.ref_num] + al.ref_begin1;
    const int refLen = al.ref_end1 - al.ref_begin1 + 1;
    const int readLen = al.read_end1 - al.read_begin1 + 1;
    const int score = al.score1;
    const int gapO = P.gap_open, gapE = P.gap_ext;
    int band_width = abs(refLen - readLen) + 1;
    int i, j, e, f, temp1, temp2, l, mx = 0, width, width_d;
    bool fail = false;
    size_t dl = 0;                                         // direction_line offset (elements)
    // h_b / e_b / h_c start zeroed (calloc-like); stale values across band doublings are kept like the reference
    for (uint32_t q = 0; q < wcap; q++) { HB(0, q) = 0; HB(1, q) = 0; HB(2, q) = 0; }
    do {
      width = band_width * 2 + 3; width_d = band_width * 2 + 1;
      if ((uint32_t)width + 2 > wcap || (uint64_t)width_d * readLen * 3 + 8 > dir_cap) { fail = true; break; }
      for (j = 1; j < width - 1; j++) HB(0, j) = 0;
      for (i = 0; i < readLen; i++) {
        int beg = 0, end = refLen - 1, u = 0, edge;
        j = i - band_width; beg = beg > j ? beg : j;
        j = i + band_width; end = end < j ? end : j;
        edge = end + 1 < width - 1 ? end + 1 : width - 1;
        f = 0; HB(0, 0) = 0; HB(1, 0) = 0; HB(0, edge) = 0; HB(1, edge) = 0; HB(2, 0) = 0;
        dl = (size_t)width_d * i * 3;
        const int rnt = (int)read_nt(rec, len, (uint32_t)(al.read_begin1 + i), reversed, 4u);
        for (j = beg; j <= end; j++) {
          int b, e1, f1, d, de, df, dh;
          TR_SET_U(u, band_width, i, j); TR_SET_U(e, band_width, i - 1, j);
          TR_SET_U(b, band_width, i, j - 1); TR_SET_U(d, band_width, i - 1, j - 1);
          TR_SET_D(de, band_width, i, j, 0);
          TR_SET_D(df, band_width, i, j, 1);
          TR_SET_D(dh, band_width, i, j, 2);
          temp1 = i == 0 ? -gapO : HB(0, e) - gapO;
          temp2 = i == 0 ? -gapE : HB(1, e) - gapE;
          int eb = temp1 > temp2 ? temp1 : temp2;
          HB(1, u) = eb;
          int8_t dde = temp1 > temp2 ? 3 : 2;
          DIRX(dl + de) = dde;
          temp1 = HB(2, b) - gapO;
          temp2 = f - gapE;
          f = temp1 > temp2 ? temp1 : temp2;
          int8_t ddf = temp1 > temp2 ? 5 : 4;
          DIRX(dl + df) = ddf;
          e1 = eb > 0 ? eb : 0;
          f1 = f > 0 ? f : 0;
          temp1 = e1 > f1 ? e1 : f1;
          const int fnt = ref[j];
          const int sc = (fnt == 4 || rnt == 4) ? P.score_N : (fnt == rnt ? P.match : P.mismatch);
          temp2 = HB(0, d) + sc;
          int hc = temp1 > temp2 ? temp1 : temp2;
          HB(2, u) = hc;
          if (hc > mx) mx = hc;
          if (temp1 <= temp2) DIRX(dl + dh) = 1;
          else DIRX(dl + dh) = e1 > f1 ? dde : ddf;
        }
        for (j = 1; j <= u; j++) HB(0, j) = HB(2, j);
      }
      band_width *= 2;
    } while (mx < score);
    uint32_t clen = 0, coff = 0;
    if (!fail) {
      band_width /= 2;
      // trace back (ssw.c:674-747); dl points at the last row
      i = readLen - 1; j = refLen - 1; e = 0; l = 0; f = 0; mx = 0; temp2 = 2;
      while (i > 0) {
        if (j < 0) { fail = true; break; }
        TR_SET_D(temp1, band_width, i, j, temp2);
        if (temp1 < 0 || temp1 >= width_d * 3) { fail = true; break; }
        int dv = DIRX(dl + temp1);
        switch (dv) {
          case 1: --i; --j; temp2 = 2; dl -= (size_t)width_d * 3; f = 0; break;
          case 2: --i; temp2 = 0; dl -= (size_t)width_d * 3; f = 1; break;
          case 3: --i; temp2 = 2; dl -= (size_t)width_d * 3; f = 1; break;
          case 4: --j; temp2 = 1; f = 2; break;
          case 5: --j; temp2 = 2; f = 2; break;
          default: fail = true; i = 0; break;
        }
        if (fail) break;
        if (f == mx) ++e;
        else {
          ++l;
          if ((uint32_t)l + 3 >= cig_cap) { fail = true; break; }
          CG(l - 1) = (uint32_t)e << 4 | (uint32_t)mx;
          mx = f; e = 1;
        }
      }
      if (!fail) {
        if ((uint32_t)l + 3 >= cig_cap) fail = true;
        else if (f == 0) { ++l; CG(l - 1) = (uint32_t)(e + 1) << 4; }
        else { l += 2; CG(l - 2) = (uint32_t)e << 4 | (uint32_t)f; CG(l - 1) = 16; }
      }
      if (!fail) {
        clen = (uint32_t)l;
        unsigned long long old = atomicAdd(&ctr[C_CIGAR_CURSOR], (unsigned long long)clen);
        if (old + clen > cigar_words) { atomicAdd(&ctr[C_ERR_CIGAR], 1ull); fail = true; }
        else { coff = (uint32_t)old; for (uint32_t q = 0; q < clen; q++) cigar_pool[coff + q] = CG(clen - 1 - q); }
      }
    }
    if (fail) { atomicAdd(&ctr[C_ERR_TRACE], 1ull); }
    else { al.has_cigar = 1; al.cigar_off = coff; al.cigar_len = clen; aln[slot] = al; }
  }
#undef DIRX
#undef HB
#undef CG
}

}  // namespace smr
