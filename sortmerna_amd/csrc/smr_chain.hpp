// smr_chain.hpp -- part of the HIP kernels of libsmr_hip (included by smr_kernels.hpp).
#pragma once

namespace smr {

// ------------------------------------------------------------------------------------------------
// k_chain helpers (block = one wave)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  for (int d = 32; d > 0; d >>= 1) { unsigned long long o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
  return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t v, uint32_t& total) {
  uint32_t incl = v; const int lane = lane_id();
  for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
  total = __shfl(incl, 63, 64);
  return incl - v;
}

// in-wave bitonic sort of n u64 keys (ascending); buffer must hold npow2 >= n entries
__device__ void wave_sort_u64(unsigned long long* keys, uint32_t n) {
  const int lane = lane_id();
  uint32_t np = 1; while (np < n) np <<= 1;
  for (uint32_t i = n + lane; i < np; i += 64) keys[i] = ~0ull;
  __syncthreads();
  for (uint32_t k = 2; k <= np; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = lane; i < np; i += 64) {
        uint32_t p = i ^ j;
        if (p > i) {
          unsigned long long a = keys[i], b = keys[p];
          bool up = (i & k) == 0;
          if ((a > b) == up) { keys[i] = b; keys[p] = a; }
        }
      }
      __syncthreads();
    }
  }
}

struct SwRes { int score, end_ref, end_read; };

// Smith-Waterman score + end cell as an anti-diagonal systolic array: lane = read row within a strip of 64 rows,
// step t computes column t-lane.  Same H as the reference's striped SSE2 kernels (ssw.c:150-575) under the
// affine model H = max(0, diag+s, E, F), first gap base costs gap_open, further bases gap_ext; end cell =
// first column (in processing order) where the maximum is first reached, smallest row in that column
// (ssw.c:305-336).  dir = +1 forward, -1 reverse (the reverse pass ssw.c:900-918 runs the same recurrence on the
// reversed prefixes; since its maximum equals the forward score, stopping at `terminate` selects the same cell).
// rdq: read in 0..4 alphabet (LDS), rfq: reference window (LDS).  bound: 2*n ints of LDS.
__device__ SwRes sw_wave(const uint8_t* rdq, int m, int rd0, int rdstep, const uint8_t* rfq, int n, int rf0, int rfstep,
                         int* bound, int match, int mismatch, int scoreN, int go, int ge) {
  const int lane = lane_id();
  int bestH = 0, bestcol = 0x7fffffff, bestrow = 0x7fffffff;
  const int nstrips = (m + 63) >> 6;
  for (int s = 0; s < nstrips; s++) {
    const int row = s * 64 + lane;
    const bool vrow = row < m;
    const int rnt = vrow ? rdq[rd0 + rdstep * row] : 4;
    int Hcur = 0, Fcur = 0, Ecur = 0, Hdiag = 0;
    const int steps = n + 63;
    for (int t = 0; t < steps; t++) {
      int upH = __shfl_up(Hcur, 1, 64);
      int upF = __shfl_up(Fcur, 1, 64);
      if (lane == 0) {
        if (s == 0 || t >= n) { upH = 0; upF = 0; }
        else { upH = bound[2 * t]; upF = bound[2 * t + 1]; }
      }
      const int col = t - lane;
      const bool act = vrow && col >= 0 && col < n;
      int Hn = Hcur, Fn = Fcur, En = Ecur;
      if (act) {
        const int fnt = rfq[rf0 + rfstep * col];
        const int sc = (fnt == 4 || rnt == 4) ? scoreN : (fnt == rnt ? match : mismatch);
        const int Hleft = (col == 0) ? 0 : Hcur;
        const int Eleft = (col == 0) ? 0 : Ecur;
        int e = max(Eleft - ge, Hleft - go);
        int f = max(upF - ge, upH - go);
        int h = Hdiag + sc;
        h = max(h, e); h = max(h, f); h = max(h, 0);
        e = max(e, 0); f = max(f, 0);
        Hn = h; Fn = f; En = e;
        if (h > bestH || (h == bestH && (col < bestcol || (col == bestcol && row < bestrow)))) {
          if (h > 0) { bestH = h; bestcol = col; bestrow = row; }
        }
      }
      // diag for the next step is the up value of this step (H(row-1, col))
      Hdiag = (col >= 0) ? upH : 0;
      Hcur = Hn; Fcur = Fn; Ecur = En;
      if (lane == 63 && act && s + 1 < nstrips) { bound[2 * col] = Hn; bound[2 * col + 1] = Fn; }
    }
    __syncthreads();
  }
  // lexicographic reduce: max H, then min col, then min row
  unsigned long long key = bestH > 0 ? (((unsigned long long)bestH << 42) | ((unsigned long long)(0x1FFFFF - bestcol) << 21) |
                                        (unsigned long long)(0x1FFFFF - bestrow)) : 0ull;
  key = wave_max_u64(key);
  SwRes r;
  if (key == 0) { r.score = 0; r.end_ref = -1; r.end_read = m - 1; return r; }
  r.score = (int)(key >> 42);
  r.end_ref = 0x1FFFFF - (int)((key >> 21) & 0x1FFFFF);
  r.end_read = 0x1FFFFF - (int)(key & 0x1FFFFF);
  return r;
}

// per-block (= per persistent wave slot) scratch in global memory
struct ChainScratch {
  uint32_t* cnt;                    // n_refs counters, all zero between reads
  unsigned long long* keys;         // candidate keys, capacity keys_cap (>= pow2(n_refs))
  unsigned long long* pairs;        // hits on one reference, capacity pairs_cap (pow2)
  uint32_t* lis;                    // 2 * pairs_cap (b and p arrays of find_lis)
  uint2* hits;                      // gathered (id,win) of the read, capacity hits_cap
  uint32_t keys_cap, pairs_cap, hits_cap;
};

#define CH_KEYS_LDS 512
#define CH_PAIRS_LDS 256
#define CH_HITS_LDS 256

// find_lis (alignment.cpp:58-98) over a[0..n): keys = ref_pos<<32 | read_pos ; compares read_pos (.second).
// executed redundantly by every lane (uniform control flow); b,p hold indices.
__device__ uint32_t find_lis_dev(const unsigned long long* a, uint32_t n, uint32_t* b, uint32_t* p) {
  if (n == 0) return 0;
  uint32_t nb = 0;
  for (uint32_t i = 0; i < n; i++) p[i] = 0;
  b[nb++] = 0;
  for (uint32_t i = 1; i < n; i++) {
    uint32_t ai = (uint32_t)a[i];
    if ((uint32_t)a[b[nb - 1]] < ai) { p[i] = b[nb - 1]; b[nb++] = i; continue; }
    uint32_t u = 0, v = nb - 1;
    while (u < v) { uint32_t c = (u + v) / 2; if ((uint32_t)a[b[c]] < ai) u = c + 1; else v = c; }
    if (ai < (uint32_t)a[b[u]]) { if (u > 0) p[i] = b[u - 1]; b[u] = i; }
  }
  for (uint32_t u = nb, v = b[nb - 1]; u--; v = p[v]) b[u] = v;
  return nb;
}

// One block (64 threads = one wave) per read, persistent.  Dynamic LDS layout (bytes), ML = max_len rounded:
//   rdq[ML] | rfq[ML+2*edges_max+64] | bound[2*(ML+...)] ints | keys[CH_KEYS_LDS] u64 | pairs[CH_PAIRS_LDS] u64 |
//   lis[2*CH_PAIRS_LDS] u32 | hits[CH_HITS_LDS] uint2
__global__ void __launch_bounds__(64) k_chain(DReads rd, DIndex ix, DParams P, int pass, int is_last_strand,
                                              RState* __restrict__ work, AlignRec* __restrict__ work_aln, RWork* __restrict__ rw,
                                              const uint32_t* __restrict__ pool, unsigned long long* __restrict__ ctr,
                                              uint32_t* g_cnt, unsigned long long* g_keys, unsigned long long* g_pairs, uint32_t* g_lis,
                                              uint2* g_hits, uint32_t keys_cap, uint32_t pairs_cap, uint32_t hits_cap,
                                              uint32_t lds_ml, uint32_t lds_rf) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  __shared__ uint32_t s_next;
  __shared__ uint32_t s_ncand;
  const int lane = lane_id();
  uint8_t* rdq = lds_raw;
  uint8_t* rfq = rdq + lds_ml;
  int* bound = (int*)(rfq + lds_rf);
  unsigned long long* l_keys = (unsigned long long*)(bound + 2 * lds_rf);
  unsigned long long* l_pairs = l_keys + CH_KEYS_LDS;
  uint32_t* l_lis = (uint32_t*)(l_pairs + CH_PAIRS_LDS);
  uint2* l_hits = (uint2*)(l_lis + 2 * CH_PAIRS_LDS);

  uint32_t* cnt = g_cnt + (size_t)blockIdx.x * ix.n_refs;
  unsigned long long* gk = g_keys + (size_t)blockIdx.x * keys_cap;
  unsigned long long* gp = g_pairs + (size_t)blockIdx.x * pairs_cap;
  uint32_t* gl = g_lis + (size_t)blockIdx.x * 2 * pairs_cap;
  uint2* gh = g_hits + (size_t)blockIdx.x * hits_cap;

  unsigned long long n_fwd = 0, n_rev = 0, n_cells = 0;   // flushed once per block (lane 0)
  uint32_t chunk_next = 0, chunk_end = 0;                 // reads are claimed 16 at a time (one atomic per chunk)
  for (;;) {
    if (chunk_next == chunk_end) {
      __syncthreads();
      if (lane == 0) s_next = (uint32_t)atomicAdd(&ctr[C_WORK_NEXT], 16ull);
      __syncthreads();
      chunk_next = s_next; chunk_end = min(chunk_next + 16u, rd.n);
      if (chunk_next >= rd.n) break;
    }
    __syncthreads();
    const uint32_t r = chunk_next++;
    RWork w = rw[r];
    if (!(w.strand_active && w.search && w.pass_n == (uint32_t)pass)) continue;
    RState st = work[r];
    const uint32_t len = rd.len[r];
    const uint32_t* rec = rd.words + rd.rec_off[r];
    int search = 1;
    const uint32_t max_SW_score = len * (uint32_t)P.match;

    if (st.hit_seeds >= (uint32_t)P.num_seeds && w.hit_total > 0) {
      // ---------------- compute_lis_alignment (alignment.cpp:100-509) ----------------
      // gather this strand's hits (all passes so far) into a flat array
      const uint32_t nh = w.hit_total;
      uint2* hits = nh <= CH_HITS_LDS ? l_hits : gh;
      bool cap_err = nh > hits_cap && nh > CH_HITS_LDS;
      if (cap_err) { if (lane == 0) atomicAdd(&ctr[C_ERR_PAIRS], 1ull); }
      else {
        uint32_t o = 0;
        for (uint32_t pp = 0; pp < 3; pp++) {              // one contiguous block per pass run so far on this strand
          const uint32_t c = w.blk_cnt[pp], bo = w.blk_off[pp];
          for (uint32_t q = lane; q < c; q += 64) hits[o + q] = make_uint2(pool[bo + 2 * q], pool[bo + 2 * q + 1]);
          o += c;
        }
        __syncthreads();
        // 1. per-reference histogram of seed hits (:117-130) with device atomics on this slot's private counters
        if (lane == 0) s_ncand = 0;
        __syncthreads();
        for (uint32_t hb = 0; hb < nh; hb += 64) {
          uint32_t h = hb + lane;
          uint32_t lo = 0, hi = 0;
          if (h < nh) { uint32_t id = hits[h].x; lo = ix.pos_off[id]; hi = ix.pos_off[id + 1]; }
          // short lists: lane-serial; long lists: the whole wave walks them together
          bool lng = (hi - lo) > 32;
          if (!lng) {
            for (uint32_t k = lo; k < hi; k++) {
              uint32_t seq = ix.pos_arr[k].y;
              uint32_t old = atomicAdd(&cnt[seq], 1u);
              if (old + 1 == (uint32_t)P.num_seeds) { uint32_t sl = atomicAdd(&s_ncand, 1u); if (sl < keys_cap) gk[sl] = seq; }
            }
          }
          unsigned long long lm = __ballot(lng);
          while (lm) {
            int src = __ffsll((long long)lm) - 1; lm &= lm - 1;
            uint32_t llo = __shfl(lo, src, 64), lhi = __shfl(hi, src, 64);
            for (uint32_t k = llo + lane; k < lhi; k += 64) {
              uint32_t seq = ix.pos_arr[k].y;
              uint32_t old = atomicAdd(&cnt[seq], 1u);
              if (old + 1 == (uint32_t)P.num_seeds) { uint32_t sl = atomicAdd(&s_ncand, 1u); if (sl < keys_cap) gk[sl] = seq; }
            }
          }
        }
        __threadfence_block();
        __syncthreads();
        uint32_t ncand = s_ncand;
        if (ncand > keys_cap) { if (lane == 0) atomicAdd(&ctr[C_ERR_PAIRS], 1ull); ncand = 0; cap_err = true; }
        unsigned long long* keys = ncand <= CH_KEYS_LDS ? l_keys : gk;
        // key = (~count, ref): ascending order == count desc, ref asc (:134-148)
        for (uint32_t c = lane; c < ncand; c += 64) {
          uint32_t ref = (uint32_t)gk[c];
          uint32_t count = __hip_atomic_load(&cnt[ref], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          keys[c] = ((unsigned long long)(0xFFFFFFFFu - count) << 32) | ref;
        }
        __syncthreads();
        // clear the counters (walk again)
        for (uint32_t hb = 0; hb < nh; hb += 64) {
          uint32_t h = hb + lane;
          uint32_t lo = 0, hi = 0;
          if (h < nh) { uint32_t id = hits[h].x; lo = ix.pos_off[id]; hi = ix.pos_off[id + 1]; }
          bool lng = (hi - lo) > 32;
          if (!lng) for (uint32_t k = lo; k < hi; k++) cnt[ix.pos_arr[k].y] = 0;
          unsigned long long lm = __ballot(lng);
          while (lm) {
            int src = __ffsll((long long)lm) - 1; lm &= lm - 1;
            uint32_t llo = __shfl(lo, src, 64), lhi = __shfl(hi, src, 64);
            for (uint32_t k = llo + lane; k < lhi; k += 64) cnt[ix.pos_arr[k].y] = 0;
          }
        }
        __syncthreads();
        if (ncand > 1) wave_sort_u64(keys, ncand);
        __syncthreads();

        // 2. candidate loop (:150-508)
        int is_aligned = 0;
        int is_search_candidates = 1;
        for (uint32_t k = 0; k < ncand && is_search_candidates && !cap_err; k++) {
          const unsigned long long ck = keys[k];
          const uint32_t max_ref = (uint32_t)ck;
          const uint32_t max_occur = 0xFFFFFFFFu - (uint32_t)(ck >> 32);
          if (max_occur < (uint32_t)P.num_seeds) break;
          if (is_aligned && P.min_lis > 0 && k > 0 && max_occur < (0xFFFFFFFFu - (uint32_t)(keys[k - 1] >> 32))) {   // :165-169
            --w.best;
            if (w.best < 1) break;
          }
          // 3. hits on this reference (:181-201): each lane binary-searches one hit's (seq-sorted) position list
          uint32_t np = 0;
          for (int phase = 0; phase < 2; phase++) {
            // phase 0 counts, phase 1 writes at deterministic offsets
            uint32_t run = 0;
            unsigned long long* pairs = np <= CH_PAIRS_LDS ? l_pairs : gp;
            for (uint32_t hb = 0; hb < nh; hb += 64) {
              uint32_t h = hb + lane;
              uint32_t first = 0, cntm = 0, win = 0;
              if (h < nh) {
                uint32_t id = hits[h].x; win = hits[h].y;
                uint32_t lo = ix.pos_off[id], hi = ix.pos_off[id + 1];
                uint32_t a = lo, b = hi;
                while (a < b) { uint32_t mid = (a + b) >> 1; if (ix.pos_arr[mid].y < max_ref) a = mid + 1; else b = mid; }
                first = a;
                uint32_t c2 = a, d2 = hi;
                while (c2 < d2) { uint32_t mid = (c2 + d2) >> 1; if (ix.pos_arr[mid].y <= max_ref) c2 = mid + 1; else d2 = mid; }
                cntm = c2 - a;
              }
              uint32_t tot; uint32_t ex = wave_excl_scan_u32(cntm, tot);
              if (phase == 1) for (uint32_t q = 0; q < cntm; q++) pairs[run + ex + q] = ((unsigned long long)ix.pos_arr[first + q].x << 32) | win;
              run += tot;
            }
            if (phase == 0) { np = run; if (np > pairs_cap && np > CH_PAIRS_LDS) { cap_err = true; break; } }
          }
          if (cap_err) { if (lane == 0) atomicAdd(&ctr[C_ERR_PAIRS], 1ull); break; }
          unsigned long long* pairs = np <= CH_PAIRS_LDS ? l_pairs : gp;
          uint32_t* lisb = np <= CH_PAIRS_LDS ? l_lis : gl;
          uint32_t* lisp = lisb + (np <= CH_PAIRS_LDS ? CH_PAIRS_LDS : pairs_cap);
          __syncthreads();
          if (np > 1) wave_sort_u64(pairs, np);
          __syncthreads();
          // 4. sliding window of read length along the reference (:203-506)
          uint32_t it = 0, ms_lo = 0, ms_hi = 0;
          uint32_t begin_ref = (uint32_t)(pairs[0] >> 32), begin_read = (uint32_t)pairs[0];
          const uint64_t reflen = ix.ref_off[max_ref + 1] - ix.ref_off[max_ref];
          while (it != np && is_search_candidates) {
            const uint64_t end_ref_max = (uint64_t)begin_ref + len - begin_read - P.lnwin + 1;
            int push = 0;
            while (it != np && (uint64_t)(uint32_t)(pairs[it] >> 32) <= end_ref_max) { ms_hi = ++it; push = 1; }
            int skip_to_pop = 0;
            if (!push && is_aligned) skip_to_pop = 1;        // heuristic 1 (:243-246)
            else is_aligned = 0;
            if (!skip_to_pop && (ms_hi - ms_lo) >= (uint32_t)P.num_seeds) {
              uint32_t nl = find_lis_dev(pairs + ms_lo, ms_hi - ms_lo, lisb, lisp);
              if (nl >= (uint32_t)P.min_lis) {
                const uint32_t lcs_ref_start = (uint32_t)(pairs[ms_lo + lisb[0]] >> 32);
                const uint32_t lcs_que_start = (uint32_t)pairs[ms_lo + lisb[0]];
                uint64_t head = 0, tail = 0, align_ref_start = 0, align_que_start = 0, align_length = 0;
                const uint64_t rlen = len;
                uint32_t edges;
                if (P.is_as_percent) edges = (uint32_t)((P.edges / 100.0) * (double)rlen);
                else edges = (uint32_t)P.edges;
                if (lcs_ref_start < lcs_que_start) {                         // :287-325
                  align_ref_start = 0; align_que_start = lcs_que_start - lcs_ref_start; head = 0;
                  if (reflen < rlen) {
                    tail = 0;
                    if (align_que_start > (rlen - reflen)) align_length = reflen - (align_que_start - (rlen - reflen));
                    else align_length = reflen;
                  } else {
                    tail = reflen - align_ref_start - rlen;
                    if (tail > (uint64_t)(uint32_t)(edges - 1)) tail = edges;
                    align_length = rlen + head + tail - align_que_start;
                  }
                } else {                                                     // :326-357
                  align_ref_start = lcs_ref_start - lcs_que_start; align_que_start = 0;
                  if (align_ref_start > (uint64_t)(uint32_t)(edges - 1)) head = edges;
                  if (align_ref_start + rlen > reflen) { tail = 0; align_length = reflen - align_ref_start - head; }
                  else {
                    tail = reflen - align_ref_start - rlen;
                    if (tail > (uint64_t)(uint32_t)(edges - 1)) tail = edges;
                    align_length = rlen + head + tail;
                  }
                }
                // read.flip34() to the 0..4 alphabet before SSW (:360-361)
                // (is03/is04 only toggle when the read has ambiguous letters; aval tracks the stored value)
                if (w.has_amb && !w.is04) { w.is04 = 1; w.aval = 4; }
                const int m = (int)(align_length - head - tail);
                const int nref = (int)align_length;
                const uint64_t rf_start = ix.ref_off[max_ref] + align_ref_start - head;
                SwRes fw; fw.score = 0; fw.end_ref = -1; fw.end_read = m - 1;
                bool sw_ok = (m > 0 && nref > 0 && (uint32_t)m <= lds_ml && (uint32_t)nref <= lds_rf);
                if (!sw_ok && (m > 0 && nref > 0)) { if (lane == 0) atomicAdd(&ctr[C_ERR_PAIRS], 1ull); cap_err = true; }
                if (sw_ok) {
                  for (int q = lane; q < m; q += 64) rdq[q] = (uint8_t)read_nt(rec, len, (uint32_t)align_que_start + q, w.reversed, w.aval);
                  for (int q = lane; q < nref; q += 64) rfq[q] = ix.ref_seq[rf_start + q];
                  __syncthreads();
                  fw = sw_wave(rdq, m, 0, 1, rfq, nref, 0, 1, bound, P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext);
                  n_fwd++; n_cells += (unsigned long long)m * nref;
                }
                int score1 = fw.score > 65535 ? 65535 : fw.score;
                int ref_begin1 = -1, read_begin1 = -1;
                const int ref_end1 = fw.end_ref, read_end1 = fw.end_read;
                if (sw_ok && (uint32_t)score1 >= (P.minimal_score & 0xFFFFu)) {   // ssw_align: flag==2 && score1 < filters -> no begin
                  // reverse pass (ssw.c:900-918) on the prefixes ending at (read_end1, ref_end1)
                  SwRes bw = sw_wave(rdq, read_end1 + 1, read_end1, -1, rfq, ref_end1 + 1, ref_end1, -1, bound,
                                     P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext);
                  ref_begin1 = ref_end1 - bw.end_ref;
                  read_begin1 = read_end1 - bw.end_read;
                  n_rev++; n_cells += (unsigned long long)(read_end1 + 1) * (ref_end1 + 1);
                }
                is_aligned = (sw_ok && (uint32_t)score1 > P.minimal_score);     // strict (:388)
                if (is_aligned) {
                  if ((uint32_t)score1 == max_SW_score) ++st.max_SW_count;
                  AlignRec al;
                  al.ref_begin1 = ref_begin1 + (int32_t)(align_ref_start - head);
                  al.ref_end1 = ref_end1 + (int32_t)(align_ref_start - head);
                  al.read_begin1 = read_begin1 + (int32_t)align_que_start;
                  al.read_end1 = read_end1 + (int32_t)align_que_start;
                  al.readlen = len; al.ref_num = max_ref;
                  al.index_num = (uint16_t)P.index_num; al.part = (uint16_t)P.part;
                  al.strand = (uint8_t)!w.reversed; al.score1 = (uint16_t)score1;
                  al.has_cigar = 0; al.cigar_off = 0; al.cigar_len = 0;
                  AlignRec* slots = work_aln + (size_t)r * P.slots;
                  if (!st.is_hit) {                                              // :411-416
                    st.is_hit = 1;
                    if (lane == 0) { atomicAdd(&ctr[C_NUM_ALIGNED], 1ull); atomicAdd(&ctr[C_PER_DB + P.index_num], 1ull); }
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
                      // :454-457 decrement/increment of reads_matched_per_db cancel (both use the NEW alignment's index)
                    }
                  }
                  __syncthreads();
                  if (P.num_alignments > 0) {                                    // :462-469
                    if (P.is_best) { if (P.num_alignments == st.max_SW_count) is_search_candidates = 0; }
                    else if (P.num_alignments == st.n_align) is_search_candidates = 0;
                  }
                  search = 0;
                }
              }
            }
            // pop (:486-506)
            if (ms_hi > ms_lo) ms_lo++;
            if (ms_hi == ms_lo) {
              if (it != np) { begin_ref = (uint32_t)(pairs[it] >> 32); begin_read = (uint32_t)pairs[it]; }
              else break;
            } else { begin_ref = (uint32_t)(pairs[ms_lo] >> 32); begin_read = (uint32_t)pairs[ms_lo]; }
          }
          __syncthreads();
        }
      }
    }
    // ---------------- pass control (paralleltraversal.cpp:253-277) ----------------
    uint32_t pass_n = w.pass_n;
    if (search) {
      if (pass_n == 2) search = 0;
      else {
        while (pass_n < 2 && P.skip[pass_n] == P.skip[pass_n + 1]) ++pass_n;
        if (++pass_n > 2) search = 0;
        else w.win_shift = P.skip[pass_n];
      }
    }
    w.pass_n = (uint8_t)pass_n; w.search = (uint8_t)search;
    if (!search) {
      // end of traverse() (:279-297)
      st.lastIndex = P.index_num; st.lastPart = P.part;
      if (P.num_alignments > 0) {
        if ((P.is_best && P.num_alignments == st.max_SW_count) || (!P.is_best && st.n_align == P.num_alignments)) st.is_done = 1;
      } else if (P.is_last_index_part && is_last_strand && st.n_align > 0) st.is_done = 1;
      w.strand_active = 0;
    }
    if (lane == 0) { work[r] = st; rw[r] = w; }
  }
  if (lane == 0) {
    if (n_fwd) ctr_add(ctr, C_SW_FWD, n_fwd);
    if (n_rev) ctr_add(ctr, C_SW_REV, n_rev);
    if (n_cells) ctr_add(ctr, C_SW_CELLS, n_cells);
  }
}

}  // namespace smr
