// smr_seed.hpp -- part of the HIP kernels of libsmr_hip (included by smr_kernels.hpp).
#pragma once

namespace smr {

// ------------------------------------------------------------------------------------------------
// k_seed
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lev_step(const uint8_t* lev, uint32_t depth, uint32_t partialwin, uint32_t bv_lo_hi_sel_nibble,
                                             uint32_t last_row_nibble, uint32_t state) {
  // depth < partialwin-2: 4-bit vector into t0; else the last row masked to (partialwin-depth+1) bits
  if (depth < partialwin - 2) return lev[bv_lo_hi_sel_nibble * 14 + state];
  uint32_t t = 3 - partialwin + depth;                         // 1,2,3
  uint32_t v = last_row_nibble & ((2u << (partialwin - depth)) - 1u);
  uint32_t base = t == 1 ? LEV_T1 : (t == 2 ? LEV_T2 : LEV_T3);
  return lev[base + v * 14 + state];
}

// characteristic bit-vectors of a 9-mer (bitvector.cpp:57-132): nibble (d, nt), bit k set iff c[d+2-k] == nt
struct BitVec {
  unsigned long long lo, hi;      // nibble index e = d*4+nt ; e < 16 -> lo, else hi
  __device__ __forceinline__ uint32_t get(uint32_t d, uint32_t nt) const {
    uint32_t e = d * 4 + nt;
    return (uint32_t)((e < 16 ? (lo >> (e * 4)) : (hi >> ((e - 16) * 4))) & 15ull);
  }
};
__device__ __forceinline__ BitVec make_bitvec(uint32_t chars /*2 bits per char, char i at bits 2i*/, uint32_t partialwin) {
  BitVec b; b.lo = 0; b.hi = 0;
  uint32_t rows = partialwin - 2;
  for (uint32_t d = 0; d < rows; d++) {
    for (uint32_t k = 0; k < 4; k++) {
      int ci = (int)d + 2 - (int)k;
      if (ci < 0) continue;
      uint32_t nt = (chars >> (2 * ci)) & 3u;
      uint32_t e = d * 4 + nt;
      if (e < 16) b.lo |= (unsigned long long)(1u << k) << (e * 4);
      else b.hi |= (unsigned long long)(1u << k) << ((e - 16) * 4);
    }
  }
  return b;
}

struct SeedCounters { uint32_t lookup, node, entry; };

// ------------------------------------------------------------------------------------------------
// k_seed: window scan + burst-trie descent, one wave per block.
//
// The wave is split into groups of `gw` lanes (gw = pow2 >= windows per read in this pass); each group owns one
// read, each lane one window.  A lane walks its two mini-tries (forward, then reverse unless the forward search
// ended with a 0-error match) exactly in the reference's DFS order (A<C<G<T, traverse_bursttrie.cpp:117), but the
// walk is cut into ROUNDS: in a round every lane advances over trie NODES only, until it stands in front of its
// next bucket; then the whole wave scans the entries of all 64 pending buckets together, one lane per ENTRY
// (prefix sum over the bucket sizes, owner found by binary search), so the dominant work -- the LEV(1) automaton
// over bucket entries -- runs with full lanes and contiguous 8-byte loads instead of one divergent lane per window.
// Accepted entries ("candidates", rare) are handed back to the owning lane in entry order, which applies the
// reference's sequential rules to its lane-local hit list in LDS:
//   entry accepted at t_a = first step with depth_b >= pw-2 and state >= 8   (traverse_bursttrie.cpp:229-235)
//   UNCOND  state 9 at depth_b == pw-1 in the accepting step itself  -> 0-error hit: list = {id}, search over (:256-262)
//   COND    accepted at pw-2 and state 9 one step later: the reference reaches that step only if the id was NOT
//           already in the list when it was accepted (otherwise the duplicate check `break`s first, :265-277)
//   PLAIN   1-error hit: appended unless the id is already present
// LDS per wave: hit lists hl[hcap][64], node-offset stacks stk[12][64], pref/bdesc/bmeta[64], bit-vectors bvw[4][64].
// ------------------------------------------------------------------------------------------------
#define SEED_STK 12
#define SEED_LDS_WORDS(hcap) (64u * (hcap) + SEED_STK * 64u + 3u * 64u + 4u * 64u)

enum { PH_F_INIT = 0, PH_F = 1, PH_R_INIT = 2, PH_R = 3, PH_DONE = 4 };
enum { CK_PLAIN = 0, CK_UNCOND = 1, CK_COND = 2 };

__global__ void __launch_bounds__(64) k_seed(DReads rd, DIndex ix, DParams P, int pass, uint32_t gw, uint32_t hcap,
                                             RState* __restrict__ work, RWork* __restrict__ rw, uint32_t* __restrict__ pool,
                                             uint32_t pool_words, unsigned long long* __restrict__ ctr) {
  extern __shared__ uint32_t lds_dyn[];
  uint32_t* hl = lds_dyn;
  uint32_t* stk = hl + 64 * hcap;
  uint32_t* pref = stk + SEED_STK * 64;
  uint32_t* bdesc = pref + 64;
  uint32_t* bmeta = bdesc + 64;
  uint32_t* bvw = bmeta + 64;
  __shared__ uint8_t s_lev[LEV_SIZE];
  for (uint32_t i = threadIdx.x; i < LEV_SIZE; i += blockDim.x) s_lev[i] = c_lev[i];
  __syncthreads();
  const int lane = lane_id();
  const uint32_t gpw = 64 / gw;                                   // groups (reads) per wave
  const uint32_t g = lane / gw, wl = lane % gw;
  const uint32_t r = blockIdx.x * gpw + g;
  const uint32_t pw = P.partialwin, L = P.lnwin;
  const uint32_t last_row = pw - 3;
  const bool full = P.is_full_search != 0;

  bool active = false;
  RWork w;
  uint32_t len = 0;
  const uint32_t* rec = nullptr;
  if (r < rd.n) {
    w = rw[r];
    active = (w.strand_active && w.search && w.pass_n == (uint32_t)pass);
    len = rd.len[r];
    rec = rd.words + rd.rec_off[r];
  }
  uint32_t aval = 0, reversed = 0;
  if (active) {
    // traverse(): `if (read.is04) read.flip34()` before every window (paralleltraversal.cpp:126)
    aval = w.is04 ? 0 : w.aval;
    reversed = w.reversed;
  }
  const uint32_t stride = P.skip[pass];
  const uint32_t numwin = active ? (len - L + stride) / stride : 0;       // :118-120

  SeedCounters sc; sc.lookup = 0; sc.node = 0; sc.entry = 0;
  uint32_t n_win_searched = 0, grp_hits_total = 0, grp_seeds = 0;
  bool overflow = false;
  uint32_t seg_head = active ? w.hit_head : NONE;

  for (uint32_t wbase = 0; __any(wbase < numwin); wbase += gw) {
    uint32_t k = wbase + wl;
    uint32_t nh = 0;
    bool mine = active && k < numwin;
    uint32_t win_pos = k * stride;
    if (mine) {                                        // read_pos_searched (paralleltraversal.cpp:128-131)
      for (int q = 0; q < pass; q++) if (win_pos % P.skip[q] == 0) mine = false;
    }
    uint32_t keyf = 0, keyr = 0, fchars = 0, rchars = 0;
    if (mine) {
      n_win_searched++;
      // window content, char i at bits 2i
      unsigned long long wchars = 0;
      for (uint32_t i = 0; i < L; i++) wchars |= (unsigned long long)read_nt(rec, len, win_pos + i, reversed, aval) << (2 * i);
      // forward half: key = first partialwin chars (MSB first, Read::hashKmer read.cpp:601-611), bit-vectors from
      // chars [pw .. 2pw) (init_win_f); reverse half: key = chars [pw .. 2pw), bit-vectors from chars pw-1 .. 0 (init_win_r)
      for (uint32_t i = 0; i < pw; i++) {
        keyf = (keyf << 2) | (uint32_t)((wchars >> (2 * i)) & 3);
        keyr = (keyr << 2) | (uint32_t)((wchars >> (2 * (pw + i))) & 3);
        fchars |= (uint32_t)((wchars >> (2 * (pw + i))) & 3) << (2 * i);
        rchars |= (uint32_t)((wchars >> (2 * (pw - 1 - i))) & 3) << (2 * i);
      }
    }
    // ---- lane-local DFS state ----
    int phase = mine ? PH_F_INIT : PH_DONE;
    bool zero = false;
    uint32_t root = 0;
    int sp = -1;
    unsigned long long cur_bits = 0, piv_bits = 0;        // 3-bit element cursors / 4-bit pivot states per level
    uint4 cur = make_uint4(0, 0, 0, 0);                   // the 4 elements of the node on top of the stack
    BitVec bv; bv.lo = 0; bv.hi = 0;

    for (;;) {
      // ---------- node walk: advance to this lane's next bucket ----------
      bool has = false;
      uint32_t b_off = 0, b_nent = 0, b_depth = 0, b_lev = 0;
      while (phase != PH_DONE && !has) {
        if (phase == PH_F_INIT || phase == PH_R_INIT) {
          const bool fwd = phase == PH_F_INIT;
          const Lookup lk = ix.lookup[fwd ? keyf : keyr]; sc.lookup++;
          const uint32_t rt = fwd ? lk.rootF : lk.rootR;
          if (lk.count > P.minoccur && rt != NONE) {
            root = rt; bv = make_bitvec(fwd ? fchars : rchars, pw);
            sp = 0; cur_bits = 0; piv_bits = 0; stk[lane] = 0;
            cur = *reinterpret_cast<const uint4*>(ix.trie + root); sc.node++;
            bvw[lane] = (uint32_t)bv.lo; bvw[64 + lane] = (uint32_t)(bv.lo >> 32);
            bvw[128 + lane] = (uint32_t)bv.hi; bvw[192 + lane] = (uint32_t)(bv.hi >> 32);
            phase = fwd ? PH_F : PH_R;
          } else phase = fwd ? PH_R_INIT : PH_DONE;
          continue;
        }
        if (sp < 0) { phase = (phase == PH_F) ? PH_R_INIT : PH_DONE; continue; }
        const uint32_t ne = (uint32_t)(cur_bits >> (3 * sp)) & 7u;
        if (ne == 4) {
          sp--;
          if (sp >= 0) cur = *reinterpret_cast<const uint4*>(ix.trie + root + stk[sp * 64 + lane]);
          continue;
        }
        cur_bits += 1ull << (3 * sp);
        const uint32_t e = ne == 0 ? cur.x : (ne == 1 ? cur.y : (ne == 2 ? cur.z : cur.w));
        const uint32_t flag = e >> ELEM_FLAG_SHIFT;
        if (flag == 0) continue;
        const uint32_t depth = (uint32_t)sp;
        const uint32_t piv = (uint32_t)(piv_bits >> (4 * sp)) & 15u;
        const uint32_t lev_t = lev_step(s_lev, depth, pw, depth < pw - 2 ? bv.get(depth, ne) : 0, bv.get(last_row, ne), piv);
        if (lev_t == 14) continue;
        if (flag == 1) {
          sp++;
          stk[sp * 64 + lane] = e & ELEM_OFF_MASK;
          cur_bits &= ~(7ull << (3 * sp));
          piv_bits = (piv_bits & ~(15ull << (4 * sp))) | ((unsigned long long)lev_t << (4 * sp));
          cur = *reinterpret_cast<const uint4*>(ix.trie + root + (e & ELEM_OFF_MASK)); sc.node++;
          continue;
        }
        has = true; b_off = root + (e & ELEM_OFF_MASK); b_nent = (e >> ELEM_NENT_SHIFT) & 0xFFu; b_depth = depth; b_lev = lev_t;
      }
      if (!__any(has)) break;
      // ---------- entry scan: one lane per entry of the 64 pending buckets ----------
      const uint32_t my_n = has ? b_nent : 0;
      sc.entry += my_n;
      uint32_t incl = my_n;
      for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
      const uint32_t T = __shfl(incl, 63, 64);
      pref[lane] = incl - my_n; bdesc[lane] = b_off; bmeta[lane] = b_depth | (b_lev << 4);
      __syncthreads();
      for (uint32_t base = 0; base < T; base += 64) {
        const uint32_t e = base + lane;
        const bool v = e < T;
        uint32_t owner = 0;
        if (v) {                                           // largest o with pref[o] <= e
          for (uint32_t step = 32; step > 0; step >>= 1) { const uint32_t t = owner + step; if (t < 64 && pref[t] <= e) owner = t; }
        }
        const uint32_t q = e - pref[owner];
        const uint32_t meta = bmeta[owner];
        uint32_t depth_b = meta & 15u, lv = meta >> 4;
        uint32_t str = 0, id = 0;
        if (v) { const uint2 en = *reinterpret_cast<const uint2*>(ix.trie + bdesc[owner] + 2 * q); str = en.x; id = en.y; }
        BitVec obv;
        obv.lo = (unsigned long long)bvw[owner] | ((unsigned long long)bvw[64 + owner] << 32);
        obv.hi = (unsigned long long)bvw[128 + owner] | ((unsigned long long)bvw[192 + owner] << 32);
        bool alive = v, acc = false;
        uint32_t kind = CK_PLAIN;
        for (uint32_t j = 0; j < pw; j++) {
          if (!__any(alive)) break;
          if (alive) {
            const uint32_t nt = str & 3u; str >>= 2; depth_b++;
            lv = lev_step(s_lev, depth_b, pw, depth_b < pw - 2 ? obv.get(depth_b, nt) : 0, obv.get(last_row, nt), lv);
            if (lv == 14) alive = false;
            else {
              if (depth_b >= pw - 2) {
                const bool z = (depth_b == pw - 1 && lv == 9 && !full);
                if (!acc) { if (lv >= 8) { acc = true; if (z) kind = CK_UNCOND; } }
                else { if (z) kind = CK_COND; alive = false; }
              }
              if (depth_b >= pw) alive = false;
            }
          }
        }
        // hand the candidates back to their owners, in entry order
        unsigned long long cm = __ballot(acc);
        while (cm) {
          const int c = __ffsll((long long)cm) - 1; cm &= cm - 1;
          const uint32_t o = __shfl(owner, c, 64), idc = __shfl(id, c, 64), kc = __shfl(kind, c, 64), qc = __shfl(q, c, 64);
          if ((uint32_t)lane == o && !zero) {
            bool present = false;
            for (uint32_t f = 0; f < nh; f++) if (hl[f * 64 + lane] == idc) { present = true; break; }
            if (kc == CK_UNCOND || (kc == CK_COND && !present)) {
              hl[lane] = idc; nh = 1; zero = true;
              sc.entry -= b_nent - (qc + 1);               // the reference stops scanning at the 0-error entry
            } else if (!present) {
              if (nh < hcap) { hl[nh * 64 + lane] = idc; nh++; } else overflow = true;
            }
          }
        }
      }
      if (zero) phase = PH_DONE;                           // accept_zero_kmer: no reverse search (paralleltraversal.cpp:188)
      __syncthreads();
    }
    // segmented (width gw) inclusive scan of nh
    uint32_t incl = nh;
    for (uint32_t d = 1; d < gw; d <<= 1) { uint32_t t = __shfl_up(incl, d, gw); if (wl >= d) incl += t; }
    uint32_t total = __shfl(incl, gw - 1, gw);
    uint32_t seeds = __popcll(__ballot(nh > 0) & (gw == 64 ? ~0ull : (((1ull << gw) - 1) << (g * gw))));
    uint32_t base = 0;
    if (wl == 0 && total > 0) {
      unsigned long long old = atomicAdd(&ctr[C_POOL_CURSOR], (unsigned long long)(2 + 2 * total));
      if (old + 2 + 2 * (unsigned long long)total > pool_words) { atomicAdd(&ctr[C_ERR_POOL], 1ull); base = NONE; }
      else { base = (uint32_t)old; pool[base] = seg_head; pool[base + 1] = total; seg_head = base; }
    }
    base = __shfl(base, 0, gw);
    if (total > 0 && base != NONE) {
      uint32_t o = base + 2 + 2 * (incl - nh);
      for (uint32_t q = 0; q < nh; q++) { pool[o + 2 * q] = hl[q * 64 + lane]; pool[o + 2 * q + 1] = win_pos; }
    }
    grp_hits_total += total; grp_seeds += seeds;
  }
  if (overflow) atomicAdd(&ctr[C_ERR_HITCAP], 1ull);
  if (active && wl == 0) {
    w.is04 = 0; w.aval = (uint8_t)aval;                     // the flip back to 0..3 is persistent
    w.hit_head = seg_head; w.hit_total += grp_hits_total;
    rw[r] = w;
    work[r].hit_seeds += grp_seeds;                          // ++read.hit_seeds per window with hits (:242-249)
  }
  // work counters
  unsigned long long v[6];
  v[0] = n_win_searched; v[1] = sc.lookup; v[2] = sc.node; v[3] = sc.entry; v[4] = (wl == 0) ? grp_hits_total : 0;
  v[5] = (active && wl == 0) ? ((len + 3) / 4) : 0;
  for (int c = 0; c < 6; c++) {
    unsigned long long x = v[c];
    for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
    if (lane == 0 && x) atomicAdd(&ctr[C_WINDOWS + c], x);
  }
}

}  // namespace smr
