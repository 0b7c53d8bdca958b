// NOT PART OF libsmr_hip (round 4): measured slower than leaving these reads to k_chain (2.48 + 4.35 vs 5.96 ms per 2 M-read launch,
// profiles/r3s17_*); kept as the record of the experiment.  The engine side (k_mark_list / k_quad / k_park_sw launches) is in the history: 2b0b0d8.
// smr_quad.hpp -- part of the HIP kernels of libsmr_hip (included by smr_kernels.hpp after smr_chain.hpp).
//
// Four reads per wave for the small majority of compute_lis_alignment (alignment.cpp:100-509).  Census of the bench workload
// (-DSMR_CHAIN_STATS, profiles/r3s15_*): of the 1.51 M reads k_cand marks per step, 99 % have <= 32 seed hits with <= 64 positions and
// 92 % also <= 8 candidate references of <= 16 pairs each; 57 % meet exactly ONE ssw_align under the assumption that it does not align,
// 17 % none.  k_chain gives each of them a wave that executes scalar control flow in 64 lanes.  Here a read gets 16 lanes:
//
//   k_mark_list   the reads k_cand marked, as a list
//   k_quad        16 lanes per listed read: hits -> positions -> exact candidate set (all in the group's LDS) -> the candidate walk
//                 (sliding window :203-506, LIS :58-98, window geometry :271-357) run under "nothing aligns" exactly like k_chain's look-ahead.
//                 No task: the read's pass ends here (pass control).  One task that fits the four-problem kernel: the task goes to a list.
//                 Anything else (a second task, a set beyond the limits above): the read stays marked for k_chain -- QM_IMMEDIATE when k_chain
//                 would find the same and walk it sequentially, so it does not look ahead a second time.
//   k_park_sw     the listed tasks, four per wave through sw_wave_x4: "no alignment" completes the read as the sequential walk would (one
//                 ssw_align call, nothing recorded, pass control); any other result hands the read to k_chain's sequential walk.
//
// The records are those of the sequential walk: k_quad only decides WHICH reads need it, by the rules k_chain applies to park a read, and every
// doubt is resolved towards k_chain.
//
// MEASURED AND NOT ADOPTED (round 3, MI355X, 2 M-read batches): with this stage k_chain drops from 5.96 to 4.35 ms per launch -- the 73 % of the
// marked reads it is spared cost it only 1.6 ms; what is left are the reads sampled from the DB, whose 40-member families are walked candidate
// by candidate -- while k_mark_list + k_quad + k_park_sw take 2.48 ms: four reads share an instruction stream, but the stream is as long as
// the slowest of the four and its loops run to fixed bounds under predicates, ~2 400 wave-instructions per read, what k_chain spends on such a
// read.  The stage stays in the library behind SMR_QUAD=1 (off by default), with its parity test (test_sixteen_lane_walk_on_and_off_...).
#pragma once

namespace smr {

#define QD_HITS 32u
#define QD_POS 64u
#define QD_CAND 8u
#define QD_PAIRS 16u
enum { QM_NONE = 0, QM_MARKED = 1, QM_EXT = 2, QM_PARKED = 3, QM_IMMEDIATE = 4 };       // values of marks[]
// u32 cursors of the stage: the list of marked reads; the parked tasks in QD_SHARDS lists of their own (one cursor would take ~60 k returning
// atomics on one address per launch: milliseconds)
#define QD_SHARDS 64u
enum { QC_LIST = 0, QC_TASKS = 16, QC_COUNT = 16 + 64 };

struct QTask { uint32_t r, max_ref, ars, head, aq, m, nref, pad; unsigned long long rf_start; };     // 40 bytes

__global__ void __launch_bounds__(1024) k_mark_list(uint32_t n, const uint8_t* __restrict__ marks, uint32_t* __restrict__ list, uint32_t* __restrict__ qc) {
  __shared__ uint32_t s_n, s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  const bool m = r < n && marks[r] == QM_MARKED;
  const unsigned long long b = __ballot(m);
  uint32_t wbase = 0;
  if (lane_id() == 0 && b) wbase = atomicAdd(&s_n, (uint32_t)__popcll(b));
  wbase = (uint32_t)__shfl((int)wbase, 0, 64);
  __syncthreads();
  if (threadIdx.x == 0 && s_n) s_base = atomicAdd(&qc[QC_LIST], s_n);
  __syncthreads();
  if (m) list[s_base + wbase + (uint32_t)__popcll(b & ((1ull << lane_id()) - 1))] = r;
}

__global__ void __launch_bounds__(64) k_quad(DReads rd, DIndex ix, DParams P, int pass, int is_last_strand, RState* __restrict__ work, RWork* __restrict__ rw,
                                             const uint32_t* __restrict__ pool, uint8_t* __restrict__ marks, const uint32_t* __restrict__ list,
                                             uint32_t* __restrict__ qc, QTask* __restrict__ tasks, uint32_t tasks_cap, uint32_t lds_mq, uint32_t lds_rq) {
  __shared__ uint32_t s_hp[4][QD_HITS + 1], s_lo[4][QD_HITS], s_hw[4][QD_HITS];
  __shared__ uint32_t s_seq[4][QD_POS], s_rp[4][QD_POS], s_win[4][QD_POS];
  __shared__ unsigned long long s_cu[4][QD_CAND], s_ck[4][QD_CAND], s_pu[4][QD_PAIRS], s_pr[4][QD_PAIRS];
  const int lane = lane_id(), g = lane >> 4, gl = lane & 15;
  const uint32_t n_list = qc[QC_LIST];
  auto gb = [&](bool p) -> uint32_t { return (uint32_t)((__ballot(p) >> (16 * g)) & 0xFFFFull); };     // the ballot of this lane's group
  for (uint32_t q0 = blockIdx.x * 4u; q0 < n_list; q0 += gridDim.x * 4u) {
    const bool have = q0 + (uint32_t)g < n_list;
    const uint32_t r = have ? list[q0 + g] : 0u;
    RWork w; RState st;
    memset(&w, 0, sizeof w); memset(&st, 0, sizeof st);
    if (have) { w = rw[r]; st = work[r]; }
    const uint32_t len = have ? rd.len[r] : 0u;
    const uint32_t nh = have ? w.hit_total : 0u;
    bool alive = have && nh <= QD_HITS;                  // false: the read is left to k_chain as it is
    bool imm = false;                                    // ... marked for its sequential walk
    // ---- hits: position-list start, length, window position; prefix over the lengths ----
    uint32_t npos = 0;
    for (uint32_t h0 = 0; h0 < QD_HITS; h0 += 16) {
      const uint32_t h = h0 + (uint32_t)gl;
      uint32_t lo = 0, ln = 0, hw = 0;
      if (alive && h < nh) {
        uint32_t o = h;
        for (uint32_t pp = 0; pp < 3; pp++) {              // the hit blocks of the passes run so far on this strand, concatenated
          const uint32_t c = w.blk_cnt[pp];
          if (o < c) { const uint32_t id = pool[w.blk_off[pp] + 2 * o]; hw = pool[w.blk_off[pp] + 2 * o + 1]; lo = ix.pos_off[id]; ln = ix.pos_off[id + 1] - lo; break; }
          o -= c;
        }
      }
      uint32_t inc = ln;
      for (int d = 1; d < 16; d <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)inc, d, 16); if (gl >= d) inc += v; }
      if (alive && h < nh) { s_hp[g][h] = npos + inc - ln; s_lo[g][h] = lo; s_hw[g][h] = hw; }
      npos += (uint32_t)__shfl((int)inc, 15, 16);
    }
    if (alive && gl == 0) s_hp[g][nh] = npos;
    if (npos > QD_POS) alive = false;
    __syncthreads();
    // ---- positions: reference number, reference position, window position of the hit ----
    for (uint32_t p0 = 0; p0 < QD_POS; p0 += 16) {
      const uint32_t p = p0 + (uint32_t)gl;
      if (alive && p < npos) {
        uint32_t h = 0;
        for (uint32_t step = 16; step > 0; step >>= 1) { const uint32_t t = h + step; if (t < nh && s_hp[g][t] <= p) h = t; }
        const uint2 pa = ix.pos_arr[s_lo[g][h] + (p - s_hp[g][h])];
        s_seq[g][p] = pa.y; s_rp[g][p] = pa.x; s_win[g][p] = s_hw[g][h];
      }
    }
    __syncthreads();
    // ---- exact counts per reference (alignment.cpp:117-130); a reference is represented by its first position ----
    uint32_t myseq[4], cnt[4]; bool rep[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const uint32_t p = (uint32_t)gl + 16u * q; myseq[q] = (alive && p < npos) ? s_seq[g][p] : 0xFFFFFFFFu; cnt[q] = 0; rep[q] = alive && p < npos; }
    {
      uint32_t mx = alive ? npos : 0u;
      for (int d = 32; d > 0; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
      for (uint32_t j = 0; j < mx; j++) {
        const uint32_t sj = (alive && j < npos) ? s_seq[g][j] : 0xFFFFFFFEu;
#pragma unroll
        for (int q = 0; q < 4; q++) if (sj == myseq[q]) { cnt[q]++; if (j < (uint32_t)gl + 16u * q) rep[q] = false; }
      }
    }
    // candidates = references with count >= num_seeds, key (~count, ref): count descending, reference ascending (:134-148)
    uint32_t ncand = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const bool c = rep[q] && cnt[q] >= (uint32_t)P.num_seeds;
      const uint32_t b = gb(c);
      if (c) { const uint32_t i = ncand + (uint32_t)__popc(b & ((1u << gl) - 1u)); if (i < QD_CAND) s_cu[g][i] = ((unsigned long long)(0xFFFFFFFFu - cnt[q]) << 32) | myseq[q]; }
      ncand += (uint32_t)__popc(b);
      if (gb(c && cnt[q] > QD_PAIRS)) alive = false;          // a candidate with more pairs than a group sorts
    }
    if (ncand > QD_CAND) alive = false;
    __syncthreads();
    {
      unsigned long long mk = (alive && (uint32_t)gl < ncand) ? s_cu[g][gl] : ~0ull;
      uint32_t rank = 0;
      for (uint32_t j = 0; j < QD_CAND; j++) { const unsigned long long kj = (alive && j < ncand) ? s_cu[g][j] : ~0ull; rank += kj < mk ? 1u : 0u; }
      if (alive && (uint32_t)gl < ncand) s_ck[g][rank] = mk;
    }
    __syncthreads();
    // ---- the candidate loop (:150-508) under "no ssw_align aligns": is_aligned stays 0, `best` is not touched, is_search_candidates stays on ----
    uint32_t n_tasks = 0;
    QTask T1; memset(&T1, 0, sizeof T1);
    const uint64_t rlen = len;
    for (uint32_t k = 0; k < QD_CAND; k++) {
      bool act = alive && k < ncand;                       // (every candidate has >= num_seeds pairs: the rule of :156-159 never ends this loop early)
      const unsigned long long ck = act ? s_ck[g][k] : 0ull;
      const uint32_t max_ref = (uint32_t)ck, np = act ? 0xFFFFFFFFu - (uint32_t)(ck >> 32) : 0u;
      if (!__any(act)) break;
      uint64_t ref0 = 0, reflen = 0;
      if (act) { ref0 = ix.ref_off[max_ref]; reflen = ix.ref_off[max_ref + 1] - ref0; }
      // its (ref_pos, read_pos) pairs (:181-201), sorted
      uint32_t base = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t p = (uint32_t)gl + 16u * q;
        const bool mt = act && p < npos && myseq[q] == max_ref;
        const uint32_t b = gb(mt);
        if (mt) { const uint32_t i = base + (uint32_t)__popc(b & ((1u << gl) - 1u)); if (i < QD_PAIRS) s_pu[g][i] = ((unsigned long long)s_rp[g][p] << 32) | s_win[g][p]; }
        base += (uint32_t)__popc(b);
      }
      __syncthreads();
      {
        const unsigned long long mv = (act && (uint32_t)gl < np) ? s_pu[g][gl] : ~0ull;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < QD_PAIRS; j++) { const unsigned long long vj = (act && j < np) ? s_pu[g][j] : ~0ull; rank += (vj < mv || (vj == mv && j < (uint32_t)gl)) ? 1u : 0u; }
        if (act && (uint32_t)gl < np) s_pr[g][rank] = mv;
      }
      __syncthreads();
      const unsigned long long S = (act && (uint32_t)gl < np) ? s_pr[g][gl] : ~0ull;       // lane j of the group holds pair j
      // the sliding window of read length (:203-506)
      uint32_t it = 0, ms_lo = 0, ms_hi = 0;
      uint32_t begin_ref = (uint32_t)((unsigned long long)__shfl((long long)S, 0, 16) >> 32), begin_read = (uint32_t)(unsigned long long)__shfl((long long)S, 0, 16);
      int pending_pop = 0;
      bool walking = act;
      for (uint32_t iter = 0; iter < 4 * QD_PAIRS + 8; iter++) {
        if (walking && it == np) walking = false;
        if (!__any(walking)) break;
        bool eval = false;
        if (walking && !pending_pop) {
          pending_pop = 1;
          eval = true;
        }
        // push: the pairs whose reference position is within reach of the window's first pair (they are a prefix of what is left)
        const uint64_t end_ref_max = (uint64_t)begin_ref + len - begin_read - P.lnwin + 1;
        const uint32_t pm = gb(eval && (uint32_t)gl >= it && (uint32_t)gl < np && (uint64_t)(uint32_t)(S >> 32) <= end_ref_max);
        if (eval && pm) { it += (uint32_t)__popc(pm); ms_hi = it; }
        const bool lis = eval && (ms_hi - ms_lo) >= (uint32_t)P.num_seeds;
        // longest increasing subsequence of the window's read positions: patience piles across the group's lanes (cf. wave_lis_first)
        uint32_t nl = 0, lis0 = 0;
        if (__any(lis)) {
          const uint32_t nw = lis ? ms_hi - ms_lo : 0u;
          const uint32_t mine = (uint32_t)(unsigned long long)__shfl((long long)S, (int)((ms_lo + (uint32_t)gl) & 15u), 16);
          uint32_t tail = 0, root = 0, nb = 0;
          uint32_t mxw = nw;
          for (int d = 32; d > 0; d >>= 1) mxw = max(mxw, (uint32_t)__shfl_xor((int)mxw, d, 64));
          for (uint32_t i = 0; i < mxw; i++) {
            const bool on = i < nw;
            const uint32_t ai = (uint32_t)__shfl((int)mine, (int)(i & 15u), 16);
            const uint32_t u = (uint32_t)__popc(gb(on && (uint32_t)gl < nb && tail < ai));
            const uint32_t tu = (uint32_t)__shfl((int)tail, (int)(u & 15u), 16);
            const uint32_t ru = (uint32_t)__shfl((int)root, (int)((u - 1u) & 15u), 16);
            const bool place = on && (u == nb || ai < tu);                                  // equal: nothing changes (find_lis :87)
            if (place) {
              const uint32_t rr = u > 0 ? ru : i;
              if ((uint32_t)gl == u) { tail = ai; root = rr; }
              if (u == nb) nb++;
            }
          }
          const uint32_t rl = (uint32_t)__shfl((int)root, (int)((nb - 1u) & 15u), 16);
          nl = nb; lis0 = nb ? rl : 0u;
        }
        const bool task = lis && nl >= (uint32_t)P.min_lis;
        const unsigned long long pl = (unsigned long long)__shfl((long long)S, (int)((ms_lo + lis0) & 15u), 16);      // (every shuffle is executed by the whole wave)
        if (task) {
          const uint32_t lcs_ref_start = (uint32_t)(pl >> 32), lcs_que_start = (uint32_t)pl;
          uint64_t head = 0, tail = 0, align_ref_start = 0, align_que_start = 0, align_length = 0;
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
          n_tasks++;
          if (n_tasks == 1) {
            T1.r = r; T1.max_ref = max_ref; T1.ars = (uint32_t)align_ref_start; T1.head = (uint32_t)head; T1.aq = (uint32_t)align_que_start;
            T1.m = (uint32_t)(int)(align_length - head - tail); T1.nref = (uint32_t)(int)align_length;
            T1.rf_start = ref0 + align_ref_start - head;
          } else { walking = false; act = false; alive = false; imm = true; }     // a second task: k_chain walks the read sequentially
        }
        // pop (:486-506) -- after a task the walk re-enters at the loop head (next_task returns and is called again)
        const bool pop = walking && !task;
        if (pop) {
          pending_pop = 0;
          if (ms_hi > ms_lo) ms_lo++;
          if (ms_hi == ms_lo && it == np) walking = false;
        }
        const uint32_t bi = ms_hi == ms_lo ? it : ms_lo;       // the window's first pair from now on
        const unsigned long long pi = (unsigned long long)__shfl((long long)S, (int)(bi & 15u), 16);
        if (pop && walking) { begin_ref = (uint32_t)(pi >> 32); begin_read = (uint32_t)pi; }
      }
      __syncthreads();
    }
    // ---- what becomes of the read ----
    const bool x4_ok = P.sw_mode >= 1 && len <= SW_X4_MAX_ROWS && sw_pk_fits((int)len, (int)lds_rq, P.match, P.mismatch, P.score_N, P.gap_open);
    bool park = false;
    if (alive && n_tasks == 1) {
      const bool fits = (int)T1.m > 0 && (int)T1.nref > 0 && T1.m <= lds_mq && T1.nref <= lds_rq;
      if (x4_ok && fits) park = true; else { alive = false; imm = true; }
    }
    const unsigned long long pb = __ballot(park && gl == 0);
    const uint32_t shard = blockIdx.x & (QD_SHARDS - 1u), shard_cap = tasks_cap / QD_SHARDS;
    uint32_t tb = 0;
    if (lane == 0 && pb) tb = atomicAdd(&qc[QC_TASKS + shard], (uint32_t)__popcll(pb));
    tb = (uint32_t)__shfl((int)tb, 0, 64);
    if (have && gl == 0) {
      if (alive && n_tasks == 0) {
        chain_finish_read(P, is_last_strand, r, st, w, 1, true, work, rw);       // no ssw_align: the pass ends (paralleltraversal.cpp:253-277)
        marks[r] = QM_NONE;
      } else if (park) {
        const uint32_t ti = tb + (uint32_t)__popcll(pb & ((1ull << lane) - 1));
        if (ti < shard_cap) { tasks[(size_t)shard * shard_cap + ti] = T1; marks[r] = QM_PARKED; }
        else marks[r] = QM_MARKED;                                              // list full: k_chain does it
      } else if (imm) marks[r] = QM_IMMEDIATE;
      // else: stays QM_MARKED
    }
    __syncthreads();
  }
}

// the parked tasks, four per wave
__global__ void __launch_bounds__(64) k_park_sw(DReads rd, DIndex ix, DParams P, int is_last_strand, RState* __restrict__ work, RWork* __restrict__ rw,
                                                uint8_t* __restrict__ marks, const uint32_t* __restrict__ qc, const QTask* __restrict__ tasks, uint32_t tasks_cap,
                                                unsigned long long* __restrict__ ctr, uint32_t lds_mq, uint32_t lds_rq) {
  SMR_DYN_LDS(unsigned char, lds_raw);                    // 4 reads of lds_mq bytes | 4 reference windows of lds_rq bytes
  const int lane = lane_id(), g = lane >> 4, gl = lane & 15;
  // block b works on task list b % QD_SHARDS, every (gridDim / QD_SHARDS)-th quad of it
  const uint32_t shard = blockIdx.x & (QD_SHARDS - 1u), shard_cap = tasks_cap / QD_SHARDS;
  const uint32_t n_tasks = min(qc[QC_TASKS + shard], shard_cap);
  tasks += (size_t)shard * shard_cap;
  uint8_t* rq = lds_raw + (size_t)g * lds_mq;
  uint8_t* fq = lds_raw + (size_t)4 * lds_mq + (size_t)g * lds_rq;
  unsigned long long n_fwd = 0, n_cells = 0;
  for (uint32_t t0 = (blockIdx.x / QD_SHARDS) * 4u; t0 < n_tasks; t0 += max(gridDim.x / QD_SHARDS, 1u) * 4u) {
    const bool have = t0 + (uint32_t)g < n_tasks;
    QTask T; memset(&T, 0, sizeof T);
    RWork w; memset(&w, 0, sizeof w);
    bool hn = false;
    if (have) {
      T = tasks[t0 + g];
      w = rw[T.r];
      const uint32_t len = rd.len[T.r];
      const uint32_t* rec = rd.words + rd.rec_off[T.r];
      const uint32_t aval = (w.has_amb && !w.is04) ? 4u : (uint32_t)w.aval;             // read.flip34() before SSW (:360-361)
      for (uint32_t q = gl; q < len; q += 16) rq[q] = (uint8_t)read_nt(rec, len, q, w.reversed, aval);
      for (uint32_t q = gl; q < T.nref; q += 16) { const uint8_t ch = ix.ref_seq[T.rf_start + q]; fq[q] = ch; hn |= ch == 4; }
    }
    const bool hasn = __any(hn);
    __syncthreads();
    int mm = have ? (int)T.m : 0;
    for (int d = 32; d > 0; d >>= 1) mm = max(mm, __shfl_xor(mm, d, 64));
    const SwRes r4 = sw_wave_x4(rq, have ? (int)T.m : 0, (int)T.aq, 1, fq, have ? (int)T.nref : 0, 0, 1, P.match, P.mismatch, P.score_N, P.gap_open, P.gap_ext, mm, hasn);
    if (have && gl == 0) {
      const int score1 = r4.score > 65535 ? 65535 : r4.score;
      if ((uint32_t)score1 > P.minimal_score) marks[T.r] = QM_IMMEDIATE;              // it aligns: the sequential walk records it (and whatever follows)
      else {
        // no alignment: the read ends as the sequential walk ends it -- one ssw_align call, nothing recorded
        RState st = work[T.r];
        if (w.has_amb && !w.is04) { w.is04 = 1; w.aval = 4; }
        chain_finish_read(P, is_last_strand, T.r, st, w, 1, true, work, rw);
        marks[T.r] = QM_NONE;
        n_fwd++; n_cells += (unsigned long long)T.m * T.nref;
      }
    }
    __syncthreads();
  }
  for (int d = 32; d > 0; d >>= 1) { n_fwd += __shfl_xor(n_fwd, d, 64); n_cells += __shfl_xor(n_cells, d, 64); }
  if (lane == 0 && n_fwd) { ctr_add(ctr, C_SW_FWD, n_fwd); ctr_add(ctr, C_SW_CELLS, n_cells); atomicAdd(&ctr[C_SW_SPEC], n_fwd); atomicAdd(&ctr[C_SW_SPEC_USED], n_fwd); }
}

}  // namespace smr
