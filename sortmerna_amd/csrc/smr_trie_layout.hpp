// smr_trie_layout.hpp -- layout of ONE mini burst trie in the compact arena format (smr_host.hpp), shared by the host index
// builder (smr_index.cpp) and the device index builder (smr_ibuild.hpp): the same function compiled for both sides, so the two
// builders cannot disagree.
//
// Input: the entries of one 9-mer key and direction, e[i] = (tail << 32) | id, sorted by tail; a tail is T nt, first nt in the
// most significant position.  Output (words relative to the mini-trie's first word; everything 16-byte aligned):
//   node   = 4 elements (A,C,G,T), element = flag:2 | nent:8 | off:22  (flag 0 empty, 1 child node at off, 2 bucket at off)
//   bucket = nent x {remaining tail nt (first nt in the low bits), id}, padded with zero words to a multiple of 4 words
// in depth-first order: a node's 4 words are reserved when it is reached, then its elements are laid out A,C,G,T, each child
// completely before the next element.  An element bursts into a child node when it holds more than 16 entries and
// depth + 1 < burst_depth (the reference bursts buckets above 128 bytes = 16 entries while depth < 7, indexdb.cpp:225-228).
#pragma once
#include <stdint.h>

#include "smr_host.hpp"

#ifndef SMR_HD
#ifdef __HIPCC__
#define SMR_HD __host__ __device__
#else
#define SMR_HD
#endif
#endif

namespace smr {

enum { TRIE_OK = 0, TRIE_ERR_BUCKET = 1, TRIE_ERR_SIZE = 2 };

SMR_HD inline uint32_t trie_nt_at(uint64_t tail, int T, int k) { return (uint32_t)(tail >> (2 * (T - 1 - k))) & 3u; }

// EMIT = false: only the size (words) and the node / bucket counts; EMIT = true: also writes out[0 .. size)
template <bool EMIT>
SMR_HD inline uint32_t minitrie_layout(const uint64_t* e, uint32_t n, int T, int burst_depth, uint32_t* out, uint32_t* n_nodes, uint32_t* n_buckets, int* status) {
  struct Frame { uint32_t lo, hi, node_off; int depth, c; };
  Frame st[24];
  int sp = 0;
  uint32_t cursor = 4, nodes = 1, buckets = 0;
  st[0] = Frame{0, n, 0, 0, 0};
  *status = TRIE_OK;
  while (sp >= 0) {
    Frame& f = st[sp];
    if (f.c == 4) { sp--; continue; }
    const int c = f.c++;
    const uint32_t p = f.lo;
    uint32_t q = p;
    while (q < f.hi && trie_nt_at(e[q] >> 32, T, f.depth) == (uint32_t)c) q++;
    f.lo = q;
    const uint32_t word = f.node_off + (uint32_t)c, cnt = q - p;
    if (cnt == 0) { if (EMIT) out[word] = 0; continue; }
    if (cursor > ELEM_OFF_MASK) { *status = TRIE_ERR_SIZE; return 0; }
    if (cnt > 16 && f.depth + 1 < burst_depth) {
      if (EMIT) out[word] = (1u << ELEM_FLAG_SHIFT) | cursor;
      const int d = f.depth + 1;
      st[++sp] = Frame{p, q, cursor, d, 0};          // (f is not used after this)
      cursor += 4; nodes++;
      continue;
    }
    if (cnt > ELEM_NENT_MAX) { *status = TRIE_ERR_BUCKET; return 0; }
    if (EMIT) {
      out[word] = (2u << ELEM_FLAG_SHIFT) | (cnt << ELEM_NENT_SHIFT) | cursor;
      const int s = T - 1 - f.depth;                 // remaining characters per entry
      for (uint32_t i = p; i < q; i++) {
        const uint64_t tail = e[i] >> 32;
        uint32_t enc = 0;
        for (int k = 0; k < s; k++) enc |= trie_nt_at(tail, T, f.depth + 1 + k) << (2 * k);
        out[cursor + 2 * (i - p)] = enc; out[cursor + 2 * (i - p) + 1] = (uint32_t)e[i];
      }
      if (cnt & 1u) { out[cursor + 2 * cnt] = 0; out[cursor + 2 * cnt + 1] = 0; }
    }
    cursor += (2 * cnt + 3) & ~3u;
    buckets++;
  }
  if (n_nodes) *n_nodes = nodes;
  if (n_buckets) *n_buckets = buckets;
  return cursor;
}

}  // namespace smr
