/*
 * smr_oracle.c -- TEST INFRASTRUCTURE, NOT THE PRODUCT.  See smr_oracle.h for scope and pinning.
 *
 * A CPU restatement, in plain C99, of the SortMeRNA v5.0.0 per-read hot path.  Every function
 * cites the reference file:line it follows (paths relative to /root/reference).  Nothing here is
 * copied from the reference: the control flow is restated over flat C arrays, the SSE2 striped
 * Smith-Waterman is restated as a lane-by-lane scalar emulation, std::map/std::sort/std::deque
 * are replaced by sort + run-length counting / index windows with the same observable order.
 */
#include "smr_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */
static void* xcalloc(size_t n, size_t sz) {
  void* p = calloc(n ? n : 1, sz ? sz : 1);
  if (!p) { fprintf(stderr, "smr_oracle: out of memory (calloc of %zu x %zu bytes)\n", n, sz); exit(1); }
  return p;
}
static void* xrealloc(void* q, size_t sz) {
  void* p = realloc(q, sz ? sz : 1);
  if (!p) { fprintf(stderr, "smr_oracle: out of memory (realloc to %zu bytes)\n", sz); exit(1); }
  return p;
}

/* include/common.hpp:68-77 nt_table: A/a C/c G/g T/t U/u -> 0 1 2 3 3, everything else 4 */
static int nt_code(int c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* LEV(1) universal automaton tables: traverse_bursttrie.cpp:68-98 (published automaton data).  */
/* table[0]: 4-bit vectors; table[1..3]: 3-, 2-, 1-bit vectors of the last three positions.      */
/* ------------------------------------------------------------------------------------------ */
static const uint8_t LEV[4][16][14] = {
  {{3,14,14,14,14,14,14,14,14,14,14,14,14,14},
   {3,14,14,14,14,14,14,14,14,14,14,14,14,14},
   {7,14,14,14,4,4,4,4,14,14,14,14,14,14},
   {7,14,14,14,4,4,4,4,14,14,14,14,14,14},
   {0,14,2,2,14,14,2,2,14,14,14,14,14,14},
   {0,14,2,2,14,14,2,2,14,14,14,14,14,14},
   {0,14,2,2,4,4,6,6,14,14,14,14,14,14},
   {0,14,2,2,4,4,6,6,14,14,14,14,14,14},
   {3,1,14,1,14,1,14,1,14,14,14,14,14,14},
   {3,1,14,1,14,1,14,1,14,14,14,14,14,14},
   {7,1,14,1,4,5,4,5,14,14,14,14,14,14},
   {7,1,14,1,4,5,4,5,14,14,14,14,14,14},
   {0,1,2,3,14,1,2,3,14,14,14,14,14,14},
   {0,1,2,3,14,1,2,3,14,14,14,14,14,14},
   {0,1,2,3,4,5,6,7,14,14,14,14,14,14},
   {0,1,2,3,4,5,6,7,14,14,14,14,14,14}},
  {{3,14,14,14,14,14,14,14,14,14,14,14,14,14},
   {13,14,14,14,10,10,10,10,14,14,14,14,14,14},
   {8,14,2,2,14,14,2,2,14,14,14,14,14,14},
   {8,14,2,2,10,10,12,12,14,14,14,14,14,14},
   {3,1,14,1,14,1,14,1,14,14,14,14,14,14},
   {13,1,14,1,10,11,10,11,14,14,14,14,14,14},
   {8,1,2,3,14,1,2,3,14,14,14,14,14,14},
   {8,1,2,3,10,11,12,13,14,14,14,14,14,14},
   {0},{0},{0},{0},{0},{0},{0},{0}},
  {{12,14,14,14,14,14,14,14,12,14,14,14,14,14},
   {9,14,10,10,14,14,10,10,9,14,14,14,10,10},
   {12,1,14,1,14,1,14,1,12,14,14,1,14,1},
   {9,1,10,12,14,1,10,12,9,14,14,1,10,12},
   {0},{0},{0},{0},{0},{0},{0},{0},{0},{0},{0},{0}},
  {{10,14,14,14,14,14,14,14,14,10,14,14,14,14},
   {10,10,14,10,14,10,14,10,14,10,14,14,10,14},
   {0},{0},{0},{0},{0},{0},{0},{0},{0},{0},{0},{0},{0},{0}}
};

/* ------------------------------------------------------------------------------------------ */
/* Index part in the reference's in-memory shape: index.cpp:143-357, indexdb.hpp:67-104         */
/* ------------------------------------------------------------------------------------------ */
typedef struct orc_elem {
  uint8_t flag;              /* 0 empty, 1 trie node, 2 bucket */
  uint32_t size;             /* bucket bytes */
  struct orc_elem* trie;     /* flag 1: child node (4 elements) */
  uint8_t* bucket;           /* flag 2: entries {u32 tail, u32 id} */
} orc_elem;

typedef struct { uint32_t count; orc_elem* trie_F; orc_elem* trie_R; } orc_kmer;
typedef struct { uint32_t* arr; uint32_t size; } orc_origin;   /* arr = size x {pos, seq} */

struct orc_index {
  uint32_t lnwin, nkmers;
  orc_kmer* lookup;
  uint32_t number_elements;
  orc_origin* positions;
  /* arenas to free */
  void** blocks; size_t nblocks, capblocks;
};

static void idx_keep(orc_index* ix, void* p) {
  if (ix->nblocks == ix->capblocks) {
    ix->capblocks = ix->capblocks ? ix->capblocks * 2 : 1024;
    ix->blocks = (void**)xrealloc(ix->blocks, ix->capblocks * sizeof(void*));
  }
  ix->blocks[ix->nblocks++] = p;
}

typedef struct { const uint8_t* p; size_t n, o; } rdbuf;
static uint8_t* slurp(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  uint8_t* b = (uint8_t*)xcalloc((size_t)sz + 1, 1);
  if (sz > 0 && fread(b, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(b); return NULL; }
  fclose(f); *n = (size_t)sz; return b;
}
static uint32_t rd_u32(rdbuf* r) { uint32_t v = 0; if (r->o + 4 <= r->n) memcpy(&v, r->p + r->o, 4); r->o += 4; return v; }
static uint8_t rd_u8(rdbuf* r) { uint8_t v = 0; if (r->o < r->n) v = r->p[r->o]; r->o += 1; return v; }

/* BFS rebuild of one mini burst trie from the on-disk stream: index.cpp:176-316.  The stream is:
 * 4 flag bytes of the root node; then for every dequeued element: flag 1 -> 4 flag bytes of the
 * child; flag 2 -> u32 bucket_bytes + entries (writer indexdb.cpp:774-865). */
static orc_elem* load_trie(orc_index* ix, rdbuf* r) {
  size_t capn = 16, nn = 1, head = 0;
  orc_elem** nodes = (orc_elem**)xcalloc(capn, sizeof(orc_elem*));
  size_t capf = 64, nf = 0, hf = 0;
  uint8_t* flags = (uint8_t*)xcalloc(capf, 1);
  orc_elem* root = (orc_elem*)xcalloc(4, sizeof(orc_elem));
  idx_keep(ix, root);
  nodes[0] = root;
  for (int i = 0; i < 4; i++) flags[nf++] = rd_u8(r);
  while (head < nn) {
    orc_elem* node = nodes[head++];
    for (int i = 0; i < 4; i++) {
      uint8_t flag = flags[hf++];
      switch (flag) {
        case 0: node->flag = 0; node->size = 0; node->trie = NULL; break;
        case 1: {
          if (nf + 4 > capf) { capf *= 2; flags = (uint8_t*)xrealloc(flags, capf); }
          for (int k = 0; k < 4; k++) flags[nf++] = rd_u8(r);
          orc_elem* child = (orc_elem*)xcalloc(4, sizeof(orc_elem));
          idx_keep(ix, child);
          node->flag = 1; node->size = 0; node->trie = child;
          if (nn == capn) { capn *= 2; nodes = (orc_elem**)xrealloc(nodes, capn * sizeof(orc_elem*)); }
          nodes[nn++] = child;
        } break;
        case 2: {
          uint32_t sz = rd_u32(r);
          uint8_t* b = (uint8_t*)xcalloc(sz, 1);
          idx_keep(ix, b);
          if (r->o + sz <= r->n) memcpy(b, r->p + r->o, sz);
          r->o += sz;
          node->flag = 2; node->bucket = b; node->size = sz;
        } break;
        default:
          fprintf(stderr, "smr_oracle: bad trie flag %d\n", flag); exit(1);
      }
      node++;
    }
  }
  free(nodes); free(flags);
  return root;
}

orc_index* orc_index_load(const char* prefix, uint32_t part, uint32_t lnwin) {
  char path[4096];
  orc_index* ix = (orc_index*)xcalloc(1, sizeof(orc_index));
  ix->lnwin = lnwin;
  ix->nkmers = 1u << lnwin;          /* index.cpp:155: limit = 1 << lnwin  (= 4^(L/2)) */
  ix->lookup = (orc_kmer*)xcalloc(ix->nkmers, sizeof(orc_kmer));
  size_t n = 0;
  /* STEP 1 kmer counts: index.cpp:146-161 */
  snprintf(path, sizeof path, "%s.kmer_%u.dat", prefix, part);
  uint8_t* kb = slurp(path, &n);
  if (!kb) { fprintf(stderr, "smr_oracle: cannot read %s\n", path); orc_index_free(ix); return NULL; }
  for (uint32_t i = 0; i < ix->nkmers && (size_t)(i + 1) * 4 <= n; i++) memcpy(&ix->lookup[i].count, kb + (size_t)i * 4, 4);
  free(kb);
  /* STEP 2 burst tries: index.cpp:163-320 */
  snprintf(path, sizeof path, "%s.bursttrie_%u.dat", prefix, part);
  uint8_t* tb = slurp(path, &n);
  if (!tb) { fprintf(stderr, "smr_oracle: cannot read %s\n", path); orc_index_free(ix); return NULL; }
  rdbuf r = { tb, n, 0 };
  for (uint32_t i = 0; i < ix->nkmers && r.o < r.n; i++) {
    uint32_t sz[2]; sz[0] = rd_u32(&r); sz[1] = rd_u32(&r);
    if (ix->lookup[i].count != 0) {
      for (int j = 0; j < 2; j++) {
        orc_elem* t = sz[j] != 0 ? load_trie(ix, &r) : NULL;
        if (j == 0) ix->lookup[i].trie_F = t; else ix->lookup[i].trie_R = t;
      }
    }
  }
  free(tb);
  /* STEP 3 positions: index.cpp:322-352 */
  snprintf(path, sizeof path, "%s.pos_%u.dat", prefix, part);
  uint8_t* pb = slurp(path, &n);
  if (!pb) { fprintf(stderr, "smr_oracle: cannot read %s\n", path); orc_index_free(ix); return NULL; }
  rdbuf q = { pb, n, 0 };
  ix->number_elements = rd_u32(&q);
  ix->positions = (orc_origin*)xcalloc(ix->number_elements, sizeof(orc_origin));
  for (uint32_t i = 0; i < ix->number_elements; i++) {
    uint32_t sz = rd_u32(&q);
    ix->positions[i].size = sz;
    ix->positions[i].arr = (uint32_t*)xcalloc((size_t)sz * 2, 4);
    if (q.o + (size_t)sz * 8 <= q.n) memcpy(ix->positions[i].arr, q.p + q.o, (size_t)sz * 8);
    q.o += (size_t)sz * 8;
  }
  free(pb);
  return ix;
}

void orc_index_free(orc_index* ix) {
  if (!ix) return;
  for (size_t i = 0; i < ix->nblocks; i++) free(ix->blocks[i]);
  free(ix->blocks);
  if (ix->positions) for (uint32_t i = 0; i < ix->number_elements; i++) free(ix->positions[i].arr);
  free(ix->positions);
  free(ix->lookup);
  free(ix);
}
uint32_t orc_index_num_ids(const orc_index* ix) { return ix->number_elements; }
uint32_t orc_index_positions(const orc_index* ix, uint32_t id, uint32_t* out, uint32_t cap_pairs) {
  if (id >= ix->number_elements) return 0;
  uint32_t n = ix->positions[id].size;
  for (uint32_t i = 0; i < n && i < cap_pairs; i++) { out[2 * i] = ix->positions[id].arr[2 * i]; out[2 * i + 1] = ix->positions[id].arr[2 * i + 1]; }
  return n;
}

/* ------------------------------------------------------------------------------------------ */
/* References::load  references.cpp:55-159 (FASTA only; nt_table -> 0..4, spaces kept out)       */
/* ------------------------------------------------------------------------------------------ */
struct orc_refs { uint32_t n; char** seq; uint32_t* len; };

orc_refs* orc_refs_load(const char* fasta, uint64_t start_part, uint32_t numseq_part) {
  size_t n = 0;
  uint8_t* b = slurp(fasta, &n);
  if (!b) { fprintf(stderr, "smr_oracle: cannot read %s\n", fasta); return NULL; }
  orc_refs* rf = (orc_refs*)xcalloc(1, sizeof(orc_refs));
  rf->seq = (char**)xcalloc(numseq_part, sizeof(char*));
  rf->len = (uint32_t*)xcalloc(numseq_part, sizeof(uint32_t));
  size_t o = (size_t)start_part;
  char* cur = NULL; size_t curlen = 0, curcap = 0; int have = 0;
  while (o < n && rf->n < numseq_part) {
    size_t e = o;
    while (e < n && b[e] != '\n') e++;
    size_t le = e;                                   /* strip trailing whitespace, references.cpp:104 */
    while (le > o && (b[le - 1] == ' ' || b[le - 1] == '\r' || b[le - 1] == '\t' || b[le - 1] == '\f' || b[le - 1] == '\v')) le--;
    if (le > o) {
      if (b[o] == '>') {
        if (have) { rf->seq[rf->n] = cur; rf->len[rf->n] = (uint32_t)curlen; rf->n++; cur = NULL; curlen = curcap = 0; }
        have = 1;
      } else if (have) {
        if (curlen + (le - o) + 1 > curcap) { curcap = (curlen + (le - o) + 1) * 2; cur = (char*)xrealloc(cur, curcap); }
        for (size_t k = o; k < le; k++) {
          /* convert_fix references.cpp:162-169: every char except ' ' goes through nt_table */
          cur[curlen++] = (b[k] != 32) ? (char)nt_code(b[k]) : (char)32;
        }
      }
    }
    o = e + 1;
  }
  if (have && rf->n < numseq_part) { rf->seq[rf->n] = cur; rf->len[rf->n] = (uint32_t)curlen; rf->n++; cur = NULL; }
  free(cur);
  free(b);
  return rf;
}
void orc_refs_free(orc_refs* rf) { if (!rf) return; for (uint32_t i = 0; i < rf->n; i++) free(rf->seq[i]); free(rf->seq); free(rf->len); free(rf); }
uint32_t orc_refs_count(const orc_refs* rf) { return rf->n; }
uint32_t orc_refs_len(const orc_refs* rf, uint32_t i) { return rf->len[i]; }
const char* orc_refs_seq(const orc_refs* rf, uint32_t i) { return rf->seq[i]; }

/* ------------------------------------------------------------------------------------------ */
/* .stats: writer indexdb.cpp:2033-2080, reader refstats.cpp:103-186                            */
/* ------------------------------------------------------------------------------------------ */
int orc_stats_load(const char* prefix, orc_stats* st) {
  char path[4096];
  snprintf(path, sizeof path, "%s.stats", prefix);
  size_t n = 0; uint8_t* b = slurp(path, &n);
  if (!b) return -1;
  size_t o = 0;
  memset(st, 0, sizeof *st);
  memcpy(&st->filesize, b + o, 8); o += 8;
  uint32_t namelen; memcpy(&namelen, b + o, 4); o += 4; o += namelen;
  memcpy(st->bg, b + o, 32); o += 32;
  memcpy(&st->full_len, b + o, 8); o += 8;
  memcpy(&st->lnwin, b + o, 4); o += 4;
  memcpy(&st->numseq, b + o, 8); o += 8;
  memcpy(&st->nparts, b + o, 2); o += 2;
  for (uint16_t j = 0; j < st->nparts && j < 256; j++) {   /* index_parts_stats {ulong, ulong, u32(+pad)} = 24 B */
    memcpy(&st->part_start[j], b + o, 8); memcpy(&st->part_bytes[j], b + o + 8, 8); memcpy(&st->part_numseq[j], b + o + 16, 4);
    o += 24;
  }
  free(b);
  return o <= n ? 0 : -1;
}

/* refstats.cpp:238-265 */
uint32_t orc_minimal_score(double lambda, double K, const double bg[4], uint64_t full_ref, uint64_t numseq,
                           uint64_t all_reads_count, uint64_t all_reads_len, double evalue,
                           uint64_t* full_ref_corr, uint64_t* full_read_corr) {
  double H = -(bg[0] * log2(bg[0]) + bg[1] * log2(bg[1]) + bg[2] * log2(bg[2]) + bg[3] * log2(bg[3]));
  uint64_t full_read = all_reads_len;
  uint64_t expect_L = (uint64_t)(log(K * full_ref * full_read / 1) / H);
  if (full_ref > expect_L * numseq) full_ref -= expect_L * numseq;
  full_read -= expect_L * all_reads_count / 1;
  uint32_t ms = (uint32_t)(log(evalue / ((double)K * full_ref * full_read / 1)) / -lambda);
  if (full_ref_corr) *full_ref_corr = full_ref;
  if (full_read_corr) *full_read_corr = full_read;
  return ms;
}

/* ------------------------------------------------------------------------------------------ */
/* id_win hit list                                                                             */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint32_t id, win; } idwin;
typedef struct { idwin* v; uint32_t n, cap; } hitvec;
static void hv_push(hitvec* h, uint32_t id, uint32_t win) {
  if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 16; h->v = (idwin*)xrealloc(h->v, h->cap * sizeof(idwin)); }
  h->v[h->n].id = id; h->v[h->n].win = win; h->n++;
}

/* bitvector.cpp:57-91 (dir=+1) and :99-132 (dir=-1): characteristic bit-vectors of the other
 * half-window.  bv is (partialwin-2)*4 bytes, zeroed by the caller. */
static void init_win(const uint8_t* p, int dir, uint8_t* bv, int numbvs) {
  uint8_t* b000 = bv; uint8_t* b010 = bv + 4;
  for (int bitn = 2; bitn >= 0; bitn--) { b000[*p] |= (uint8_t)(1 << bitn); p += dir; }
  uint8_t* setbit = b010; uint8_t* w1 = b000; uint8_t* w2 = b010;
  for (int i = 1; i <= numbvs; i++) {
    *w2 = (uint8_t)((*w1++) << 1);
    *w2++ &= 15;
    if (!(i & 3)) { setbit[*p] |= 1; p += dir; setbit = w2; }
  }
}

/* traverse_bursttrie.cpp:100-298.  Returns through *accept_zero / hits exactly as the reference. */
static void traversetrie_align(const orc_elem* trie_t, uint32_t lev_t, uint8_t depth,
                               const uint8_t* win_k1_ptr, const uint8_t* win_k1_full,
                               int* accept_zero_kmer, hitvec* id_hits, uint32_t win_num,
                               uint32_t partialwin, int is_full_search, orc_counters* ctr) {
  uint32_t pivot = lev_t;
  if (ctr) ctr->n_node++;
  for (uint32_t ne = 0; ne < 4; ne++) {
    uint8_t value = trie_t->flag;
    if (value == 0) { lev_t = pivot; trie_t++; continue; }
    if (depth < partialwin - 2)
      lev_t = LEV[0][win_k1_ptr[(depth << 2) + ne]][lev_t];                         /* :131-135 */
    else
      lev_t = LEV[3 - partialwin + depth][win_k1_full[ne] & ((2u << (partialwin - depth)) - 1)][lev_t]; /* :136-139 */
    if (lev_t == 14) { lev_t = pivot; trie_t++; continue; }
    if (value == 1) {                                                               /* :152-175 */
      traversetrie_align(trie_t->trie, lev_t, (uint8_t)(depth + 1), win_k1_ptr, win_k1_full,
                         accept_zero_kmer, id_hits, win_num, partialwin, is_full_search, ctr);
      if (*accept_zero_kmer) return;
      lev_t = pivot; trie_t++;
    } else {                                                                        /* :178-291 */
      uint32_t bucket_pivot = lev_t;
      uint32_t s = partialwin - depth;
      const uint8_t* sb = trie_t->bucket;
      const uint8_t* eb = sb + trie_t->size;
      while (sb != eb) {
        uint32_t depth_b = depth;
        lev_t = bucket_pivot;
        int local_accept = 0;
        uint32_t entry_str; memcpy(&entry_str, sb, 4);
        if (ctr) ctr->n_entry++;
        for (uint32_t j = 0; j < s; j++) {
          uint32_t nt = entry_str & 3;
          depth_b++;
          if (depth_b < partialwin - 2)
            lev_t = LEV[0][win_k1_ptr[(depth_b << 2) + nt]][lev_t];
          else
            lev_t = LEV[3 - partialwin + depth_b][win_k1_full[nt] & ((2u << (partialwin - depth_b)) - 1)][lev_t];
          if (lev_t == 14) break;
          if (depth_b >= partialwin - 2) {
            if (lev_t >= 8) local_accept = 1;
            if (depth_b == partialwin - 1 && lev_t == 9) {
              *accept_zero_kmer = 1;
              if (is_full_search) *accept_zero_kmer = 0;
            }
          }
          if (local_accept) {
            uint32_t id; memcpy(&id, sb + 4, 4);
            if (*accept_zero_kmer) { id_hits->n = 0; hv_push(id_hits, id, win_num); return; }
            if (id_hits->n) {
              int found = 0;
              for (uint32_t f = 0; f < id_hits->n; f++) if (id_hits->v[f].id == id) { found = 1; break; }
              if (found) break;
            }
            hv_push(id_hits, id, win_num);
          }
          entry_str >>= 2;
        }
        sb += 8;
      }
      lev_t = pivot; trie_t++;
    }
  }
}

/* The LEV(1) automaton of traversetrie_align() run over ONE complete candidate string: pattern = the partialwin chars the
 * bit-vectors are built from (2 bits/char, char i at bits 2i), text = partialwin+1 chars of a trie path + bucket tail.
 * Node and bucket levels use the same transitions; acceptance is only tested at depth >= partialwin-2 (:229-246).
 * Returns bit0 = accepted at some step, bit1 = state 9 at depth partialwin-1 (0-error), bits 8.. = first accepting depth. */
uint32_t orc_lev_accepts(uint32_t pchars, uint32_t tchars, uint32_t partialwin) {
  uint8_t p[16], bv[64];
  memset(bv, 0, sizeof bv);
  for (uint32_t i = 0; i < partialwin; i++) p[i] = (uint8_t)((pchars >> (2 * i)) & 3);
  init_win(p, 1, bv, (int)(4 * (partialwin - 3)));
  const uint8_t* win_k1_ptr = bv;
  const uint8_t* win_k1_full = bv + (partialwin - 3) * 4;
  uint32_t lev_t = 0, res = 0;
  for (uint32_t depth_b = 0; depth_b <= partialwin; depth_b++) {
    uint32_t nt = (tchars >> (2 * depth_b)) & 3;
    if (depth_b < partialwin - 2) lev_t = LEV[0][win_k1_ptr[(depth_b << 2) + nt]][lev_t];
    else lev_t = LEV[3 - partialwin + depth_b][win_k1_full[nt] & ((2u << (partialwin - depth_b)) - 1)][lev_t];
    if (lev_t == 14) break;
    if (depth_b >= partialwin - 2) {
      if (lev_t >= 8 && !(res & 1)) res |= 1u | (depth_b << 8);
      if (depth_b == partialwin - 1 && lev_t == 9) res |= 2u;
    }
  }
  return res;
}

/* number of leading text chars the automaton survives (state != 14): 0 .. partialwin+1 */
uint32_t orc_lev_alive_depth(uint32_t pchars, uint32_t tchars, uint32_t partialwin) {
  uint8_t p[16], bv[64];
  memset(bv, 0, sizeof bv);
  for (uint32_t i = 0; i < partialwin; i++) p[i] = (uint8_t)((pchars >> (2 * i)) & 3);
  init_win(p, 1, bv, (int)(4 * (partialwin - 3)));
  const uint8_t* win_k1_ptr = bv;
  const uint8_t* win_k1_full = bv + (partialwin - 3) * 4;
  uint32_t lev_t = 0;
  for (uint32_t depth_b = 0; depth_b <= partialwin; depth_b++) {
    uint32_t nt = (tchars >> (2 * depth_b)) & 3;
    if (depth_b < partialwin - 2) lev_t = LEV[0][win_k1_ptr[(depth_b << 2) + nt]][lev_t];
    else lev_t = LEV[3 - partialwin + depth_b][win_k1_full[nt] & ((2u << (partialwin - depth_b)) - 1)][lev_t];
    if (lev_t == 14) return depth_b;
  }
  return partialwin + 1;
}

/* hash of a partialwin-mer, MSB first: read.cpp:601-611 */
static uint32_t hash_kmer(const uint8_t* s, uint32_t len) {
  uint32_t h = 0;
  for (uint32_t i = 0; i < len; i++) h = (h << 2) | s[i];
  return h;
}

/* one window: paralleltraversal.cpp:131-240 */
static void window_search(const orc_index* ix, const uint8_t* iseq, uint32_t win_pos, uint32_t lnwin,
                          uint32_t minoccur, int is_full_search, hitvec* id_hits, int* accept_zero,
                          orc_counters* ctr) {
  uint32_t partialwin = lnwin / 2;
  uint32_t numbvs = 4 * (partialwin - 3);              /* refstats.cpp:156 */
  uint32_t bitvec_size = (partialwin - 2) << 2;        /* paralleltraversal.cpp:107 */
  uint32_t offset = (partialwin - 3) << 2;             /* :110 */
  uint8_t bitvec[64];
  *accept_zero = 0;
  memset(bitvec, 0, bitvec_size);
  init_win(iseq + win_pos + partialwin, +1, bitvec, (int)numbvs);
  uint32_t keyf = hash_kmer(iseq + win_pos, partialwin);
  if (ctr) ctr->n_lookup++;
  if (ix->lookup[keyf].count > minoccur && ix->lookup[keyf].trie_F != NULL)
    traversetrie_align(ix->lookup[keyf].trie_F, 0, 0, bitvec, bitvec + offset, accept_zero, id_hits, win_pos,
                       partialwin, is_full_search, ctr);
  if (!*accept_zero) {
    memset(bitvec, 0, bitvec_size);
    init_win(iseq + win_pos + partialwin - 1, -1, bitvec, (int)numbvs);
    uint32_t keyr = hash_kmer(iseq + win_pos + partialwin, partialwin);
    if (ctr) ctr->n_lookup++;
    if (ix->lookup[keyr].count > minoccur && ix->lookup[keyr].trie_R != NULL)
      traversetrie_align(ix->lookup[keyr].trie_R, 0, 0, bitvec, bitvec + offset, accept_zero, id_hits, win_pos,
                         partialwin, is_full_search, ctr);
  }
}

uint32_t orc_window_hits(const orc_index* ix, const uint8_t* iseq, uint32_t win_pos, uint32_t lnwin,
                         uint32_t minoccur, int is_full_search, uint32_t* ids, uint32_t cap, int* zero_err) {
  hitvec h = { 0, 0, 0 }; int az = 0;
  window_search(ix, iseq, win_pos, lnwin, minoccur, is_full_search, &h, &az, NULL);
  for (uint32_t i = 0; i < h.n && i < cap; i++) ids[i] = h.v[i].id;
  if (zero_err) *zero_err = az;
  uint32_t n = h.n; free(h.v); return n;
}

/* ------------------------------------------------------------------------------------------ */
/* Striped Smith-Waterman, restated as a lane-by-lane scalar emulation of the SSE2 kernels.     */
/* ------------------------------------------------------------------------------------------ */
typedef struct { uint16_t score; int32_t ref, read; } aln_end;

static inline uint8_t adds_u8(uint8_t a, uint8_t b) { unsigned s = (unsigned)a + b; return (uint8_t)(s > 255 ? 255 : s); }
static inline uint8_t subs_u8(uint8_t a, uint8_t b) { return (uint8_t)(a > b ? a - b : 0); }
static inline uint8_t max_u8(uint8_t a, uint8_t b) { return a > b ? a : b; }

/* ssw.c:116-141 qP_byte: striped profile, lane segNum of vector i holds read position i+segNum*segLen */
static uint8_t* qp_byte(const int8_t* read, const int8_t* mat, int32_t readLen, int32_t n, uint8_t bias) {
  int32_t segLen = (readLen + 15) / 16;
  uint8_t* t = (uint8_t*)xcalloc((size_t)n * segLen * 16, 1);
  uint8_t* p = t;
  for (int32_t nt = 0; nt < n; nt++)
    for (int32_t i = 0; i < segLen; i++) {
      int32_t j = i;
      for (int32_t seg = 0; seg < 16; seg++) { *p++ = j >= readLen ? bias : (uint8_t)(mat[nt * n + read[j]] + bias); j += segLen; }
    }
  return t;
}

/* ssw.c:150-373 sw_sse2_byte */
static aln_end sw_byte(const int8_t* ref, int ref_dir, int32_t refLen, int32_t readLen, uint8_t gapO, uint8_t gapE,
                       const uint8_t* prof, uint8_t terminate, uint8_t bias) {
  uint8_t max = 0;
  int32_t end_read = readLen - 1, end_ref = -1;
  int32_t segLen = (readLen + 15) / 16;
  size_t vb = (size_t)segLen * 16;
  uint8_t* HStore = (uint8_t*)xcalloc(vb, 1);
  uint8_t* HLoad = (uint8_t*)xcalloc(vb, 1);
  uint8_t* E = (uint8_t*)xcalloc(vb, 1);
  uint8_t* Hmax = (uint8_t*)xcalloc(vb, 1);
  uint8_t vMaxScore[16] = { 0 }, vMaxMark[16] = { 0 };
  int32_t begin = 0, end = refLen, step = 1;
  if (ref_dir == 1) { begin = refLen - 1; end = -1; step = -1; }
  for (int32_t i = begin; i != end; i += step) {
    uint8_t e[16], vF[16] = { 0 }, vMaxColumn[16] = { 0 }, vH[16];
    /* vH = pvHStore[segLen-1] << 1 byte */
    vH[0] = 0;
    for (int k = 1; k < 16; k++) vH[k] = HStore[(size_t)(segLen - 1) * 16 + k - 1];
    const uint8_t* vP = prof + (size_t)ref[i] * segLen * 16;
    uint8_t* pv = HLoad; HLoad = HStore; HStore = pv;
    for (int32_t j = 0; j < segLen; j++) {
      for (int k = 0; k < 16; k++) {
        uint8_t h = adds_u8(vH[k], vP[(size_t)j * 16 + k]);
        h = subs_u8(h, bias);
        e[k] = E[(size_t)j * 16 + k];
        h = max_u8(h, e[k]);
        h = max_u8(h, vF[k]);
        vMaxColumn[k] = max_u8(vMaxColumn[k], h);
        HStore[(size_t)j * 16 + k] = h;
        h = subs_u8(h, gapO);
        e[k] = subs_u8(e[k], gapE);
        e[k] = max_u8(e[k], h);
        E[(size_t)j * 16 + k] = e[k];
        vF[k] = subs_u8(vF[k], gapE);
        vF[k] = max_u8(vF[k], h);
        vH[k] = HLoad[(size_t)j * 16 + k];
      }
    }
    /* Lazy-F loop ssw.c:267-299 (E is deliberately not updated) */
    {
      int32_t j = 0;
      for (int k = 15; k >= 1; k--) vF[k] = vF[k - 1];
      vF[0] = 0;
      for (;;) {
        int all = 1;
        for (int k = 0; k < 16; k++) {
          uint8_t h = HStore[(size_t)j * 16 + k];
          if (subs_u8(vF[k], subs_u8(h, gapO)) != 0) { all = 0; break; }
        }
        if (all) break;
        for (int k = 0; k < 16; k++) {
          uint8_t h = max_u8(HStore[(size_t)j * 16 + k], vF[k]);
          vMaxColumn[k] = max_u8(vMaxColumn[k], h);
          HStore[(size_t)j * 16 + k] = h;
          vF[k] = subs_u8(vF[k], gapE);
        }
        j++;
        if (j >= segLen) { j = 0; for (int k = 15; k >= 1; k--) vF[k] = vF[k - 1]; vF[0] = 0; }
      }
    }
    int changed = 0;
    for (int k = 0; k < 16; k++) { vMaxScore[k] = max_u8(vMaxScore[k], vMaxColumn[k]); if (vMaxScore[k] != vMaxMark[k]) changed = 1; }
    if (changed) {
      uint8_t temp = 0;
      for (int k = 0; k < 16; k++) { vMaxMark[k] = vMaxScore[k]; temp = max_u8(temp, vMaxScore[k]); }
      if (temp > max) {
        max = temp;
        if (max + bias >= 255) break;
        end_ref = i;
        memcpy(Hmax, HStore, vb);
      }
    }
    uint8_t mc = 0;
    for (int k = 0; k < 16; k++) mc = max_u8(mc, vMaxColumn[k]);
    if (mc == terminate) break;
  }
  int32_t column_len = segLen * 16;
  for (int32_t i = 0; i < column_len; i++) {
    if (Hmax[i] == max) { int32_t temp = i / 16 + i % 16 * segLen; if (temp < end_read) end_read = temp; }
  }
  free(Hmax); free(E); free(HLoad); free(HStore);
  aln_end b; b.score = (uint16_t)(max + bias >= 255 ? 255 : max); b.ref = end_ref; b.read = end_read;
  return b;
}

/* ssw.c:375-397 qP_word */
static int16_t* qp_word(const int8_t* read, const int8_t* mat, int32_t readLen, int32_t n) {
  int32_t segLen = (readLen + 7) / 8;
  int16_t* t = (int16_t*)xcalloc((size_t)n * segLen * 8, 2);
  int16_t* p = t;
  for (int32_t nt = 0; nt < n; nt++)
    for (int32_t i = 0; i < segLen; i++) {
      int32_t j = i;
      for (int32_t seg = 0; seg < 8; seg++) { *p++ = j >= readLen ? 0 : mat[nt * n + read[j]]; j += segLen; }
    }
  return t;
}
static inline int16_t adds_i16(int16_t a, int16_t b) { int s = (int)a + b; return (int16_t)(s > 32767 ? 32767 : (s < -32768 ? -32768 : s)); }
static inline int16_t subs_u16(int16_t a, int16_t b) { uint16_t x = (uint16_t)a, y = (uint16_t)b; return (int16_t)(x > y ? x - y : 0); }
static inline int16_t max_i16(int16_t a, int16_t b) { return a > b ? a : b; }

/* ssw.c:399-575 sw_sse2_word */
static aln_end sw_word(const int8_t* ref, int ref_dir, int32_t refLen, int32_t readLen, uint8_t gapO8, uint8_t gapE8,
                       const int16_t* prof, uint16_t terminate) {
  uint16_t max = 0;
  int32_t end_read = readLen - 1, end_ref = 0;
  int32_t segLen = (readLen + 7) / 8;
  size_t vb = (size_t)segLen * 8;
  int16_t* HStore = (int16_t*)xcalloc(vb, 2);
  int16_t* HLoad = (int16_t*)xcalloc(vb, 2);
  int16_t* E = (int16_t*)xcalloc(vb, 2);
  int16_t* Hmax = (int16_t*)xcalloc(vb, 2);
  int16_t gapO = gapO8, gapE = gapE8;
  int16_t vMaxScore[8] = { 0 }, vMaxMark[8] = { 0 };
  int32_t begin = 0, end = refLen, step = 1;
  if (ref_dir == 1) { begin = refLen - 1; end = -1; step = -1; }
  for (int32_t i = begin; i != end; i += step) {
    int16_t e[8], vF[8] = { 0 }, vMaxColumn[8] = { 0 }, vH[8];
    vH[0] = 0;
    for (int k = 1; k < 8; k++) vH[k] = HStore[(size_t)(segLen - 1) * 8 + k - 1];
    int16_t* pv = HLoad;
    const int16_t* vP = prof + (size_t)ref[i] * segLen * 8;
    HLoad = HStore; HStore = pv;
    for (int32_t j = 0; j < segLen; j++) {
      for (int k = 0; k < 8; k++) {
        int16_t h = adds_i16(vH[k], vP[(size_t)j * 8 + k]);
        e[k] = E[(size_t)j * 8 + k];
        h = max_i16(h, e[k]);
        h = max_i16(h, vF[k]);
        vMaxColumn[k] = max_i16(vMaxColumn[k], h);
        HStore[(size_t)j * 8 + k] = h;
        h = subs_u16(h, gapO);
        e[k] = subs_u16(e[k], gapE);
        e[k] = max_i16(e[k], h);
        E[(size_t)j * 8 + k] = e[k];
        vF[k] = subs_u16(vF[k], gapE);
        vF[k] = max_i16(vF[k], h);
        vH[k] = HLoad[(size_t)j * 8 + k];
      }
    }
    /* Lazy-F ssw.c:496-507 */
    {
      int done = 0;
      for (int kk = 0; kk < 8 && !done; kk++) {
        for (int k = 7; k >= 1; k--) vF[k] = vF[k - 1];
        vF[0] = 0;
        for (int32_t j = 0; j < segLen; j++) {
          int any = 0;
          for (int k = 0; k < 8; k++) {
            int16_t h = max_i16(HStore[(size_t)j * 8 + k], vF[k]);
            HStore[(size_t)j * 8 + k] = h;
            h = subs_u16(h, gapO);
            vF[k] = subs_u16(vF[k], gapE);
            if (vF[k] > h) any = 1;
          }
          if (!any) { done = 1; break; }
        }
      }
    }
    int changed = 0;
    for (int k = 0; k < 8; k++) { vMaxScore[k] = max_i16(vMaxScore[k], vMaxColumn[k]); if (vMaxScore[k] != vMaxMark[k]) changed = 1; }
    if (changed) {
      int16_t temp16 = vMaxScore[0];
      for (int k = 0; k < 8; k++) { vMaxMark[k] = vMaxScore[k]; temp16 = max_i16(temp16, vMaxScore[k]); }
      uint16_t temp = (uint16_t)temp16;
      if (temp > max) { max = temp; end_ref = i; memcpy(Hmax, HStore, vb * 2); }
    }
    int16_t mc = vMaxColumn[0];
    for (int k = 0; k < 8; k++) mc = max_i16(mc, vMaxColumn[k]);
    if ((uint16_t)mc == terminate) break;
  }
  int32_t column_len = segLen * 8;
  for (int32_t i = 0; i < column_len; i++) {
    if ((uint16_t)Hmax[i] == max) { int32_t temp = i / 8 + i % 8 * segLen; if (temp < end_read) end_read = temp; }
  }
  free(Hmax); free(E); free(HLoad); free(HStore);
  aln_end b; b.score = max; b.ref = end_ref; b.read = end_read;
  return b;
}

/* ssw.c:69-72 band coordinate macros */
#define SET_U(u, w, i, j) { int x = (i) - (w); x = x > 0 ? x : 0; (u) = (j) - x + 1; }
#define SET_D(u, w, i, j, p) { int x = (i) - (w); x = x > 0 ? x : 0; x = (j) - x; (u) = x * 3 + (p); }

/* ssw.c:577-773 banded_sw.  Returns cigar length (cigar written BAM-style (len<<4)|op), or -1. */
static int32_t banded_sw(const int8_t* ref, const int8_t* read, int32_t refLen, int32_t readLen, int32_t score,
                         uint32_t gapO, uint32_t gapE, int32_t band_width, const int8_t* mat, int32_t n,
                         uint32_t** cigar_out) {
  int32_t i, j, e, f, temp1, temp2, l, max = 0;
  int32_t width, width_d;
  int32_t *h_b = NULL, *e_b = NULL, *h_c = NULL;
  int8_t *direction = NULL, *direction_line = NULL;
  size_t cap_w = 0, cap_d = 0;
  do {
    width = band_width * 2 + 3; width_d = band_width * 2 + 1;
    if ((size_t)width + 2 > cap_w) {
      size_t nw = (size_t)width + 2;
      h_b = (int32_t*)xrealloc(h_b, nw * 4); e_b = (int32_t*)xrealloc(e_b, nw * 4); h_c = (int32_t*)xrealloc(h_c, nw * 4);
      memset(h_b + cap_w, 0, (nw - cap_w) * 4); memset(e_b + cap_w, 0, (nw - cap_w) * 4); memset(h_c + cap_w, 0, (nw - cap_w) * 4);
      cap_w = nw;
    }
    size_t need_d = (size_t)width_d * readLen * 3 + 8;
    if (need_d > cap_d) { direction = (int8_t*)xrealloc(direction, need_d); memset(direction + cap_d, 0, need_d - cap_d); cap_d = need_d; }
    direction_line = direction;
    for (j = 1; j < width - 1; j++) h_b[j] = 0;
    for (i = 0; i < readLen; i++) {
      int32_t beg = 0, end = refLen - 1, u = 0, edge;
      j = i - band_width; beg = beg > j ? beg : j;
      j = i + band_width; end = end < j ? end : j;
      edge = end + 1 < width - 1 ? end + 1 : width - 1;
      f = h_b[0] = e_b[0] = h_b[edge] = e_b[edge] = h_c[0] = 0;
      direction_line = direction + (size_t)width_d * i * 3;
      for (j = beg; j <= end; j++) {
        int32_t b, e1, f1, d, de, df, dh;
        SET_U(u, band_width, i, j); SET_U(e, band_width, i - 1, j);
        SET_U(b, band_width, i, j - 1); SET_U(d, band_width, i - 1, j - 1);
        SET_D(de, band_width, i, j, 0);
        SET_D(df, band_width, i, j, 1);
        SET_D(dh, band_width, i, j, 2);
        temp1 = i == 0 ? -(int32_t)gapO : h_b[e] - (int32_t)gapO;
        temp2 = i == 0 ? -(int32_t)gapE : e_b[e] - (int32_t)gapE;
        e_b[u] = temp1 > temp2 ? temp1 : temp2;
        direction_line[de] = temp1 > temp2 ? 3 : 2;
        temp1 = h_c[b] - (int32_t)gapO;
        temp2 = f - (int32_t)gapE;
        f = temp1 > temp2 ? temp1 : temp2;
        direction_line[df] = temp1 > temp2 ? 5 : 4;
        e1 = e_b[u] > 0 ? e_b[u] : 0;
        f1 = f > 0 ? f : 0;
        temp1 = e1 > f1 ? e1 : f1;
        temp2 = h_b[d] + mat[ref[j] * n + read[i]];
        h_c[u] = temp1 > temp2 ? temp1 : temp2;
        if (h_c[u] > max) max = h_c[u];
        if (temp1 <= temp2) direction_line[dh] = 1;
        else direction_line[dh] = e1 > f1 ? direction_line[de] : direction_line[df];
      }
      for (j = 1; j <= u; j++) h_b[j] = h_c[j];
    }
    band_width *= 2;
  } while (max < score);
  band_width /= 2;
  /* trace back ssw.c:674-747 */
  size_t ccap = 16; uint32_t* c = (uint32_t*)xcalloc(ccap, 4);
  i = readLen - 1; j = refLen - 1; e = 0; l = 0; f = max = 0; temp2 = 2;
  while (i > 0) {
    SET_D(temp1, band_width, i, j, temp2);
    switch (direction_line[temp1]) {
      case 1: --i; --j; temp2 = 2; direction_line -= width_d * 3; f = 0; break;
      case 2: --i; temp2 = 0; direction_line -= width_d * 3; f = 1; break;
      case 3: --i; temp2 = 2; direction_line -= width_d * 3; f = 1; break;
      case 4: --j; temp2 = 1; f = 2; break;
      case 5: --j; temp2 = 2; f = 2; break;
      default:
        free(c); free(direction); free(h_b); free(e_b); free(h_c);
        return -1;       /* reference prints "Trace back error" and exits */
    }
    if (f == max) ++e;
    else {
      ++l;
      if ((size_t)l + 2 >= ccap) { ccap *= 2; c = (uint32_t*)xrealloc(c, ccap * 4); }
      c[l - 1] = (uint32_t)e << 4 | (uint32_t)max;
      max = f; e = 1;
    }
  }
  if ((size_t)l + 3 >= ccap) { ccap = ccap * 2 + 4; c = (uint32_t*)xrealloc(c, ccap * 4); }
  if (f == 0) { ++l; c[l - 1] = (uint32_t)(e + 1) << 4; }
  else { l += 2; c[l - 2] = (uint32_t)e << 4 | (uint32_t)f; c[l - 1] = 16; }
  uint32_t* c1 = (uint32_t*)xcalloc((size_t)l, 4);
  for (int32_t s = 0; s < l; s++) c1[s] = c[l - 1 - s];
  free(c); free(direction); free(h_b); free(e_b); free(h_c);
  *cigar_out = c1;
  return l;
}

typedef struct {
  uint32_t* cigar; int32_t cigarLen;
  int32_t ref_begin1, ref_end1, read_begin1, read_end1;
  uint16_t score1;
} ssw_res;

/* ssw_init (ssw.c:788-814, score_size=2) + ssw_align (ssw.c:834-941, flag=2, filterd=0, maskLen=0).
 * returns 0 when the reference would return NULL. */
static int ssw_run(const int8_t* read, int32_t readLen, const int8_t* ref, int32_t refLen, const int8_t* mat, int32_t n,
                   uint8_t gapO, uint8_t gapE, uint16_t filters, ssw_res* r, orc_counters* ctr) {
  int32_t bias = 0;
  for (int32_t i = 0; i < n * n; i++) if (mat[i] < bias) bias = mat[i];
  bias = abs(bias);
  uint8_t* pb = qp_byte(read, mat, readLen, n, (uint8_t)bias);
  int16_t* pw = qp_word(read, mat, readLen, n);
  int word = 0;
  r->ref_begin1 = -1; r->read_begin1 = -1; r->cigar = NULL; r->cigarLen = 0;
  if (ctr) ctr->n_sw_fwd++;
  aln_end best = sw_byte(ref, 0, refLen, readLen, gapO, gapE, pb, (uint8_t)-1, (uint8_t)bias);
  if (best.score == 255) { best = sw_word(ref, 0, refLen, readLen, gapO, gapE, pw, (uint16_t)-1); word = 1; }
  free(pb); free(pw);
  r->score1 = best.score; r->ref_end1 = best.ref; r->read_end1 = best.read;
  if (r->score1 < filters) return 1;                              /* flag == 2 && score1 < filters */
  /* reverse pass ssw.c:900-918 */
  if (ctr) ctr->n_sw_rev++;
  int32_t rl = r->read_end1 + 1;
  int8_t* rev = (int8_t*)xcalloc((size_t)rl + 1, 1);
  for (int32_t k = 0; k < rl; k++) rev[k] = read[r->read_end1 - k];
  aln_end br;
  if (!word) {
    uint8_t* vp = qp_byte(rev, mat, rl, n, (uint8_t)bias);
    br = sw_byte(ref, 1, r->ref_end1 + 1, rl, gapO, gapE, vp, (uint8_t)r->score1, (uint8_t)bias);
    free(vp);
  } else {
    int16_t* vp = qp_word(rev, mat, rl, n);
    br = sw_word(ref, 1, r->ref_end1 + 1, rl, gapO, gapE, vp, r->score1);
    free(vp);
  }
  free(rev);
  r->ref_begin1 = br.ref;
  r->read_begin1 = r->read_end1 - br.read;
  /* (2 & flag) != 0 && score1 < filters already handled; cigar ssw.c:921-936 */
  int32_t rfl = r->ref_end1 - r->ref_begin1 + 1;
  int32_t rdl = r->read_end1 - r->read_begin1 + 1;
  int32_t band = abs(rfl - rdl) + 1;
  if (ctr) ctr->n_traceback++;
  uint32_t* cg = NULL;
  int32_t cl = banded_sw(ref + r->ref_begin1, read + r->read_begin1, rfl, rdl, r->score1, gapO, gapE, band, mat, n, &cg);
  if (cl < 0) return 0;
  r->cigar = cg; r->cigarLen = cl;
  return 1;
}

int orc_ssw(const int8_t* read, int32_t readLen, const int8_t* ref, int32_t refLen, const int8_t* mat5x5,
            uint8_t gap_open, uint8_t gap_ext, uint16_t filters, orc_ssw_result* out) {
  ssw_res r;
  int ok = ssw_run(read, readLen, ref, refLen, mat5x5, 5, gap_open, gap_ext, filters, &r, NULL);
  memset(out, 0, sizeof *out);
  if (!ok) return 0;
  out->score1 = r.score1; out->ref_begin1 = r.ref_begin1; out->ref_end1 = r.ref_end1;
  out->read_begin1 = r.read_begin1; out->read_end1 = r.read_end1;
  out->cigar_len = (uint32_t)r.cigarLen;
  for (int32_t i = 0; i < r.cigarLen && i < 4096; i++) out->cigar[i] = r.cigar[i];
  free(r.cigar);
  return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* Per-read state: read.hpp:80-173, ssw.hpp:44-140                                             */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  uint32_t* cigar; uint32_t cigar_len;
  uint32_t ref_num; int32_t ref_begin1, ref_end1, read_begin1, read_end1; uint32_t readlen;
  uint16_t score1, part, index_num; uint8_t strand;
} orc_align;

typedef struct {
  uint32_t lastIndex, lastPart;
  uint8_t is_done, is_hit;
  uint16_t max_SW_count; int32_t num_alignments; uint32_t hit_seeds;
  uint32_t min_index, max_index;
  orc_align* alignv; uint32_t n_align, cap_align;
} orc_state;

struct orc_batch { uint32_t n; orc_state* st; };

static void align_free(orc_align* a) { free(a->cigar); a->cigar = NULL; }
static orc_align align_clone(const orc_align* a) {
  orc_align b = *a;
  b.cigar = (uint32_t*)xcalloc(a->cigar_len, 4);
  memcpy(b.cigar, a->cigar, (size_t)a->cigar_len * 4);
  return b;
}
static void state_clear(orc_state* s) {
  for (uint32_t i = 0; i < s->n_align; i++) align_free(&s->alignv[i]);
  free(s->alignv);
  memset(s, 0, sizeof *s);
}
static void state_copy(orc_state* dst, const orc_state* src) {
  state_clear(dst);
  *dst = *src;
  dst->alignv = (orc_align*)xcalloc(src->n_align ? src->n_align : 1, sizeof(orc_align));
  dst->cap_align = src->n_align ? src->n_align : 1;
  for (uint32_t i = 0; i < src->n_align; i++) dst->alignv[i] = align_clone(&src->alignv[i]);
}
static void state_push(orc_state* s, const orc_align* a) {
  if (s->n_align == s->cap_align) { s->cap_align = s->cap_align ? s->cap_align * 2 : 2; s->alignv = (orc_align*)xrealloc(s->alignv, s->cap_align * sizeof(orc_align)); }
  s->alignv[s->n_align++] = align_clone(a);
}

orc_batch* orc_batch_new(uint32_t n) {
  orc_batch* b = (orc_batch*)xcalloc(1, sizeof *b);
  b->n = n; b->st = (orc_state*)xcalloc(n, sizeof(orc_state));
  return b;
}
void orc_batch_free(orc_batch* b) { if (!b) return; for (uint32_t i = 0; i < b->n; i++) state_clear(&b->st[i]); free(b->st); free(b); }
int orc_batch_is_hit(const orc_batch* b, uint32_t i) { return b->st[i].is_hit; }

/* Read::toBinString read.cpp:429-462 + alignment_struct2::toString :74-95 + s_align2::toString ssw.hpp:106-140 */
size_t orc_batch_record(const orc_batch* b, uint32_t i, uint8_t* buf, size_t cap) {
  const orc_state* s = &b->st[i];
  if (s->n_align == 0) return 0;
  size_t need = 4 * 6 + 3 + 2 + 4 + 4 + 8 + 4 + 4 + 8;
  for (uint32_t k = 0; k < s->n_align; k++) need += 8 + 8 + (size_t)s->alignv[k].cigar_len * 4 + 4 * 6 + 2 * 3 + 1;
  if (!buf || cap < need) return need;
  uint8_t* p = buf;
#define PUT(v) do { memcpy(p, &(v), sizeof(v)); p += sizeof(v); } while (0)
  uint32_t z32 = 0; uint8_t z8 = 0;
  PUT(s->lastIndex); PUT(s->lastPart); PUT(z32); PUT(z32); PUT(z32); PUT(z32);
  PUT(s->is_done); PUT(s->is_hit); PUT(z8);
  PUT(s->max_SW_count); PUT(s->num_alignments); PUT(s->hit_seeds);
  uint64_t asz = 4 + 4 + 8;
  for (uint32_t k = 0; k < s->n_align; k++) asz += 8 + 8 + (uint64_t)s->alignv[k].cigar_len * 4 + 4 * 6 + 2 * 3 + 1;
  PUT(asz);
  PUT(s->min_index); PUT(s->max_index);
  uint64_t nal = s->n_align; PUT(nal);
  for (uint32_t k = 0; k < s->n_align; k++) {
    const orc_align* a = &s->alignv[k];
    uint64_t rl = 8 + (uint64_t)a->cigar_len * 4 + 4 * 6 + 2 * 3 + 1; PUT(rl);
    uint64_t cl = a->cigar_len; PUT(cl);
    memcpy(p, a->cigar, (size_t)a->cigar_len * 4); p += (size_t)a->cigar_len * 4;
    PUT(a->ref_num); PUT(a->ref_begin1); PUT(a->ref_end1); PUT(a->read_begin1); PUT(a->read_end1); PUT(a->readlen);
    PUT(a->score1); PUT(a->part); PUT(a->index_num); PUT(a->strand);
  }
#undef PUT
  return (size_t)(p - buf);
}

/* working copy of one read while it is processed against one index part */
typedef struct {
  uint8_t* iseq; uint32_t len;
  int is03, is04, reversed;
  uint32_t* amb; uint32_t n_amb;
  int8_t mat[25];
  int32_t best;
  int is_new_hit;
  hitvec hits;           /* read.id_win_hits */
  orc_state st;          /* fields that round-trip through the KVDB */
} wread;

/* read.cpp:379-401 */
static void flip34(wread* r) {
  if (r->n_amb > 0) {
    uint8_t val = r->is03 ? 4 : 0;
    if (r->reversed) for (uint32_t p = 0; p < r->n_amb; p++) r->iseq[(r->len - r->amb[p]) - 1] = val;
    else for (uint32_t p = 0; p < r->n_amb; p++) r->iseq[r->amb[p]] = val;
    r->is03 = !r->is03; r->is04 = !r->is04;
  }
}
/* read.cpp:350-357 */
static void rev_int_str(wread* r) {
  static const uint8_t comp[5] = { 3, 2, 1, 0, 4 };
  for (uint32_t i = 0, j = r->len; i < j--; i++) { uint8_t t = r->iseq[i]; r->iseq[i] = r->iseq[j]; r->iseq[j] = t; }
  for (uint32_t i = 0; i < r->len; i++) r->iseq[i] = comp[r->iseq[i]];
  r->reversed = !r->reversed;
}

/* find_lis alignment.cpp:58-98 over a[lo..lo+n) (the deque), comparing .second (read pos). */
typedef struct { uint32_t first, second; } u32pair;
static uint32_t find_lis(const u32pair* a, uint32_t n, uint32_t* b, uint32_t* p) {
  uint32_t nb = 0;
  if (n == 0) return 0;
  memset(p, 0, (size_t)n * 4);            /* the reference's p is a fresh zeroed vector per call */
  b[nb++] = 0;
  for (uint32_t i = 1; i < n; i++) {
    if (a[b[nb - 1]].second < a[i].second) { p[i] = b[nb - 1]; b[nb++] = i; continue; }
    uint32_t u = 0, v = nb - 1;
    while (u < v) { uint32_t c = (u + v) / 2; if (a[b[c]].second < a[i].second) u = c + 1; else v = c; }
    if (a[i].second < a[b[u]].second) { if (u > 0) p[i] = b[u - 1]; b[u] = i; }
  }
  for (uint32_t u = nb, v = b[nb - 1]; u--; v = p[v]) b[u] = v;
  return nb;
}

static int cmp_u32(const void* x, const void* y) { uint32_t a = *(const uint32_t*)x, b = *(const uint32_t*)y; return a < b ? -1 : a > b; }
static int cmp_cand(const void* x, const void* y) {   /* count desc, ref asc: alignment.cpp:143-148 */
  const u32pair* a = (const u32pair*)x; const u32pair* b = (const u32pair*)y;
  if (a->second == b->second) return a->first < b->first ? -1 : a->first > b->first;
  return a->second > b->second ? -1 : 1;
}
static int cmp_hit(const void* x, const void* y) {    /* ref pos asc, read pos asc: alignment.cpp:197-201 */
  const u32pair* a = (const u32pair*)x; const u32pair* b = (const u32pair*)y;
  if (a->first == b->first) return a->second < b->second ? -1 : a->second > b->second;
  return a->first < b->first ? -1 : 1;
}

static uint32_t find_min_index(const orc_state* s) {  /* alignment.cpp findMinIndex */
  uint32_t mi = 0;
  for (uint32_t i = 1; i < s->n_align; i++) if (s->alignv[i].score1 < s->alignv[mi].score1) mi = i;
  return mi;
}
static uint32_t find_max_index(const orc_state* s) {
  uint32_t mi = 0;
  for (uint32_t i = 1; i < s->n_align; i++) if (s->alignv[i].score1 > s->alignv[mi].score1) mi = i;
  return mi;
}

/* alignment.cpp:100-509 */
static void compute_lis_alignment(wread* read, const orc_params* o, const orc_index* ix, const orc_refs* refs,
                                  orc_counters* ctr, int* search, uint32_t max_SW_score) {
  int is_aligned = 0;
  /* 1. per-reference seed-hit histogram (:117-130): std::map iteration = ascending ref number */
  size_t tot = 0;
  for (uint32_t h = 0; h < read->hits.n; h++) tot += ix->positions[read->hits.v[h].id].size;
  uint32_t* seqs = (uint32_t*)xcalloc(tot, 4);
  size_t t = 0;
  for (uint32_t h = 0; h < read->hits.n; h++) {
    const orc_origin* po = &ix->positions[read->hits.v[h].id];
    for (uint32_t j = 0; j < po->size; j++) seqs[t++] = po->arr[2 * j + 1];
  }
  qsort(seqs, tot, 4, cmp_u32);
  u32pair* cand = (u32pair*)xcalloc(tot, sizeof(u32pair)); uint32_t ncand = 0;
  for (size_t i = 0; i < tot;) {
    size_t j = i; while (j < tot && seqs[j] == seqs[i]) j++;
    if ((uint32_t)(j - i) >= (uint32_t)o->num_seeds) { cand[ncand].first = seqs[i]; cand[ncand].second = (uint32_t)(j - i); ncand++; }
    i = j;
  }
  free(seqs);
  qsort(cand, ncand, sizeof(u32pair), cmp_cand);

  u32pair* hits_on_ref = (u32pair*)xcalloc(tot, sizeof(u32pair));
  uint32_t* lis_b = (uint32_t*)xcalloc(tot + 1, 4);
  uint32_t* lis_p = (uint32_t*)xcalloc(tot + 1, 4);
  int is_search_candidates = 1;
  for (uint32_t k = 0; k < ncand && is_search_candidates; k++) {
    uint32_t max_ref = cand[k].first, max_occur = cand[k].second;
    if (max_occur < (uint32_t)o->num_seeds) break;
    if (is_aligned && o->min_lis > 0 && k > 0 && max_occur < cand[k - 1].second) {   /* :165-169 */
      --read->best;
      if (read->best < 1) break;
    }
    /* 3. hits on this reference (:181-201) */
    uint32_t nh = 0;
    for (uint32_t h = 0; h < read->hits.n; h++) {
      const orc_origin* po = &ix->positions[read->hits.v[h].id];
      for (uint32_t j = 0; j < po->size; j++)
        if (po->arr[2 * j + 1] == max_ref) { hits_on_ref[nh].first = po->arr[2 * j]; hits_on_ref[nh].second = read->hits.v[h].win; nh++; }
    }
    qsort(hits_on_ref, nh, sizeof(u32pair), cmp_hit);
    /* 4. sliding window (:203-506); deque match_set = hits_on_ref[ms_lo, ms_hi) */
    uint32_t it = 0, ms_lo = 0, ms_hi = 0;
    uint32_t begin_ref = hits_on_ref[0].first, begin_read = hits_on_ref[0].second;
    while (it != nh && is_search_candidates) {
      size_t end_ref_max = (size_t)begin_ref + read->len - begin_read - o->lnwin + 1;       /* :231 */
      int push = 0;
      while (it != nh && hits_on_ref[it].first <= end_ref_max) { ms_hi = ++it; push = 1; }
      int skip_to_pop = 0;
      if (!push && is_aligned) skip_to_pop = 1;                                             /* heuristic 1 :243-246 */
      else is_aligned = 0;
      if (!skip_to_pop && (ms_hi - ms_lo) >= (uint32_t)o->num_seeds) {
        uint32_t nl = find_lis(hits_on_ref + ms_lo, ms_hi - ms_lo, lis_b, lis_p);
        if (nl >= (uint32_t)o->min_lis) {
          uint32_t lcs_ref_start = hits_on_ref[ms_lo + lis_b[0]].first;
          uint32_t lcs_que_start = hits_on_ref[ms_lo + lis_b[0]].second;
          size_t head = 0, tail = 0, align_ref_start = 0, align_que_start = 0, align_length = 0;
          size_t reflen = refs->len[max_ref];
          size_t rlen = read->len;
          uint32_t edges;
          if (o->is_as_percent) edges = (uint32_t)((o->edges / 100.0) * rlen);
          else edges = (uint32_t)o->edges;
          if (lcs_ref_start < lcs_que_start) {                                              /* :287-325 */
            align_ref_start = 0;
            align_que_start = lcs_que_start - lcs_ref_start;
            head = 0;
            if (reflen < rlen) {
              tail = 0;
              if (align_que_start > (rlen - reflen)) align_length = reflen - (align_que_start - (rlen - reflen));
              else align_length = reflen;
            } else {
              tail = reflen - align_ref_start - rlen;
              if (tail > (uint32_t)(edges - 1)) tail = edges;
              align_length = rlen + head + tail - align_que_start;
            }
          } else {                                                                          /* :326-357 */
            align_ref_start = lcs_ref_start - lcs_que_start;
            align_que_start = 0;
            if (align_ref_start > (uint32_t)(edges - 1)) head = edges;
            if (align_ref_start + rlen > reflen) {
              tail = 0;
              align_length = reflen - align_ref_start - head;
            } else {
              tail = reflen - align_ref_start - rlen;
              if (tail > (uint32_t)(edges - 1)) tail = edges;
              align_length = rlen + head + tail;
            }
          }
          if (read->is03) flip34(read);                                                     /* :360-361 */
          ssw_res res;
          /* With large -edges a read that hangs off the end of its reference can leave align_length - head - tail (size_t in the reference,
           * :341-343) at or below zero: the reference then hands ssw_align a wrapped length -- undefined, it crashes or reads out of bounds.
           * The oracle defines the case the way libsmr_hip does (smr_walk.hpp task_ok): no Smith-Waterman call, the candidate does not align. */
          int ok = 0;
          if ((int64_t)align_length - (int64_t)head - (int64_t)tail > 0 && (int64_t)align_length > 0)
            ok = ssw_run((const int8_t*)read->iseq + align_que_start, (int32_t)(align_length - head - tail),
                         (const int8_t*)refs->seq[max_ref] + align_ref_start - head, (int32_t)align_length,
                         read->mat, 5, (uint8_t)o->gap_open, (uint8_t)o->gap_ext, (uint16_t)o->minimal_score, &res, ctr);
          is_aligned = (ok && res.score1 > o->minimal_score);                               /* :388 */
          if (is_aligned) {
            if (res.score1 == max_SW_score) ++read->st.max_SW_count;
            orc_align al;
            al.cigar = res.cigar; al.cigar_len = (uint32_t)res.cigarLen;
            al.ref_begin1 = res.ref_begin1 + (int32_t)(align_ref_start - head);
            al.ref_end1 = res.ref_end1 + (int32_t)(align_ref_start - head);
            al.read_begin1 = res.read_begin1 + (int32_t)align_que_start;
            al.read_end1 = res.read_end1 + (int32_t)align_que_start;
            al.readlen = (uint32_t)rlen; al.ref_num = max_ref;
            al.index_num = (uint16_t)o->index_num; al.part = (uint16_t)o->part;
            al.strand = (uint8_t)!read->reversed; al.score1 = res.score1;
            if (!read->st.is_hit) {                                                         /* :411-416 */
              read->st.is_hit = 1;
              ctr->num_aligned++;
              ctr->reads_matched_per_db[o->index_num]++;
            }
            if (o->num_alignments == 0 || !o->is_best || (o->is_best && read->st.n_align < o->num_alignments)) {
              state_push(&read->st, &al);
              read->is_new_hit = 1;
            } else if (o->is_best && read->st.n_align == o->num_alignments &&
                       read->st.alignv[read->st.min_index].score1 < res.score1) {           /* :425-459 */
              if (o->num_alignments > 1 && read->st.max_index == 0 && read->st.min_index == 0) {
                read->st.min_index = find_min_index(&read->st);
                read->st.max_index = find_max_index(&read->st);
              }
              uint32_t mn = read->st.min_index, mx = read->st.max_index;
              uint16_t old_index_num = read->st.alignv[mn].index_num;
              (void)old_index_num;
              align_free(&read->st.alignv[mn]);
              read->st.alignv[mn] = align_clone(&al);
              read->is_new_hit = 1;
              if (res.score1 > read->st.alignv[mx].score1 && read->st.n_align > 1) {
                read->st.max_index = mn;
                read->st.min_index = find_min_index(&read->st);
              }
              /* :454-457 -- note: the reference reads index_num AFTER the replacement */
              --ctr->reads_matched_per_db[read->st.alignv[mn].index_num];
              ++ctr->reads_matched_per_db[o->index_num];
            }
            if (o->num_alignments > 0) {                                                    /* :462-469 */
              if (o->is_best) { if (o->num_alignments == read->st.max_SW_count) is_search_candidates = 0; }
              else if (o->num_alignments == read->st.n_align) is_search_candidates = 0;
            }
            *search = 0;
          }
          if (ok) free(res.cigar);
        }
      }
      /* pop: (:486-506) */
      if (ms_hi > ms_lo) ms_lo++;
      if (ms_hi == ms_lo) {
        if (it != nh) { begin_ref = hits_on_ref[it].first; begin_read = hits_on_ref[it].second; }
        else break;
      } else { begin_ref = hits_on_ref[ms_lo].first; begin_read = hits_on_ref[ms_lo].second; }
    }
  }
  free(lis_b); free(lis_p); free(hits_on_ref); free(cand);
}

/* paralleltraversal.cpp:81-298 */
static void traverse(wread* read, const orc_params* o, const orc_index* ix, const orc_refs* refs,
                     orc_counters* ctr, int isLastStrand) {
  read->st.lastIndex = o->index_num;
  read->st.lastPart = o->part;
  uint32_t win_shift = o->skiplengths[0];
  uint8_t* searched = (uint8_t*)xcalloc(read->len, 1);
  size_t pass_n = 0;
  uint32_t max_SW_score = read->len * (uint32_t)o->match;
  for (int search = 1; search;) {
    uint32_t numwin = (read->len - o->lnwin + win_shift) / win_shift;
    uint32_t win_pos = 0;
    for (uint32_t win_num = 0; win_num < numwin; ++win_num) {
      if (read->is04) flip34(read);
      if (!searched[win_pos]) {
        searched[win_pos] = 1;
        hitvec id_hits = { 0, 0, 0 }; int accept_zero = 0;
        if (ctr) ctr->n_windows++;
        window_search(ix, read->iseq, win_pos, o->lnwin, o->minoccur, o->is_full_search, &id_hits, &accept_zero, ctr);
        if (id_hits.n) {
          for (uint32_t h = 0; h < id_hits.n; h++) hv_push(&read->hits, id_hits.v[h].id, id_hits.v[h].win);
          if (ctr) ctr->n_hit += id_hits.n;
          ++read->st.hit_seeds;
        }
        free(id_hits.v);
      }
      if (win_num == numwin - 1) {
        if (read->st.hit_seeds >= (uint32_t)o->num_seeds)
          compute_lis_alignment(read, o, ix, refs, ctr, &search, max_SW_score);
        if (search) {
          if (pass_n == 2) search = 0;
          else {
            while (pass_n < 2 && o->skiplengths[pass_n] == o->skiplengths[pass_n + 1]) ++pass_n;
            if (++pass_n > 2) search = 0;
            else win_shift = o->skiplengths[pass_n];
          }
        }
        break;
      }
      win_pos += win_shift;
    }
  }
  free(searched);
  if (o->num_alignments > 0) {                                                              /* :286-291 */
    if ((o->is_best && o->num_alignments == read->st.max_SW_count) ||
        (!o->is_best && read->st.n_align == o->num_alignments))
      read->st.is_done = 1;
  } else {
    if (o->is_last_index_part && isLastStrand && read->st.n_align > 0) read->st.is_done = 1;
  }
}

/* processor.cpp:104-161 for a batch */
void orc_align_part(const orc_index* ix, const orc_refs* refs, const orc_params* o,
                    const char* seqs, const uint64_t* offs, uint32_t n_reads,
                    orc_batch* batch, orc_counters* ctr) {
  for (uint32_t ri = 0; ri < n_reads; ri++) {
    const char* s = seqs + offs[ri];
    uint32_t len = (uint32_t)(offs[ri + 1] - offs[ri]);
    if (len < o->lnwin) { ctr->num_short++; continue; }               /* processor.cpp:109-114 */
    orc_state* saved = &batch->st[ri];
    if (saved->is_done) continue;                                    /* :120-126 */
    wread r; memset(&r, 0, sizeof r);
    r.len = len;
    r.iseq = (uint8_t*)xcalloc(len + 1, 1);
    r.amb = (uint32_t*)xcalloc(len + 1, 4);
    for (uint32_t i = 0; i < len; i++) {                             /* seqToIntStr read.cpp:334-347 */
      int c = nt_code((unsigned char)s[i]);
      if (c == 4) { r.amb[r.n_amb++] = i; c = 0; }
      r.iseq[i] = (uint8_t)c;
    }
    r.is03 = 1;
    for (int l = 0, q = 0; l < 4; l++) {                             /* initScoringMatrix read.cpp:274-288 */
      for (int m = 0; m < 4; m++) r.mat[q++] = (int8_t)(l == m ? o->match : o->mismatch);
      r.mat[q++] = (int8_t)o->score_N;
      if (l == 3) for (int m = 0; m < 5; m++) r.mat[q++] = (int8_t)o->score_N;
    }
    state_copy(&r.st, saved);                                        /* load_db read.cpp:467-539 */
    if (o->num_alignments > 0) r.st.num_alignments = (int32_t)o->num_alignments;   /* init read.cpp:264-271 */
    if (o->min_lis > 0) r.best = o->min_lis;
    int single = (o->is_forward != 0) ^ (o->is_reverse != 0);
    int num_strands = single ? 1 : 2;
    for (int count = 0; count < num_strands && !r.st.is_done; ++count) {
      if ((single && o->is_reverse) || count == 1) { if (!r.reversed) rev_int_str(&r); }
      traverse(&r, o, ix, refs, ctr, single || count == 1);
      r.hits.n = 0;
    }
    if (r.is_new_hit) state_copy(saved, &r.st);                      /* kvdb.put :150-155 */
    state_clear(&r.st);
    free(r.hits.v); free(r.iseq); free(r.amb);
  }
}
