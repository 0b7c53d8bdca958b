// smr_reads.cpp -- host side of the read batch: FASTA/FASTQ record reader and the 2-bit packer.
//
// Replaces, for the hot path, Readfeed::next() -> Read(readstr) -> Read::init()
// (/root/reference/src/sortmerna/readfeed.cpp:776-873, read.cpp:264-347): the sequence line is mapped
// with nt_table (include/common.hpp:68-77); letters outside ACGTU become 0 and their position is kept in
// a bit mask (Read::ambiguous_nt).  Unlike the reference's INDEXED feed we read multi-line FASTA records
// completely and do not drop an unterminated last line.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "smr_host.hpp"

namespace {
inline int code_of(unsigned char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
  }
}
void append_read(smr_reads& r, const char* s, size_t len) {
  size_t cw = (len + 15) / 16, mw = (len + 31) / 32;
  size_t base = r.words.size();
  r.words.resize(base + cw + mw, 0);
  uint32_t* cp = r.words.data() + base;
  uint32_t* mp = cp + cw;
  for (size_t k = 0; k < len; k++) {
    int c = code_of((unsigned char)s[k]);
    if (c == 4) { mp[k >> 5] |= 1u << (k & 31); c = 0; }
    cp[k >> 4] |= (uint32_t)c << ((k & 15) * 2);
  }
  r.len.push_back((uint32_t)len);
  r.rec_off.push_back(r.words.size());
  r.total_len += len;
  if (r.n == 0) { r.min_len = r.max_len = (uint32_t)len; }
  else { r.min_len = std::min<uint32_t>(r.min_len, (uint32_t)len); r.max_len = std::max<uint32_t>(r.max_len, (uint32_t)len); }
  r.n++;
}
}  // namespace

extern "C" int smr_reads_pack(const char* seqs, const uint64_t* offs, uint32_t n_reads, smr_reads** out) {
  if ((!seqs && n_reads) || !offs || !out) return SMR_ERR_ARG;
  auto r = new smr_reads();
  r->rec_off.push_back(0);
  for (uint32_t i = 0; i < n_reads; i++) append_read(*r, seqs + offs[i], (size_t)(offs[i + 1] - offs[i]));
  *out = r;
  return SMR_OK;
}

extern "C" int smr_reads_load_fastx(const char* path, uint64_t first, uint64_t count, smr_reads** out, char* err, size_t errcap) {
  if (!path || !out) return SMR_ERR_ARG;
  FILE* f = fopen(path, "rb");
  if (!f) { if (err && errcap) snprintf(err, errcap, "cannot open %s", path); return SMR_ERR_IO; }
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<char> b((size_t)sz + 1);
  if (sz > 0 && fread(b.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); if (err && errcap) snprintf(err, errcap, "short read on %s", path); return SMR_ERR_IO; }
  fclose(f);
  auto r = new smr_reads();
  r->rec_off.push_back(0);
  size_t n = (size_t)sz, o = 0;
  uint64_t rec = 0;
  std::string seq;
  auto line_end = [&](size_t p) { while (p < n && b[p] != '\n') p++; return p; };
  auto want = [&](uint64_t k) { return k >= first && (count == 0 || k < first + count); };
  while (o < n) {
    if (b[o] == '\n' || b[o] == '\r') { o++; continue; }
    if (b[o] == '>') {
      o = line_end(o) + 1;
      seq.clear();
      while (o < n && b[o] != '>') {
        size_t e = line_end(o), le = e;
        while (le > o && (b[le - 1] == '\r' || b[le - 1] == ' ' || b[le - 1] == '\t')) le--;
        seq.append(b.data() + o, le - o);
        o = e + 1;
      }
      if (want(rec)) append_read(*r, seq.data(), seq.size());
      rec++;
    } else if (b[o] == '@') {
      o = line_end(o) + 1;
      size_t e = line_end(o), le = e;
      while (le > o && (b[le - 1] == '\r' || b[le - 1] == ' ' || b[le - 1] == '\t')) le--;
      if (want(rec)) append_read(*r, b.data() + o, le - o);
      rec++;
      o = e + 1;
      o = line_end(o) + 1;   // '+'
      o = line_end(o) + 1;   // quality
    } else {
      delete r;
      if (err && errcap) snprintf(err, errcap, "%s: unexpected character at byte %zu", path, o);
      return SMR_ERR_IO;
    }
    if (count != 0 && rec >= first + count) break;
  }
  *out = r;
  return SMR_OK;
}

extern "C" void smr_reads_free(smr_reads* r) { delete r; }
extern "C" uint32_t smr_reads_count(const smr_reads* r) { return r ? r->n : 0; }
extern "C" uint64_t smr_reads_total_len(const smr_reads* r) { return r ? r->total_len : 0; }
extern "C" uint32_t smr_reads_min_len(const smr_reads* r) { return r ? r->min_len : 0; }
extern "C" uint32_t smr_reads_max_len(const smr_reads* r) { return r ? r->max_len : 0; }

extern "C" void smr_params_default(smr_params* p) {
  if (!p) return;
  memset(p, 0, sizeof *p);
  p->num_seeds = 2; p->min_lis = 2; p->edges = 4; p->is_as_percent = 0;
  p->match = 2; p->mismatch = -3; p->score_N = -3; p->gap_open = 5; p->gap_ext = 2;
  p->num_alignments = 1; p->is_best = 1; p->is_full_search = 0; p->is_forward = 1; p->is_reverse = 1;
  p->minoccur = 0; p->minimal_score = 0; p->index_num = 0; p->part = 0; p->is_last_index_part = 1;
}
