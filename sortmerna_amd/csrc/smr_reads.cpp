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
#include <thread>
#include <vector>

#include "smr_host.hpp"
#include "smr_hostmem.hpp"

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

// ---- multi-threaded front-end -----------------------------------------------------------------
// Replaces the reader side of Readfeed (readfeed.cpp:776-873 next(), :1170-1400 split()) for the hot path: the reference
// splits the input into per-thread files up front and inflates/parses them one line at a time; here the file is mapped
// (or, for .gz, inflated once: izlib.cpp:95-210), cut into byte ranges that start at record boundaries (FASTA: a line
// starting with '>'; FASTQ: a line starting with '@' whose line after next starts with '+' -- a quality line may start
// with '@', but the line after next of a quality line is a sequence line), and processed in two parallel sweeps:
//   sweep 1  every thread lists its records (offset of the sequence text, length);
//   (prefix sums over the per-thread totals give every record its slot in the final arrays)
//   sweep 2  every thread 2-bit packs its records straight into the final batch.
// Same result as smr_reads_load_fastx(path, 0, 0) (smr_reads_digest equal), for any thread count.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

namespace {
struct Bytes {              // the input text: a private mapping of the file, or the inflated copy of a .gz
  const char* p = nullptr; size_t n = 0; void* map = nullptr; size_t map_n = 0; std::vector<char> own;
  ~Bytes() { if (map) munmap(map, map_n); }
};
bool slurp(const char* path, Bytes& b, std::string& why) {
  int fd = open(path, O_RDONLY);
  if (fd < 0) { why = std::string("cannot open ") + path; return false; }
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); why = std::string("cannot stat ") + path; return false; }
  if (st.st_size == 0) { close(fd); return true; }
  void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { why = std::string("cannot map ") + path; return false; }
  madvise(m, (size_t)st.st_size, MADV_WILLNEED);
  b.map = m; b.map_n = (size_t)st.st_size;
  const unsigned char* u = (const unsigned char*)m;
  if (b.map_n >= 2 && u[0] == 0x1f && u[1] == 0x8b) {   // gzip (possibly several members back to back)
    z_stream z; memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 15 + 16) != Z_OK) { why = "inflateInit2 failed"; return false; }
    b.own.resize(std::max<size_t>(b.map_n * 4, 1u << 16));
    size_t out = 0, fed = 0;
    z.next_in = (Bytef*)u; z.avail_in = 0;
    for (;;) {
      if (z.avail_in == 0 && fed < b.map_n) { const size_t c = std::min<size_t>(b.map_n - fed, 1u << 30); z.next_in = (Bytef*)u + fed; z.avail_in = (uInt)c; fed += c; }
      if (out == b.own.size()) b.own.resize(b.own.size() * 2);
      z.next_out = (Bytef*)b.own.data() + out; z.avail_out = (uInt)std::min<size_t>(b.own.size() - out, 1u << 30);
      const uInt room = z.avail_out;
      const int rc = inflate(&z, Z_NO_FLUSH);
      out += room - z.avail_out;
      if (rc == Z_STREAM_END) {
        if (z.avail_in == 0 && fed == b.map_n) break;
        if (inflateReset(&z) != Z_OK) { inflateEnd(&z); why = "inflateReset failed"; return false; }   // next member
      } else if (rc == Z_BUF_ERROR) {
        if (z.avail_in == 0 && fed == b.map_n && z.avail_out != 0) { inflateEnd(&z); why = std::string(path) + ": truncated gzip stream"; return false; }
      } else if (rc != Z_OK) { inflateEnd(&z); why = std::string(path) + ": corrupt gzip stream"; return false; }
    }
    inflateEnd(&z);
    b.own.resize(out);
    munmap(b.map, b.map_n); b.map = nullptr;
    b.p = b.own.data(); b.n = out;
  } else { b.p = (const char*)m; b.n = b.map_n; }
  return true;
}
inline size_t eol(const char* p, size_t o, size_t n) { if (o >= n) return n; const void* q = memchr(p + o, '\n', n - o); return q ? (size_t)((const char*)q - p) : n; }
inline size_t next_line(const char* p, size_t o, size_t n) { const size_t e = eol(p, o, n); return e < n ? e + 1 : n; }
inline size_t rtrim(const char* p, size_t o, size_t e) { while (e > o && (p[e - 1] == '\r' || p[e - 1] == ' ' || p[e - 1] == '\t')) e--; return e; }
bool is_record_start(const char* p, size_t o, size_t n, bool fastq) {
  if (o >= n) return true;
  if (!fastq) return p[o] == '>';
  if (p[o] != '@') return false;
  const size_t l2 = next_line(p, next_line(p, o, n), n);
  return l2 < n && p[l2] == '+';
}
struct Rec { size_t off; uint32_t len; size_t hdr; };          // sequence text starts at off (FASTA: first sequence line), len letters; header line at hdr
// sweep 1: records of [o, end)
void list_range(const char* p, size_t n, size_t o, size_t end, std::vector<Rec>& recs, std::string& why) {
  while (o < end) {
    if (p[o] == '\n' || p[o] == '\r') { o++; continue; }
    if (p[o] == '>') {
      const size_t h0 = o;
      o = next_line(p, o, n);
      const size_t s0 = o; size_t len = 0;
      while (o < n && p[o] != '>') { const size_t e = eol(p, o, n); len += rtrim(p, o, e) - o; o = e < n ? e + 1 : n; }
      recs.push_back({s0, (uint32_t)len, h0});
    } else if (p[o] == '@') {
      const size_t h0 = o;
      o = next_line(p, o, n);
      const size_t e = eol(p, o, n);
      recs.push_back({std::min(o, n), (uint32_t)(rtrim(p, o, e) - o), h0});
      o = e < n ? e + 1 : n;
      o = next_line(p, o, n);   // '+'
      o = next_line(p, o, n);   // quality
    } else { why = "unexpected character at byte " + std::to_string(o); return; }
  }
}
struct Lut { uint8_t v[256]; Lut() { for (int c = 0; c < 256; c++) v[c] = (uint8_t)code_of((unsigned char)c); } };
const Lut g_lut;
// packs letters s[0..m) as letters k0.. of the record whose code words start at cp and mask words at mp (zero-initialised)
inline void pack_piece(uint32_t* cp, uint32_t* mp, size_t k0, const char* s, size_t m) {
  size_t k = k0, i = 0;
  while (i < m && (k & 15)) { const uint32_t c = g_lut.v[(unsigned char)s[i]]; if (c == 4) mp[k >> 5] |= 1u << (k & 31); else cp[k >> 4] |= c << ((k & 15) * 2); i++; k++; }
  for (; i + 16 <= m; i += 16, k += 16) {
    uint32_t w = 0, amb = 0;
    for (int j = 0; j < 16; j++) { const uint32_t c = g_lut.v[(unsigned char)s[i + j]]; w |= (c & 3) << (2 * j); amb |= (c >> 2) << j; }
    cp[k >> 4] = w;
    if (amb) mp[k >> 5] |= amb << (k & 31);
  }
  for (; i < m; i++, k++) { const uint32_t c = g_lut.v[(unsigned char)s[i]]; if (c == 4) mp[k >> 5] |= 1u << (k & 31); else cp[k >> 4] |= c << ((k & 15) * 2); }
}
}  // namespace

namespace {
int load_fastx_impl(const char* path, uint32_t threads, bool keep_text, smr_reads** out, char* err, size_t errcap) {
  if (!path || !out) return SMR_ERR_ARG;
  auto fail = [&](const std::string& w) { if (err && errcap) snprintf(err, errcap, "%s", w.c_str()); return SMR_ERR_IO; };
  auto bp = std::make_shared<Bytes>();
  Bytes& b = *bp; std::string w0;
  if (!slurp(path, b, w0)) return fail(w0);
  const char* p = b.p; const size_t n = b.n;
  if (threads == 0) threads = smr::host_threads();
  size_t first = 0;
  while (first < n && (p[first] == '\n' || p[first] == '\r')) first++;
  const bool fastq = first < n && p[first] == '@';
  if (first < n && p[first] != '@' && p[first] != '>') return fail(std::string(path) + ": neither FASTA nor FASTQ");
  threads = (uint32_t)std::min<size_t>(threads, std::max<size_t>(1, n / (1u << 16)));
  std::vector<size_t> cut(threads + 1, n);
  cut[0] = first;
  for (uint32_t t = 1; t < threads; t++) {
    size_t o = std::max(cut[t - 1], n / threads * t);
    o = o == 0 ? 0 : next_line(p, o - 1, n);                       // start of the next line
    while (o < n && !is_record_start(p, o, n, fastq)) o = next_line(p, o, n);
    cut[t] = o;
  }
  auto run = [&](auto&& fn) { std::vector<std::thread> th; for (uint32_t t = 1; t < threads; t++) th.emplace_back(fn, t); fn(0u); for (auto& x : th) x.join(); };
  std::vector<std::vector<Rec>> recs(threads);
  std::vector<std::string> why(threads);
  std::vector<uint64_t> nrec(threads + 1, 0), nword(threads + 1, 0);
  std::vector<uint64_t> tlen(threads, 0);
  std::vector<uint32_t> tmin(threads, 0xffffffffu), tmax(threads, 0);
  run([&](uint32_t t) {
    if (cut[t] >= cut[t + 1]) return;
    recs[t].reserve((cut[t + 1] - cut[t]) / (fastq ? 256 : 128) + 16);
    list_range(p, n, cut[t], cut[t + 1], recs[t], why[t]);
    uint64_t wsum = 0;
    for (const Rec& r : recs[t]) { wsum += (r.len + 15) / 16 + (r.len + 31) / 32; tlen[t] += r.len; tmin[t] = std::min(tmin[t], r.len); tmax[t] = std::max(tmax[t], r.len); }
    nrec[t + 1] = recs[t].size(); nword[t + 1] = wsum;
  });
  for (uint32_t t = 0; t < threads; t++) if (!why[t].empty()) return fail(std::string(path) + ": " + why[t]);
  for (uint32_t t = 0; t < threads; t++) { nrec[t + 1] += nrec[t]; nword[t + 1] += nword[t]; }
  if (nrec[threads] > 0xfffffff0ull) return fail(std::string(path) + ": more than 2^32 records in one batch");
  auto r = new smr_reads();
  r->n = (uint32_t)nrec[threads];
  r->fastq = fastq;
  // (the arrays are value-initialised by the one thread that sizes them: 2 MB pages make that first touch cheap -- smr_hostmem.hpp)
  if (keep_text) { r->text_owner = bp; r->text = p; r->text_n = n; smr::reserve_huge(r->hdr_off, r->n); smr::reserve_huge(r->seq_off, r->n); r->hdr_off.resize(r->n); r->seq_off.resize(r->n); }
  smr::reserve_huge(r->len, r->n); smr::reserve_huge(r->rec_off, (size_t)r->n + 1); smr::reserve_huge(r->words, nword[threads]);
  r->len.resize(r->n); r->rec_off.resize((size_t)r->n + 1); r->words.resize(nword[threads]);
  r->rec_off[0] = 0;
  uint32_t lo = 0xffffffffu, hi = 0;
  for (uint32_t t = 0; t < threads; t++) { r->total_len += tlen[t]; lo = std::min(lo, tmin[t]); hi = std::max(hi, tmax[t]); }
  r->min_len = r->n ? lo : 0; r->max_len = hi;
  run([&](uint32_t t) {
    uint64_t wo = nword[t]; size_t k = (size_t)nrec[t];
    for (const Rec& rc : recs[t]) {
      uint32_t* cp = r->words.data() + wo; uint32_t* mp = cp + (rc.len + 15) / 16;
      if (fastq) pack_piece(cp, mp, 0, p + rc.off, rc.len);
      else { size_t o = rc.off, done = 0; while (done < rc.len) { const size_t e = eol(p, o, n), le = rtrim(p, o, e); pack_piece(cp, mp, done, p + o, le - o); done += le - o; o = e + 1; } }
      wo += (rc.len + 15) / 16 + (rc.len + 31) / 32;
      if (keep_text) { r->hdr_off[k] = rc.hdr; r->seq_off[k] = rc.off; }
      r->len[k] = rc.len; r->rec_off[++k] = wo;
    }
  });
  *out = r;
  return SMR_OK;
}
}  // namespace

extern "C" int smr_reads_load_fastx_mt(const char* path, uint32_t threads, smr_reads** out, char* err, size_t errcap) {
  return load_fastx_impl(path, threads, false, out, err, errcap);
}
extern "C" int smr_reads_load_fastx_text(const char* path, uint32_t threads, smr_reads** out, char* err, size_t errcap) {
  return load_fastx_impl(path, threads, true, out, err, errcap);
}
extern "C" int smr_reads_is_fastq(const smr_reads* r) { return r && r->fastq ? 1 : 0; }

// header line, letters (line breaks removed) and quality line of record i of a batch loaded by smr_reads_load_fastx_text
extern "C" int smr_reads_record_text(const smr_reads* r, uint32_t i, char* hdr, size_t hdr_cap, char* seq, size_t seq_cap, char* qual, size_t qual_cap, size_t lens[3]) {
  if (!r || !r->text || i >= r->n) return SMR_ERR_ARG;
  const char* p = r->text; const size_t n = r->text_n;
  auto put = [](char* dst, size_t cap, const char* src, size_t len, size_t at) {      // copies src[0..len) to dst[at..), keeps room for the NUL
    if (!dst || cap == 0 || at + 1 >= cap) return;
    const size_t k = std::min(len, cap - 1 - at);
    memcpy(dst + at, src, k); dst[at + k] = 0;
  };
  if (hdr && hdr_cap) hdr[0] = 0;
  if (seq && seq_cap) seq[0] = 0;
  if (qual && qual_cap) qual[0] = 0;
  size_t o = r->hdr_off[i];
  size_t e = eol(p, o, n), le = rtrim(p, o, e);
  put(hdr, hdr_cap, p + o, le - o, 0);
  size_t l0 = le - o, l1 = 0, l2 = 0;
  o = r->seq_off[i];
  while (l1 < r->len[i]) {
    e = eol(p, o, n); le = rtrim(p, o, e);
    put(seq, seq_cap, p + o, le - o, l1);
    l1 += le - o; o = e < n ? e + 1 : n;
    if (o >= n) break;
  }
  if (r->fastq) {
    if (r->len[i] == 0) o = next_line(p, r->seq_off[i], n);      // empty sequence line
    o = next_line(p, o, n);                                      // the '+' line
    e = eol(p, o, n); le = rtrim(p, o, e);
    put(qual, qual_cap, p + o, le > o ? le - o : 0, 0);
    l2 = le > o ? le - o : 0;
  }
  if (lens) { lens[0] = l0; lens[1] = l1; lens[2] = l2; }
  return SMR_OK;
}

extern "C" void smr_reads_free(smr_reads* r) { delete r; }

// records [first, first + count) of a packed batch as a batch of their own: the host-side read shard of one rank / one pipeline chunk
extern "C" int smr_reads_slice(const smr_reads* r, uint64_t first, uint64_t count, smr_reads** out) {
  if (!r || !out || first > r->n || count > r->n - first) return SMR_ERR_ARG;
  auto s = new smr_reads();
  s->n = (uint32_t)count;
  s->len.assign(r->len.begin() + (size_t)first, r->len.begin() + (size_t)(first + count));
  const uint64_t w0 = r->rec_off[(size_t)first], w1 = r->rec_off[(size_t)(first + count)];
  s->words.assign(r->words.begin() + (size_t)w0, r->words.begin() + (size_t)w1);
  s->rec_off.resize((size_t)count + 1);
  for (uint64_t i = 0; i <= count; i++) s->rec_off[(size_t)i] = r->rec_off[(size_t)(first + i)] - w0;
  s->min_len = count ? 0xFFFFFFFFu : 0; s->max_len = 0; s->total_len = 0;
  for (uint32_t l : s->len) { s->total_len += l; s->min_len = std::min(s->min_len, l); s->max_len = std::max(s->max_len, l); }
  s->fastq = r->fastq;
  *out = s;
  return SMR_OK;
}
// FNV-1a over lengths, record offsets and packed words: two batches with the same digest hold the same reads in the same order
extern "C" uint64_t smr_reads_digest(const smr_reads* r) {
  if (!r) return 0;
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
  mix(&r->n, 4);
  mix(r->len.data(), r->len.size() * 4);
  mix(r->rec_off.data(), r->rec_off.size() * 8);
  mix(r->words.data(), r->words.size() * 4);
  return h;
}
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
