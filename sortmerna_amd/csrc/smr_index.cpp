// smr_index.cpp -- host side of the index: loader for reference-built index files, our own builder,
// writer of the reference's on-disk format, .stats handling and the minimal-score arithmetic.
//
// Reference behaviour restated here (paths under /root/reference):
//   on-disk format      writer src/sortmerna/indexdb.cpp:730-865,1930-2080 ; reader index.cpp:143-357
//   19-mer geometry     indexdb.cpp:1434-1550 (forward tail under 9-mer prefix, reversed head under 9-mer suffix)
//   burst rule          indexdb.cpp:225-228 (bucket > 128 B bursts while depth < L+1 - L/2 - 3)
//   nucleotide maps     index: map_nt indexdb.cpp:83-109 ; SW refs: nt_table include/common.hpp:68-77
//   part split          indexdb.cpp:1384-1420 (9.5e-6 MB per (L+1)-mer, -m limit)
//   positions           indexdb.cpp:318-349,1716-1723 (file order, truncated at max_pos)
//   minimal score       refstats.cpp:238-265
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <exception>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "smr_host.hpp"
#include "smr_trie_layout.hpp"
#include "smr_hostmem.hpp"

using namespace smr;

namespace smr {
uint32_t host_threads() {
  uint32_t t = std::max(1u, std::thread::hardware_concurrency());
  if (const char* e = getenv("SMR_HOST_THREADS")) t = std::min<uint32_t>(t, (uint32_t)std::max(1, atoi(e)));
  return t;
}
}  // namespace smr

namespace {

void set_err(char* err, size_t cap, const std::string& m) {
  if (err && cap) { snprintf(err, cap, "%s", m.c_str()); }
}

bool slurp(const std::string& path, std::vector<uint8_t>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize((size_t)sz);
  bool ok = sz == 0 || fread(out.data(), 1, (size_t)sz, f) == (size_t)sz;
  fclose(f);
  return ok;
}

// read-only view of a whole file through the page cache: the GB-sized index files are parsed where they are, not copied first
struct Mapped {
  const uint8_t* p = nullptr; size_t n = 0;
  Mapped() = default;
  Mapped(const Mapped&) = delete;
  Mapped& operator=(const Mapped&) = delete;
  ~Mapped() { if (p && n) munmap(const_cast<uint8_t*>(p), n); }
  bool open(const std::string& path) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat sb;
    if (fstat(fd, &sb) != 0) { ::close(fd); return false; }
    n = (size_t)sb.st_size;
    if (n) {
      void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m == MAP_FAILED) { ::close(fd); n = 0; return false; }
      madvise(m, n, MADV_SEQUENTIAL);
      p = static_cast<const uint8_t*>(m);
    }
    ::close(fd);
    return true;
  }
  const uint8_t* data() const { return p; }
  size_t size() const { return n; }
};

// include/common.hpp:68-77 nt_table
inline uint8_t nt_sw(int c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
  }
}
// indexdb.cpp:83-109 map_nt (values of the table, not of its comment): everything else -> 0
inline uint8_t nt_index(int c) {
  switch (c) {
    case 'B': case 'C': case 'D': case 'W': case 'Y': case 'b': case 'c': case 'w': case 'y': return 1;
    case 'G': case 'K': case 'S': case 'X': case 'g': case 'k': case 's': case 'x': return 2;
    case 'T': case 'U': case 't': case 'u': return 3;
    default: return 0;
  }
}

struct Stats {
  uint64_t filesize = 0;
  std::string fasta_name;
  double bg[4] = {0, 0, 0, 0};
  uint64_t full_len = 0;
  uint32_t lnwin = 0;
  uint64_t numseq = 0;
  uint16_t nparts = 0;
  std::vector<PartStats> parts;
  std::vector<std::pair<std::string, uint32_t>> sq;
};

bool load_stats(const std::string& prefix, Stats& st) {
  std::vector<uint8_t> b;
  if (!slurp(prefix + ".stats", b)) return false;
  size_t o = 0;
  auto rd = [&](void* dst, size_t n) { if (o + n > b.size()) { o = b.size() + 1; return; } memcpy(dst, b.data() + o, n); o += n; };
  rd(&st.filesize, 8);
  uint32_t nl = 0; rd(&nl, 4);
  if (o + nl <= b.size()) st.fasta_name.assign((const char*)b.data() + o, nl ? nl - 1 : 0);
  o += nl;
  rd(st.bg, 32); rd(&st.full_len, 8); rd(&st.lnwin, 4); rd(&st.numseq, 8); rd(&st.nparts, 2);
  for (uint16_t j = 0; j < st.nparts; j++) {
    uint8_t raw[24] = {0}; rd(raw, 24);     // struct index_parts_stats {ulong, ulong, u32 + pad}
    PartStats p; memcpy(&p.start_part, raw, 8); memcpy(&p.seq_part_size, raw + 8, 8); memcpy(&p.numseq_part, raw + 16, 4);
    st.parts.push_back(p);
  }
  uint32_t nsq = 0; rd(&nsq, 4);
  for (uint32_t i = 0; i < nsq && o <= b.size(); i++) {
    uint32_t li = 0; rd(&li, 4);
    std::string id; if (o + li <= b.size()) id.assign((const char*)b.data() + o, li); o += li;
    uint32_t ls = 0; rd(&ls, 4);
    st.sq.emplace_back(id, ls);
  }
  return o <= b.size();
}

struct StageTimer {
  bool on = getenv("SMR_IB_TIMING") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char* what) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[smr index build] %-28s %.3f s\n", what, std::chrono::duration<double>(n - t).count());
    t = n;
  }
};

// (an exception in a worker -- std::bad_alloc on a damaged file's sizes -- is carried to the caller's thread instead of ending the process)
template <class F> void parallel_for(uint32_t threads, size_t n, F f) {
  if (threads <= 1 || n < 2) { f(0, n, 0u); return; }
  std::vector<std::thread> th;
  std::exception_ptr failed;
  std::mutex fm;
  size_t chunk = (n + threads - 1) / threads;
  for (uint32_t t = 0; t < threads; t++) {
    size_t lo = (size_t)t * chunk, hi = std::min(n, lo + chunk);
    if (lo >= hi) break;
    th.emplace_back([=, &failed, &fm]() { try { f(lo, hi, t); } catch (...) { std::lock_guard<std::mutex> l(fm); if (!failed) failed = std::current_exception(); } });
  }
  for (auto& x : th) x.join();
  if (failed) std::rethrow_exception(failed);
}

// References::load (references.cpp:55-159), FASTA: numseq records starting at byte `start`.
bool load_refs(const std::string& fasta, uint64_t start, uint32_t numseq, smr_index& ix) {
  Mapped mf;
  if (!mf.open(fasta)) return false;
  const uint8_t* b = mf.data();
  ix.ref_seq.clear(); ix.ref_off.assign(1, 0);
  size_t o = (size_t)start, n = mf.size();
  reserve_huge(ix.ref_seq, n > o ? n - o : 0);
  bool have = false; uint32_t done = 0;
  while (o < n && done < numseq) {
    size_t e = o;
    while (e < n && b[e] != '\n') e++;
    size_t le = e;
    while (le > o && (b[le - 1] == ' ' || b[le - 1] == '\r' || b[le - 1] == '\t' || b[le - 1] == '\f' || b[le - 1] == '\v')) le--;
    if (le > o) {
      if (b[o] == '>') {
        if (have) { ix.ref_off.push_back(ix.ref_seq.size()); done++; }
        have = true;
      } else if (have) {
        for (size_t k = o; k < le; k++) ix.ref_seq.push_back(b[k] != 32 ? nt_sw(b[k]) : (uint8_t)32);
      }
    }
    o = e + 1;
  }
  if (have && done < numseq) { ix.ref_off.push_back(ix.ref_seq.size()); done++; }
  return done == numseq;
}

// ---- compact arena construction from an explicit node list -------------------------------------
struct TmpElem { uint8_t flag = 0; uint32_t child = 0; uint32_t ent_begin = 0, ent_count = 0; };
struct TmpNode { TmpElem e[4]; };

// lays out one mini-trie (nodes[0] = root) in DFS order; entries = {tail,id} pairs in `ents`
struct TrieCounts { uint64_t n_nodes = 0, n_buckets = 0, n_entries = 0; };
bool emit_minitrie(const std::vector<TmpNode>& nodes, const std::vector<uint32_t>& ents, std::vector<uint32_t>& arena,
                   size_t& root_off, TrieCounts& ix, std::string& why) {
  // every node (4 words) and bucket starts on a 16-byte boundary so that the kernels can fetch a node with one
  // 128-bit load and a bucket entry with one 64-bit load: buckets with an odd entry count are padded by 2 zero words
  arena.resize((arena.size() + 3) & ~(size_t)3, 0);
  size_t base = arena.size();
  if (base > 0xFFFFFFF0ull) { why = "trie arena exceeds 2^32 words"; return false; }
  root_off = base;
  // DFS with explicit stack; a node's 4 words are reserved when it is first visited
  std::vector<uint32_t> node_at(nodes.size(), 0);
  struct Fr { uint32_t node; int next; };
  std::vector<Fr> st;
  arena.resize(base + 4, 0);
  node_at[0] = 0;
  st.push_back({0, 0});
  ix.n_nodes++;
  while (!st.empty()) {
    Fr& f = st.back();
    if (f.next == 4) { st.pop_back(); continue; }
    int k = f.next++;
    uint32_t nd = f.node;
    const TmpElem& el = nodes[nd].e[k];
    size_t word = base + node_at[nd] + k;
    if (el.flag == 0) { arena[word] = 0; continue; }
    size_t rel = arena.size() - base;
    if (rel > ELEM_OFF_MASK) { why = "mini-trie larger than 2^22 words"; return false; }
    if (el.flag == 2) {
      if (el.ent_count > ELEM_NENT_MAX) { why = "bucket with more than 255 entries"; return false; }
      arena[word] = (2u << ELEM_FLAG_SHIFT) | (el.ent_count << ELEM_NENT_SHIFT) | (uint32_t)rel;
      arena.insert(arena.end(), ents.begin() + (size_t)el.ent_begin * 2, ents.begin() + ((size_t)el.ent_begin + el.ent_count) * 2);
      arena.resize((arena.size() + 3) & ~(size_t)3, 0);
      ix.n_buckets++; ix.n_entries += el.ent_count;
    } else {
      arena[word] = (1u << ELEM_FLAG_SHIFT) | (uint32_t)rel;
      node_at[el.child] = (uint32_t)rel;
      arena.resize(arena.size() + 4, 0);
      ix.n_nodes++;
      st.push_back({el.child, 0});   // note: invalidates f
    }
  }
  return true;
}

// BFS stream of one mini-trie (index.cpp:176-316) -> TmpNode list
bool parse_bfs(const uint8_t* b, size_t bn, size_t& o, std::vector<TmpNode>& nodes, std::vector<uint32_t>& ents, std::vector<uint8_t>& flags) {
  nodes.clear(); ents.clear(); flags.clear();
  size_t hf = 0;
  auto rd8 = [&]() -> uint8_t { return o < bn ? b[o++] : (o++, (uint8_t)0); };
  nodes.emplace_back();
  for (int i = 0; i < 4; i++) flags.push_back(rd8());
  for (size_t head = 0; head < nodes.size(); head++) {
    for (int i = 0; i < 4; i++) {
      uint8_t fl = flags[hf++];
      TmpElem el;
      el.flag = fl;
      if (fl == 1) {
        for (int k = 0; k < 4; k++) flags.push_back(rd8());
        el.child = (uint32_t)nodes.size();
        nodes.emplace_back();
      } else if (fl == 2) {
        uint32_t sz = 0;
        if (o + 4 <= bn) memcpy(&sz, b + o, 4);
        o += 4;
        if (o + sz > bn) return false;
        el.ent_begin = (uint32_t)(ents.size() / 2); el.ent_count = sz / 8;
        size_t old = ents.size(); ents.resize(old + sz / 4);
        memcpy(ents.data() + old, b + o, sz);
        o += sz;
      } else if (fl != 0) {
        return false;
      }
      nodes[head].e[i] = el;
    }
  }
  return o <= bn;
}

// the same walk without building anything: where does the stream of this mini-trie end?
bool skip_bfs(const uint8_t* b, size_t bn, size_t& o, std::vector<uint8_t>& flags) {
  flags.clear();
  size_t hf = 0, n_nodes = 1;
  auto rd8 = [&]() -> uint8_t { return o < bn ? b[o++] : (o++, (uint8_t)0); };
  for (int i = 0; i < 4; i++) flags.push_back(rd8());
  for (size_t head = 0; head < n_nodes; head++) {
    for (int i = 0; i < 4; i++) {
      const uint8_t fl = flags[hf++];
      if (fl == 1) { for (int k = 0; k < 4; k++) flags.push_back(rd8()); n_nodes++; }
      else if (fl == 2) {
        uint32_t sz = 0;
        if (o + 4 <= bn) memcpy(&sz, b + o, 4);
        o += 4 + (size_t)sz;
        if (o > bn) return false;
      } else if (fl != 0) return false;
    }
  }
  return o <= bn;
}

}  // namespace

// =================================================================================================
extern "C" int smr_index_load_files(const char* prefix, uint32_t part, const char* ref_fasta, smr_index** out, char* err, size_t errcap) {
  if (!prefix || !ref_fasta || !out) return SMR_ERR_ARG;
  Stats st;
  if (!load_stats(prefix, st)) { set_err(err, errcap, std::string("cannot read ") + prefix + ".stats"); return SMR_ERR_IO; }
  if (part >= st.nparts) { set_err(err, errcap, "part out of range"); return SMR_ERR_ARG; }
  if (st.lnwin < 8 || st.lnwin > 20 || (st.lnwin & 1)) { set_err(err, errcap, "seed length L must be even, 8..20"); return SMR_ERR_ARG; }
  auto ix = new smr_index();
  ix->lnwin = st.lnwin; ix->part = part; ix->n_parts = st.nparts;
  memcpy(ix->bg, st.bg, sizeof st.bg); ix->full_len = st.full_len; ix->numseq = st.numseq; ix->filesize = st.filesize;
  ix->parts = st.parts; ix->sq_header = st.sq;
  std::string p = std::to_string(part);
  uint32_t nk = 1u << st.lnwin;
  StageTimer tm;
  Mapped kb, tb, pb;                                          // parsed in place (page cache), by all host threads
  if (!kb.open(std::string(prefix) + ".kmer_" + p + ".dat") || !tb.open(std::string(prefix) + ".bursttrie_" + p + ".dat") ||
      !pb.open(std::string(prefix) + ".pos_" + p + ".dat")) {
    delete ix; set_err(err, errcap, "cannot read index part files"); return SMR_ERR_IO;
  }
  uint32_t threads = std::min<uint32_t>(64, smr::host_threads());
  if (const char* e = getenv("SMR_LOAD_THREADS")) threads = std::min<uint32_t>(256, std::max(1, atoi(e)));      // test aid: many loader threads on a small machine
  // the reference sequences and the position lists load in threads of their own while the tries are parsed
  bool refs_ok = false;
  // (an exception inside a std::thread would end the process: the bodies catch and report -- a damaged file must come back as SMR_ERR_IO)
  std::thread t_refs([&]() { try { refs_ok = load_refs(ref_fasta, st.parts[part].start_part, st.parts[part].numseq_part, *ix); } catch (const std::exception&) { refs_ok = false; } });
  std::string pos_err;
  std::thread t_pos([&]() { try {
    // positions (index.cpp:322-352) -> CSR; keep file order (sorted by seq, then pos, by construction)
    const uint8_t* b = pb.data(); const size_t bn = pb.size();
    uint32_t nid = 0;
    if (bn >= 4) memcpy(&nid, b, 4);
    if (bn < 4 || (size_t)nid > (bn - 4) / 4) { pos_err = "malformed pos file: it cannot hold the number of lists it announces"; return; }      // every list has at least its 4-byte size: checked BEFORE anything is sized by nid
    ix->pos_off.assign((size_t)nid + 1, 0);
    std::vector<size_t> src((size_t)nid + 1, 0);             // byte offset of list i in the file
    size_t o = 4; uint64_t total = 0;
    for (uint32_t i = 0; i < nid; i++) {
      uint32_t sz = 0;
      if (o + 4 <= bn) memcpy(&sz, b + o, 4);
      o += 4;
      if (o + (size_t)sz * 8 > bn) { pos_err = "malformed pos file"; return; }
      src[i] = o; o += (size_t)sz * 8; total += sz;
      if (total > 0xFFFFFFFFull) { pos_err = "more than 2^32 positions"; return; }
      ix->pos_off[i + 1] = (uint32_t)total;
    }
    reserve_huge(ix->pos_arr, (size_t)total * 2);
    ix->pos_arr.resize((size_t)total * 2);
    parallel_for(std::max(1u, threads / 4), nid, [&](size_t lo, size_t hi, uint32_t) {
      std::vector<std::pair<uint32_t, uint32_t>> v;
      for (size_t i = lo; i < hi; i++) {
        const uint32_t sz = ix->pos_off[i + 1] - ix->pos_off[i];
        uint32_t* dst = ix->pos_arr.data() + (size_t)ix->pos_off[i] * 2;
        memcpy(dst, b + src[i], (size_t)sz * 8);
        // the device code binary-searches each list by seq: enforce (seq,pos) order
        bool sorted = true;
        for (uint32_t k = 1; k < sz && sorted; k++) {
          const uint32_t* a = dst + (size_t)(k - 1) * 2;
          if (a[1] > a[3] || (a[1] == a[3] && a[0] > a[2])) sorted = false;
        }
        if (!sorted) {
          v.resize(sz);
          for (uint32_t k = 0; k < sz; k++) v[k] = {dst[2 * k + 1], dst[2 * k]};
          std::sort(v.begin(), v.end());
          for (uint32_t k = 0; k < sz; k++) { dst[2 * k] = v[k].second; dst[2 * k + 1] = v[k].first; }
        }
      }
    });
  } catch (const std::exception& e) { pos_err = std::string("pos file: ") + e.what(); } });
  ix->lookup.assign(nk, Lookup{0, NONE, NONE, 0, 0});
  if (kb.size() < (size_t)nk * 4) { t_pos.join(); t_refs.join(); delete ix; set_err(err, errcap, "malformed kmer file: shorter than 4^(L/2) counts"); return SMR_ERR_IO; }
  for (uint32_t i = 0; i < nk && (size_t)(i + 1) * 4 <= kb.size(); i++) memcpy(&ix->lookup[i].count, kb.data() + (size_t)i * 4, 4);
  // mini-tries: one sequential walk over the BFS streams finds where each begins (the sizes in the file are the reference's in-memory
  // sizes, index.cpp:178-190, not stream lengths); then the streams are parsed and laid out by all threads, chunks of TRIE_CHUNK k-mers
  // handed out dynamically (the key ranges are far from equally heavy), every chunk into a vector of its exact size, and the arena is the
  // chunks in k-mer order (so the layout does not depend on who parsed what).
  // (Walk and parse side by side -- chunks handed out as soon as the walk had passed them -- was measured too: 4.3 s -> 2.0 s on 8 cores,
  // but on the 64 loader threads of the GPU box the WALK went from 0.56 s to 4.3 s: its page faults on the mapped file queue behind the
  // address-space lock that the parsers' allocations take for writing.  profiles/r3s29_e2e_quick.log)
  std::string why;
  bool tries_ok = true;
  {
    const uint8_t* b = tb.data(); const size_t bn = tb.size();
    constexpr size_t TRIE_CHUNK = 256;
    const size_t n_chunks = ((size_t)nk + TRIE_CHUNK - 1) / TRIE_CHUNK;
    std::vector<size_t> start(2 * (size_t)nk, (size_t)-1);
    {
      std::vector<uint8_t> flags;
      size_t o = 0;
      for (uint32_t i = 0; i < nk && tries_ok; i++) {
        uint32_t sz[2] = {0, 0};
        if (o + 8 > bn) { tries_ok = false; why = "file ends before the last k-mer"; break; }      // the reference writes the two sizes for every k-mer (indexdb.cpp:719-742)
        memcpy(sz, b + o, 8);
        o += 8;
        if (ix->lookup[i].count == 0) continue;       // index.cpp:190: tries are only read when count != 0
        for (int j = 0; j < 2 && tries_ok; j++) {
          if (sz[j] == 0) continue;
          start[2 * (size_t)i + j] = o;
          if (!skip_bfs(b, bn, o, flags)) { tries_ok = false; why = "truncated stream"; }
        }
      }
    }
    tm.lap("load: mini-trie boundaries");
    std::atomic<bool> parse_failed{false};
    std::atomic<size_t> next_chunk{0};
    std::vector<std::vector<uint32_t>> local(n_chunks);
    std::vector<size_t> rootw(2 * (size_t)nk, (size_t)-1), endw(2 * (size_t)nk, 0);     // word offsets inside the chunk's vector
    const uint32_t workers = std::max(1u, threads);
    std::vector<TrieCounts> cnt(workers);
    std::vector<std::string> twhy(workers);
    auto work = [&](uint32_t tid) {
      try {
        std::vector<TmpNode> nodes; std::vector<uint32_t> ents, buf; std::vector<uint8_t> flags;
        for (;;) {
          const size_t c = next_chunk.fetch_add(1);
          if (c >= n_chunks || parse_failed.load(std::memory_order_relaxed)) return;
          const size_t lo = c * TRIE_CHUNK, hi = std::min((size_t)nk, lo + TRIE_CHUNK);
          buf.clear();                                 // the thread's scratch keeps its capacity: one allocation per chunk (the exact copy below), not a series of regrowths
          for (size_t i = lo; i < hi; i++)
            for (int j = 0; j < 2; j++) {
              size_t o = start[2 * i + j];
              if (o == (size_t)-1) continue;
              if (!parse_bfs(b, bn, o, nodes, ents, flags) || !emit_minitrie(nodes, ents, buf, rootw[2 * i + j], cnt[tid], twhy[tid])) {
                if (twhy[tid].empty()) twhy[tid] = "malformed stream";
                parse_failed.store(true);
                return;
              }
              endw[2 * i + j] = buf.size();
            }
          buf.resize((buf.size() + 3) & ~(size_t)3, 0);
          local[c].assign(buf.begin(), buf.end());
        }
      } catch (const std::exception& e) { twhy[tid] = std::string("mini-tries: ") + e.what(); parse_failed.store(true); }
    };
    if (tries_ok) {
      std::vector<std::thread> th;
      for (uint32_t t = 1; t < workers; t++) th.emplace_back(work, t);
      work(0);
      for (auto& x : th) x.join();
    }
    tm.lap("load: mini-tries parsed");
    for (uint32_t t = 0; t < workers && tries_ok; t++) if (!twhy[t].empty()) { tries_ok = false; why = twhy[t]; }
    size_t total = 0;
    std::vector<size_t> tbase(n_chunks, 0);
    for (size_t c = 0; c < n_chunks; c++) { tbase[c] = total; total += local[c].size(); }
    if (tries_ok && total > 0xFFFFFFF0ull) { tries_ok = false; why = "trie arena exceeds 2^32 words"; }
    if (tries_ok) {
      reserve_huge(ix->trie, total);
      ix->trie.resize(total);
      tm.lap("load: arena sized");
      parallel_for(threads, n_chunks, [&](size_t lo, size_t hi, uint32_t) {
        for (size_t c = lo; c < hi; c++) {
          if (!local[c].empty()) { memcpy(ix->trie.data() + tbase[c], local[c].data(), local[c].size() * 4); std::vector<uint32_t>().swap(local[c]); }   // (freed as it goes: the peak is not twice the arena)
          for (size_t i = c * TRIE_CHUNK; i < std::min((size_t)nk, (c + 1) * TRIE_CHUNK); i++)
            for (int j = 0; j < 2; j++) {
              if (rootw[2 * i + j] == (size_t)-1) continue;
              const uint32_t root = (uint32_t)(tbase[c] + rootw[2 * i + j]), words = (uint32_t)(endw[2 * i + j] - rootw[2 * i + j]);
              if (j == 0) { ix->lookup[i].rootF = root; ix->lookup[i].wordsF = words; }
              else { ix->lookup[i].rootR = root; ix->lookup[i].wordsR = words; }
            }
        }
      });
      for (uint32_t t = 0; t < workers; t++) { ix->n_nodes += cnt[t].n_nodes; ix->n_buckets += cnt[t].n_buckets; ix->n_entries += cnt[t].n_entries; }
    }
  }
  tm.lap("load: mini-tries");
  t_pos.join(); t_refs.join();
  if (!tries_ok) { delete ix; set_err(err, errcap, "malformed burst trie file: " + why); return SMR_ERR_IO; }
  if (!pos_err.empty()) { delete ix; set_err(err, errcap, pos_err); return SMR_ERR_IO; }
  if (!refs_ok) { delete ix; set_err(err, errcap, std::string("cannot load reference sequences from ") + ref_fasta); return SMR_ERR_IO; }
  tm.lap("load: positions and reference sequences (waited for)");
  smr_build_lkc(*ix);                                       // (the pigeonhole layout of the tries is built on the device by smr_index_upload)
  *out = ix;
  return SMR_OK;
}

namespace {
struct PgEnt { uint32_t str, id; };
// entries below `node` in the reference's DFS order (A<C<G<T, bucket order) as complete strings; `pre` = the plen chars of the path to `node`
void pg_collect(const uint32_t* t, uint32_t node, uint32_t pre, uint32_t plen, std::vector<PgEnt>& out) {
  for (uint32_t ne = 0; ne < 4; ne++) {
    const uint32_t e = t[node + ne], fl = e >> ELEM_FLAG_SHIFT;
    const uint32_t p2 = pre | (ne << (2 * plen));
    if (fl == 2) {
      const uint32_t n = (e >> ELEM_NENT_SHIFT) & 0xFFu;
      const uint32_t* b = t + (e & ELEM_OFF_MASK);
      for (uint32_t q = 0; q < n; q++) out.push_back({p2 | (b[2 * q] << (2 * (plen + 1))), b[2 * q + 1]});
    } else if (fl == 1) pg_collect(t, e & ELEM_OFF_MASK, p2, plen + 1, out);
  }
}
}  // namespace

// ---- self check of the pigeonhole layout ----------------------------------------------------------
// Both arrays of a block must hold exactly the reference-shaped mini-trie's entries (rank r = the r-th entry of its DFS), sorted by their
// keys, and the directories must bound the keys.
extern "C" int smr_index_selfcheck(smr_index* ix, char* err, size_t errcap) {
  if (!ix) return SMR_ERR_ARG;
  std::string why;
  if (!smr_build_pigeonhole(*ix, 0, why)) { set_err(err, errcap, why); return SMR_ERR_CAPACITY; }
  const uint32_t pw = ix->lnwin / 2, h = pw / 2;
  std::vector<PgEnt> a;
  std::vector<uint8_t> seen;
  for (size_t k = 0; k < ix->lookup.size(); k++)
    for (int d = 0; d < 2; d++) {
      const uint32_t r1 = d == 0 ? ix->lookup[k].rootF : ix->lookup[k].rootR, r3 = ix->root3[2 * (2 * k + d)], meta = ix->root3[2 * (2 * k + d) + 1];
      const std::string at = " at key " + std::to_string(k);
      if ((r1 == NONE) != (r3 == NONE)) { set_err(err, errcap, "pigeonhole layout: root presence differs" + at); return SMR_ERR_STATE; }
      if (r1 == NONE) continue;
      a.clear();
      pg_collect(ix->trie.data() + r1, 0, 0, 0, a);
      const uint32_t n = meta & 0xFFFFFFu, cA = (meta >> 24) & 15u, cB = meta >> 28;
      uint32_t wA, wB;
      pg_chars(n, pw, wA, wB);
      if (n != a.size() || cA != wA || cB != wB) { set_err(err, errcap, "pigeonhole layout: block header wrong" + at); return SMR_ERR_STATE; }
      const uint32_t* blk = ix->pg.data() + (size_t)r3 * 4;
      const uint32_t nA = cA ? (1u << (2 * cA)) + 1 : 0, nB = cA ? (1u << (2 * cB)) + 1 : 0;
      for (int o = 0; o < (cA ? 2 : 1); o++) {
        const uint32_t* Ts = blk + nA + nB + (o ? n : 0);
        const uint32_t* Rs = blk + nA + nB + (cA ? 2 : 1) * (size_t)n + (o ? 2 * (size_t)n : 0);
        const uint32_t* dir = o ? blk + nA : blk;
        const uint32_t c = o ? cB : cA, from = o ? h : 0;
        seen.assign(n, 0);
        uint64_t prev = 0;
        for (uint32_t i = 0; i < n; i++) {
          const uint32_t T = Ts[i], r = Rs[2 * i];
          if (r >= n || seen[r] || a[r].str != T || ix->pos_off[a[r].id] + a[r].id != Rs[2 * i + 1]) { set_err(err, errcap, "pigeonhole layout: entries differ" + at); return SMR_ERR_STATE; }      // (the layout's ids are the places of the seeds' position lists: pos_off[id] + id)
          seen[r] = 1;
          if (!cA) { if (r != i) { set_err(err, errcap, "pigeonhole layout: scan block not in DFS order" + at); return SMR_ERR_STATE; } continue; }
          const uint64_t key = o ? pg_key(T, h, pw - h) : pg_key(T, 0, pw + 1);
          if (i && key < prev) { set_err(err, errcap, "pigeonhole layout: array not sorted" + at); return SMR_ERR_STATE; }
          prev = key;
          const uint32_t kk = pg_key(T, from, c);
          if (!(dir[kk] <= i && i < dir[kk + 1])) { set_err(err, errcap, "pigeonhole layout: directory wrong" + at); return SMR_ERR_STATE; }
        }
        if (cA && (dir[0] != 0 || dir[(1u << (2 * c))] != n)) { set_err(err, errcap, "pigeonhole layout: directory ends wrong" + at); return SMR_ERR_STATE; }
      }
    }
  return SMR_OK;
}

extern "C" void smr_index_free(smr_index* ix) { delete ix; }

extern "C" int smr_index_get_info(const smr_index* ix, smr_index_info* o) {
  if (!ix || !o) return SMR_ERR_ARG;
  o->lnwin = ix->lnwin; o->n_kmers = (uint32_t)ix->lookup.size(); o->trie_words = ix->trie.size();
  o->n_ids = ix->n_ids(); o->n_pos = ix->pos_arr.size() / 2; o->n_refs = ix->n_refs(); o->ref_bytes = ix->ref_seq.size();
  o->n_nodes = ix->n_nodes; o->n_buckets = ix->n_buckets; o->n_entries = ix->n_entries;
  memcpy(o->bg, ix->bg, sizeof o->bg); o->full_len = ix->full_len; o->numseq = ix->numseq; o->n_parts = ix->n_parts;
  return SMR_OK;
}

// refstats.cpp:238-257: expected HSP length, length-corrected sizes
// (full_read_scale: 1, or the number of processing threads under -score_split -- refstats.cpp:247: the reads are then scored per split)
extern "C" void smr_refstats_corrected_split(double K, const double bg[4], uint64_t full_ref, uint64_t numseq, uint64_t all_reads_count, uint64_t all_reads_len,
                                             uint32_t full_read_scale, uint64_t* full_ref_corr, uint64_t* full_read_corr) {
  const int scale = full_read_scale ? (int)full_read_scale : 1;
  double H = -(bg[0] * std::log2(bg[0]) + bg[1] * std::log2(bg[1]) + bg[2] * std::log2(bg[2]) + bg[3] * std::log2(bg[3]));
  uint64_t full_read = all_reads_len;
  uint64_t expect_L = static_cast<uint64_t>(std::log(K * full_ref * full_read / scale) / H);
  if (full_ref > expect_L * numseq) full_ref -= expect_L * numseq;
  full_read -= expect_L * all_reads_count / scale;
  if (full_ref_corr) *full_ref_corr = full_ref;
  if (full_read_corr) *full_read_corr = full_read;
}
extern "C" void smr_refstats_corrected(double K, const double bg[4], uint64_t full_ref, uint64_t numseq, uint64_t all_reads_count, uint64_t all_reads_len,
                                       uint64_t* full_ref_corr, uint64_t* full_read_corr) {
  smr_refstats_corrected_split(K, bg, full_ref, numseq, all_reads_count, all_reads_len, 1, full_ref_corr, full_read_corr);
}

// refstats.cpp:238-265
extern "C" uint32_t smr_minimal_score_split(double lambda, double K, const double bg[4], uint64_t full_ref, uint64_t numseq,
                                            uint64_t all_reads_count, uint64_t all_reads_len, double evalue, uint32_t full_read_scale) {
  const int scale = full_read_scale ? (int)full_read_scale : 1;
  uint64_t full_read = 0;
  smr_refstats_corrected_split(K, bg, full_ref, numseq, all_reads_count, all_reads_len, (uint32_t)scale, &full_ref, &full_read);
  return static_cast<uint32_t>(std::log(evalue / ((double)K * full_ref * full_read / scale)) / -lambda);
}
extern "C" uint32_t smr_minimal_score(double lambda, double K, const double bg[4], uint64_t full_ref, uint64_t numseq,
                                      uint64_t all_reads_count, uint64_t all_reads_len, double evalue) {
  return smr_minimal_score_split(lambda, K, bg, full_ref, numseq, all_reads_count, all_reads_len, evalue, 1);
}

// Test seam: the pigeonhole layout of a host index as the HOST transform builds it (smr_build_pigeonhole; built on first use) -- the arena
// and the block table k_seed_pg reads (smr_host.hpp).  The pointers stay valid as long as the index.
extern "C" int smr_index_pigeonhole(smr_index* ix, const uint32_t** pg, uint64_t* pg_words, const uint32_t** root3, uint64_t* root3_words, char* err, size_t errcap) {
  if (!ix || !pg || !pg_words || !root3 || !root3_words) { set_err(err, errcap, "smr_index_pigeonhole: null argument"); return SMR_ERR_ARG; }
  std::string why;
  if (!smr_build_pigeonhole(*ix, 0, why)) { set_err(err, errcap, why); return SMR_ERR_CAPACITY; }
  *pg = ix->pg.data(); *pg_words = ix->pg.size(); *root3 = ix->root3.data(); *root3_words = ix->root3.size();
  return SMR_OK;
}

// =================================================================================================
// Our own builder.
// =================================================================================================
namespace {

struct SeqRec { uint64_t file_start; uint64_t file_end; uint64_t seq_begin; uint32_t len; std::string id; };

// parse the FASTA like build_index() does (indexdb.cpp:1198-1260): header up to '\n', sequence = every
// char except '\n' and ' '
bool parse_fasta(const std::vector<uint8_t>& b, std::vector<SeqRec>& recs, std::vector<uint8_t>& raw, std::string& why) {
  size_t o = 0, n = b.size();
  while (o < n) {
    if (b[o] != '>') { why = "each reference header must begin with '>'"; return false; }
    SeqRec r; r.file_start = o;
    size_t e = o + 1; bool stop = false;
    while (e < n && b[e] != '\n') {
      if (b[e] != ' ' && b[e] != '\t' && !stop) r.id.push_back((char)b[e]); else stop = true;
      e++;
    }
    o = e + 1;
    r.seq_begin = raw.size();
    while (o < n && b[o] != '>') { if (b[o] != '\n' && b[o] != ' ') raw.push_back(b[o]); o++; }
    r.len = (uint32_t)(raw.size() - r.seq_begin);
    r.file_end = o;
    recs.push_back(std::move(r));
  }
  return true;
}

// SMR_IB_TIMING=1: stage times of the index build on stderr

// sort u64 keys: partition by the top `topbits` bits (counting), then std::sort every bucket in parallel
void bucket_sort_u64(std::vector<uint64_t>& a, int keybits, uint32_t threads) {
  size_t n = a.size();
  if (n < (1u << 16) || threads <= 1) { std::sort(a.begin(), a.end()); return; }
  const int topbits = 12; const uint32_t nb = 1u << topbits; const int sh = keybits - topbits;
  std::vector<std::vector<size_t>> hist(threads, std::vector<size_t>(nb, 0));
  parallel_for(threads, n, [&](size_t lo, size_t hi, uint32_t t) { for (size_t i = lo; i < hi; i++) hist[t][a[i] >> sh]++; });
  std::vector<size_t> start(nb + 1, 0);
  std::vector<std::vector<size_t>> off(threads, std::vector<size_t>(nb, 0));
  size_t acc = 0;
  for (uint32_t b = 0; b < nb; b++) { start[b] = acc; for (uint32_t t = 0; t < threads; t++) { off[t][b] = acc; acc += hist[t][b]; } }
  start[nb] = acc;
  std::vector<uint64_t> tmp(n);
  parallel_for(threads, n, [&](size_t lo, size_t hi, uint32_t t) { for (size_t i = lo; i < hi; i++) tmp[off[t][a[i] >> sh]++] = a[i]; });
  a.swap(tmp);
  std::atomic<uint32_t> next(0);
  std::vector<std::thread> th;
  for (uint32_t t = 0; t < threads; t++) th.emplace_back([&]() {
    for (;;) { uint32_t b = next.fetch_add(1); if (b >= nb) break; std::sort(a.begin() + start[b], a.begin() + start[b + 1]); }
  });
  for (auto& x : th) x.join();
}

}  // namespace

// ---- pigeonhole arena (smr_host.hpp) -------------------------------------------------------------

namespace {
uint32_t pg_count(const uint32_t* t, uint32_t node) {
  uint32_t n = 0;
  for (int ne = 0; ne < 4; ne++) {
    const uint32_t e = t[node + ne], fl = e >> ELEM_FLAG_SHIFT;
    if (fl == 2) n += (e >> ELEM_NENT_SHIFT) & 0xFFu;
    else if (fl == 1) n += pg_count(t, e & ELEM_OFF_MASK);
  }
  return n;
}
// pg_collect that also notes where each bucket starts in `out`
void pg_collect_b(const uint32_t* t, uint32_t node, uint32_t pre, uint32_t plen, std::vector<PgEnt>& out, std::vector<uint32_t>& bstart) {
  for (uint32_t ne = 0; ne < 4; ne++) {
    const uint32_t e = t[node + ne], fl = e >> ELEM_FLAG_SHIFT;
    const uint32_t p2 = pre | (ne << (2 * plen));
    if (fl == 2) {
      const uint32_t n = (e >> ELEM_NENT_SHIFT) & 0xFFu;
      const uint32_t* b = t + (e & ELEM_OFF_MASK);
      bstart.push_back((uint32_t)out.size());
      for (uint32_t q = 0; q < n; q++) out.push_back({p2 | (b[2 * q] << (2 * (plen + 1))), b[2 * q + 1]});
    } else if (fl == 1) pg_collect_b(t, e & ELEM_OFF_MASK, p2, plen + 1, out, bstart);
  }
}
inline size_t pg_block_words(uint32_t n, uint32_t pw) {
  uint32_t cA, cB;
  pg_chars(n, pw, cA, cB);
  const size_t w = cA ? (size_t)(1u << (2 * cA)) + 1 + (1u << (2 * cB)) + 1 + 6 * (size_t)n : 3 * (size_t)n;
  return (w + 3) & ~(size_t)3;
}
}  // namespace

bool smr_build_pigeonhole(smr_index& ix, uint32_t threads, std::string& why) {
  std::lock_guard<std::mutex> once(ix.pg_mutex);
  if (!ix.root3.empty()) return true;
  StageTimer tm;
  const size_t nk = ix.lookup.size();
  const uint32_t pw = ix.lnwin / 2, h = pw / 2;
  if (threads == 0) threads = smr::host_threads();
  threads = std::min<uint32_t>(threads, 64);
  // pass 1: entries per mini-trie -> block sizes -> block offsets
  std::vector<uint32_t> cnt(2 * nk, 0);
  parallel_for(threads, nk, [&](size_t lo, size_t hi, uint32_t) {
    for (size_t k = lo; k < hi; k++)
      for (int d = 0; d < 2; d++) {
        const uint32_t root = d == 0 ? ix.lookup[k].rootF : ix.lookup[k].rootR;
        if (root != NONE) cnt[2 * k + d] = pg_count(ix.trie.data() + root, 0);
      }
  });
  std::vector<uint32_t> root3(4 * nk, 0);
  size_t total = 0;
  for (size_t i = 0; i < 2 * nk; i++) {
    const bool present = (i & 1 ? ix.lookup[i >> 1].rootR : ix.lookup[i >> 1].rootF) != NONE;
    if (!present) { root3[2 * i] = NONE; continue; }
    if (cnt[i] > 0xFFFFFFu) { why = "a mini-trie is too large for the pigeonhole layout"; return false; }
    if (total / 4 > 0xFFFFFFF0ull) { why = "pigeonhole arena exceeds 2^34 words"; return false; }
    uint32_t cA, cB;
    pg_chars(cnt[i], pw, cA, cB);
    root3[2 * i] = (uint32_t)(total / 4);
    root3[2 * i + 1] = cnt[i] | (cA << 24) | (cB << 28);
    total += pg_block_words(cnt[i], pw);
  }
  if (!ix.pg.resize_uninitialized(total + 4)) { why = "out of host memory for the pigeonhole arena"; return false; }      // + one block of slack: a 16-byte read at the last word stays inside
  for (size_t i = 0; i < 4; i++) ix.pg.data()[total + i] = 0;
  tm.lap("pigeonhole layout: sizes");
  // pass 2: every block in place
  parallel_for(threads, nk, [&](size_t lo, size_t hi, uint32_t) {
    std::vector<PgEnt> v;
    std::vector<uint32_t> bstart, ordB, cntB;
    std::vector<uint64_t> ord;
    for (size_t k = lo; k < hi; k++) {
      for (int d = 0; d < 2; d++) {
        const uint32_t root = d == 0 ? ix.lookup[k].rootF : ix.lookup[k].rootR;
        if (root == NONE) continue;
        v.clear(); bstart.clear();
        pg_collect_b(ix.trie.data() + root, 0, 0, 0, v, bstart);   // complete strings (char j at bits 2j) in DFS order
        for (auto& e : v) e.id = ix.pos_off[e.id] + e.id;          // (what the searches hand on: where the seed's position list lies on the device -- behind a header word, k_pos2_build)
        const uint32_t n = (uint32_t)v.size();
        uint32_t cA, cB;
        pg_chars(n, pw, cA, cB);
        uint32_t* blk = ix.pg.data() + (size_t)root3[2 * (2 * k + d)] * 4;
        const size_t words = pg_block_words(n, pw);
        if (cA == 0) {
          for (uint32_t r = 0; r < n; r++) { blk[r] = v[r].str; blk[n + 2 * r] = r; blk[n + 2 * r + 1] = v[r].id; }
          for (size_t q = 3 * (size_t)n; q < words; q++) blk[q] = 0;
          continue;
        }
        const uint32_t nA = (1u << (2 * cA)) + 1, nB = (1u << (2 * cB)) + 1;
        uint32_t* dirA = blk; uint32_t* dirB = dirA + nA; uint32_t* TT = dirB + nB; uint32_t* RR = TT + 2 * (size_t)n;     // TA TB | RA RB
        for (size_t q = nA + nB + 6 * (size_t)n; q < words; q++) blk[q] = 0;
        // TA / RA: buckets come in string order (their paths are prefix-free and met in A<C<G<T order), so only each bucket is sorted
        {
          uint32_t next = 0, i = 0;
          bstart.push_back(n);
          for (size_t bq = 0; bq + 1 < bstart.size(); bq++) {
            const uint32_t b0 = bstart[bq], b1 = bstart[bq + 1];
            ord.resize(b1 - b0);
            for (uint32_t r = b0; r < b1; r++) ord[r - b0] = ((uint64_t)pg_key(v[r].str, 0, pw + 1) << 32) | r;
            std::sort(ord.begin(), ord.end());
            for (uint32_t q = 0; q < b1 - b0; q++, i++) {
              const uint32_t r = (uint32_t)(ord[q] & 0xFFFFFFFFull);
              TT[i] = v[r].str; RR[2 * i] = r; RR[2 * i + 1] = v[r].id;
              const uint32_t kk = (uint32_t)(ord[q] >> (32 + 2 * (pw + 1 - cA)));
              while (next <= kk) dirA[next++] = i;
            }
          }
          while (next < nA) dirA[next++] = n;
        }
        // TB / RB: counting sort by chars h..pw-1 (ranks ascending within a key)
        {
          const uint32_t kb = pw - h, nkeys = 1u << (2 * kb);
          cntB.assign(nkeys + 1, 0);
          ordB.resize(n);
          for (uint32_t r = 0; r < n; r++) { const uint32_t kk = pg_key(v[r].str, h, kb); ordB[r] = kk; cntB[kk + 1]++; }
          for (uint32_t q = 0; q < nkeys; q++) cntB[q + 1] += cntB[q];
          const uint32_t sh = 2 * (kb - cB);
          for (uint32_t q = 0; q < nB; q++) dirB[q] = q < nB - 1 ? cntB[(size_t)q << sh] : n;
          uint32_t* TB = TT + n; uint32_t* RB = RR + 2 * (size_t)n;
          for (uint32_t r = 0; r < n; r++) { const uint32_t i = cntB[ordB[r]]++; TB[i] = v[r].str; RB[2 * i] = r; RB[2 * i + 1] = v[r].id; }
        }
      }
    }
  });
  tm.lap("pigeonhole layout: blocks");
  ix.root3.swap(root3);
  return true;
}

// what the window scan needs of `lookup`, in one word per key (cached; several contexts may upload the same host index)
void smr_build_lkc(smr_index& ix) {
  std::lock_guard<std::mutex> once(ix.pg_mutex);
  if (!ix.lkc.empty()) return;
  const size_t nk = ix.lookup.size();
  ix.lkc.resize(nk);
  for (size_t k = 0; k < nk; k++)
    ix.lkc[k] = std::min<uint32_t>(ix.lookup[k].count, 0x3FFFFFFFu) | (ix.lookup[k].rootF != NONE ? 1u << 30 : 0u) | (ix.lookup[k].rootR != NONE ? 1u << 31 : 0u);
}

// occurrences of one part -> lookup table, mini-tries, positions (host version)
int ib_part_host(void*, const smr::IBuildInput& in, smr_index& ix, std::string& why) {
  const uint32_t L = in.L, max_pos = in.max_pos, threads = in.threads;
  const uint32_t P = L / 2, W = L + 1, T = P + 1;
    // occurrences: key = (18-mer prefix, 2L bits) << occbits | occurrence number (scan order)
    std::vector<uint64_t> occ_start((size_t)in.n_seqs + 1, 0);
    for (size_t m = 0; m < in.n_seqs; m++) occ_start[m + 1] = occ_start[m] + (in.seq_off[m + 1] - in.seq_off[m] - W + 1);
    uint64_t N = occ_start.back();
    int occbits = 1; while ((1ull << occbits) < N) occbits++;
    if ((int)(2 * L) + occbits > 64) { why = "part too large for the builder (reduce -m)"; return SMR_ERR_ARG; }
    StageTimer tm;
    std::vector<uint64_t> keys(N);
    std::vector<uint8_t> last_nt(N);     // nt at position p+L of occurrence (the 19th)
    parallel_for(threads, in.n_seqs, [&](size_t lo, size_t hi, uint32_t) {
      for (size_t m = lo; m < hi; m++) {
        const uint8_t* s = in.codes + in.seq_off[m];
        const uint32_t len = (uint32_t)(in.seq_off[m + 1] - in.seq_off[m]);
        uint64_t code = 0, mask = (L == 32) ? ~0ull : ((1ull << (2 * L)) - 1);
        for (uint32_t k = 0; k < L; k++) code = (code << 2) | s[k];
        uint64_t o = occ_start[m];
        for (uint32_t p = 0; p + W <= len; p++) {
          keys[o + p] = (code << occbits) | (o + p);
          last_nt[o + p] = s[p + L];
          if (p + W < len) code = ((code << 2) | s[p + L]) & mask;
        }
      }
    });
    tm.lap("keys");
    bucket_sort_u64(keys, 2 * (int)L + occbits, threads);
    tm.lap("sort");
    // ids, positions CSR, unique 19-mers
    const uint64_t occmask = (1ull << occbits) - 1;
    auto occ_to_seqpos = [&](uint64_t occ, uint32_t& seq, uint32_t& pos) {
      size_t m = std::upper_bound(occ_start.begin(), occ_start.end(), occ) - occ_start.begin() - 1;
      seq = (uint32_t)m; pos = (uint32_t)(occ - occ_start[m]);
    };
    // F entries (keyF, tail, id) come out already sorted; R entries need regrouping by keyR.
    // The scan over the sorted windows runs in chunks that start at group heads: every chunk builds its ids (relative), position
    // lists and entries locally, prefix sums give the chunk bases, then the pieces are copied to their final place.
    struct Chunk {
      uint64_t lo = 0, hi = 0; uint32_t n_ids = 0;
      std::vector<uint32_t> pos_cnt, pos_arr, f_key, r_key; std::vector<uint64_t> f_tail_id, r_tail_id;
    };
    const size_t nchunk = std::max<size_t>(1, std::min<size_t>((size_t)threads * 4, N / 4096 + 1));
    std::vector<Chunk> ch(nchunk);
    for (size_t c = 0; c < nchunk; c++) {
      uint64_t lo = N / nchunk * c;
      while (c > 0 && lo < N && lo > 0 && (keys[lo] >> occbits) == (keys[lo - 1] >> occbits)) lo++;      // move to the next group head
      ch[c].lo = lo;
    }
    for (size_t c = 0; c < nchunk; c++) { ch[c].hi = c + 1 < nchunk ? ch[c + 1].lo : N; if (ch[c].hi < ch[c].lo) ch[c].hi = ch[c].lo; }
    parallel_for(threads, nchunk, [&](size_t c0, size_t c1, uint32_t) {
      for (size_t c = c0; c < c1; c++) {
        Chunk& k = ch[c];
        uint32_t id = 0;
        for (uint64_t i = k.lo; i < k.hi;) {
          const uint64_t pre = keys[i] >> occbits;
          uint64_t j = i; uint32_t present = 0, stored = 0;
          while (j < k.hi && (keys[j] >> occbits) == pre) {
            const uint64_t occ = keys[j] & occmask;
            present |= 1u << last_nt[occ];
            if (max_pos == 0 || stored == 0 || stored < max_pos) {      // indexdb.cpp:318-349
              uint32_t sq_, ps_; occ_to_seqpos(occ, sq_, ps_);
              k.pos_arr.push_back(ps_); k.pos_arr.push_back(sq_); stored++;
            }
            j++;
          }
          k.pos_cnt.push_back(stored);
          for (uint32_t cc = 0; cc < 4; cc++) if (present & (1u << cc)) {
            const uint64_t code19 = (pre << 2) | cc;                       // 2W bits
            const uint32_t keyF = (uint32_t)(code19 >> (2 * T));           // first P nt
            const uint64_t tailF = code19 & ((1ull << (2 * T)) - 1);       // last T nt, MSB-first
            k.f_key.push_back(keyF); k.f_tail_id.push_back((tailF << 32) | id);
            const uint32_t keyR = (uint32_t)(code19 & ((1ull << (2 * P)) - 1));   // last P nt
            const uint64_t head = code19 >> (2 * P);                       // first T nt, MSB-first
            uint64_t tailR = 0;                                            // reversed head (indexdb.cpp:1441-1444)
            for (uint32_t q = 0; q < T; q++) tailR = (tailR << 2) | ((head >> (2 * q)) & 3);
            k.r_key.push_back(keyR); k.r_tail_id.push_back((tailR << 32) | id);
          }
          id++;
          i = j;
        }
        k.n_ids = id;
      }
    });
    std::vector<uint64_t> id_base(nchunk + 1, 0), pos_base(nchunk + 1, 0), ent_base(nchunk + 1, 0);
    for (size_t c = 0; c < nchunk; c++) {
      id_base[c + 1] = id_base[c] + ch[c].n_ids; pos_base[c + 1] = pos_base[c] + ch[c].pos_arr.size() / 2; ent_base[c + 1] = ent_base[c] + ch[c].f_key.size();
    }
    if (id_base[nchunk] > 0xFFFFFFF0ull || pos_base[nchunk] > 0xFFFFFFF0ull || ent_base[nchunk] > 0xFFFFFFF0ull) { why = "part too large for the builder (reduce -m)"; return SMR_ERR_ARG; }
    std::vector<uint32_t> f_key(ent_base[nchunk]), r_key(ent_base[nchunk]); std::vector<uint64_t> f_tail_id(ent_base[nchunk]), r_tail_id(ent_base[nchunk]);
    ix.pos_off.assign((size_t)id_base[nchunk] + 1, 0);
    ix.pos_arr.resize((size_t)2 * pos_base[nchunk]);
    parallel_for(threads, nchunk, [&](size_t c0, size_t c1, uint32_t) {
      for (size_t c = c0; c < c1; c++) {
        const Chunk& k = ch[c];
        uint64_t po = pos_base[c];
        for (uint32_t g = 0; g < k.n_ids; g++) { po += k.pos_cnt[g]; ix.pos_off[(size_t)id_base[c] + g + 1] = (uint32_t)po; }
        if (!k.pos_arr.empty()) memcpy(ix.pos_arr.data() + 2 * pos_base[c], k.pos_arr.data(), k.pos_arr.size() * 4);
        const size_t e0 = (size_t)ent_base[c];
        for (size_t e = 0; e < k.f_key.size(); e++) {
          f_key[e0 + e] = k.f_key[e]; r_key[e0 + e] = k.r_key[e];
          f_tail_id[e0 + e] = k.f_tail_id[e] + id_base[c]; r_tail_id[e0 + e] = k.r_tail_id[e] + id_base[c];
        }
      }
    });
    ch.clear(); ch.shrink_to_fit();
    tm.lap("ids, positions, entries");
    keys.clear(); keys.shrink_to_fit(); last_nt.clear(); last_nt.shrink_to_fit();
    const uint32_t NK = 1u << L;
    ix.lookup.assign(NK, Lookup{0, NONE, NONE, 0, 0});
    // group R by key (counting sort), sort each group by tail
    size_t M = f_key.size();
    std::vector<size_t> rstart((size_t)NK + 1, 0), fstart((size_t)NK + 1, 0);
    {
      std::vector<std::atomic<uint32_t>> cr(NK), cf(NK);            // value-initialised = 0
      parallel_for(threads, M, [&](size_t lo, size_t hi, uint32_t) {
        for (size_t i = lo; i < hi; i++) { cr[r_key[i]].fetch_add(1, std::memory_order_relaxed); cf[f_key[i]].fetch_add(1, std::memory_order_relaxed); }
      });
      for (uint32_t k = 0; k < NK; k++) { rstart[k + 1] = rstart[k] + cr[k].load(std::memory_order_relaxed); fstart[k + 1] = fstart[k] + cf[k].load(std::memory_order_relaxed); }
      std::vector<uint64_t> rs(M);
      for (uint32_t k = 0; k < NK; k++) cr[k].store(0, std::memory_order_relaxed);
      parallel_for(threads, M, [&](size_t lo, size_t hi, uint32_t) {          // the order inside a key does not matter: every key is sorted next
        for (size_t i = lo; i < hi; i++) rs[rstart[r_key[i]] + cr[r_key[i]].fetch_add(1, std::memory_order_relaxed)] = r_tail_id[i];
      });
      r_tail_id.swap(rs);
    }
    std::vector<uint64_t>& rsorted = r_tail_id;
    r_key.clear(); r_key.shrink_to_fit();
    parallel_for(threads, NK, [&](size_t lo, size_t hi, uint32_t) { for (size_t k = lo; k < hi; k++) std::sort(rsorted.begin() + rstart[k], rsorted.begin() + rstart[k + 1]); });
    tm.lap("group reverse entries");
    // emit tries: size of every mini-trie (parallel) -> offsets -> layout (parallel); one function shared with the device builder
    const int burst_depth = (int)(W - P - 3);
    std::vector<uint64_t> toff((size_t)2 * NK + 1, 0);
    std::vector<uint32_t> tnodes((size_t)2 * NK, 0), tbuckets((size_t)2 * NK, 0);
    std::atomic<int> bad{TRIE_OK};
    parallel_for(threads, NK, [&](size_t lo, size_t hi, uint32_t) {
      for (size_t k = lo; k < hi; k++) for (int j = 0; j < 2; j++) {
        const size_t cnt = j == 0 ? fstart[k + 1] - fstart[k] : rstart[k + 1] - rstart[k];
        if (!cnt) continue;
        int st = TRIE_OK;
        toff[2 * k + j + 1] = minitrie_layout<false>(j == 0 ? f_tail_id.data() + fstart[k] : rsorted.data() + rstart[k], (uint32_t)cnt, (int)T, burst_depth,
                                                     nullptr, &tnodes[2 * k + j], &tbuckets[2 * k + j], &st);
        if (st != TRIE_OK) bad = st;
      }
    });
    if (bad != TRIE_OK) { why = bad == TRIE_ERR_BUCKET ? "bucket with more than 255 entries" : "mini-trie larger than 2^22 words"; return SMR_ERR_IO; }
    for (size_t i = 0; i < (size_t)2 * NK; i++) { toff[i + 1] += toff[i]; ix.n_nodes += tnodes[i]; ix.n_buckets += tbuckets[i]; }
    if (toff.back() > 0xFFFFFFF0ull) { why = "trie arena exceeds 2^32 words"; return SMR_ERR_IO; }
    ix.n_entries += 2 * M;
    ix.trie.resize(toff.back());
    parallel_for(threads, NK, [&](size_t lo, size_t hi, uint32_t) {
      for (size_t k = lo; k < hi; k++) {
        const size_t nf = fstart[k + 1] - fstart[k], nr = rstart[k + 1] - rstart[k];
        ix.lookup[k].count = (uint32_t)(nf + nr);             // only tested as `count > minoccur`
        for (int j = 0; j < 2; j++) {
          const size_t cnt = j == 0 ? nf : nr;
          if (!cnt) continue;
          int st = TRIE_OK;
          const uint32_t root = (uint32_t)toff[2 * k + j];
          const uint32_t words = minitrie_layout<true>(j == 0 ? f_tail_id.data() + fstart[k] : rsorted.data() + rstart[k], (uint32_t)cnt, (int)T, burst_depth,
                                                       ix.trie.data() + root, nullptr, nullptr, &st);
          if (j == 0) { ix.lookup[k].rootF = root; ix.lookup[k].wordsF = words; }
          else { ix.lookup[k].rootR = root; ix.lookup[k].wordsR = words; }
        }
      }
    });
    tm.lap("mini-tries");
  return SMR_OK;
}

// Everything of the build that is not per-occurrence work: FASTA parsing, statistics, the split into parts, the reference
// sequences for SW; the occurrences -> (lookup, tries, positions) step of every part is done by `fn` (host: ib_part_host below;
// device: smr_index_build_gpu in smr_engine.hip).
int smr_index_build_with(const char* ref_fasta, uint32_t L, double max_mb, uint32_t max_pos, uint32_t threads, smr_ibuild_part_fn fn, void* user,
                         smr_index** parts_out, uint32_t cap_parts, uint32_t* n_parts_out, char* err, size_t errcap) {
  if (!ref_fasta || !parts_out || !n_parts_out || !fn) return SMR_ERR_ARG;
  if (L < 8 || L > 18 || (L & 1)) { set_err(err, errcap, "builder supports even seed lengths 8..18"); return SMR_ERR_ARG; }
  if (threads == 0) threads = smr::host_threads();
  std::vector<uint8_t> file;
  if (!slurp(ref_fasta, file)) { set_err(err, errcap, std::string("cannot read ") + ref_fasta); return SMR_ERR_IO; }
  StageTimer tmd;
  std::vector<SeqRec> recs; std::vector<uint8_t> raw; std::string why;
  if (!parse_fasta(file, recs, raw, why)) { set_err(err, errcap, why); return SMR_ERR_IO; }
  tmd.lap("read + parse FASTA");
  const uint32_t W = L + 1;
  // STEP 1 statistics (indexdb.cpp:1198-1268)
  double bgc[4] = {0, 0, 0, 0}; uint64_t full_len = 0;
  for (auto& r : recs) {
    if (r.len < W) { set_err(err, errcap, "at least one sequence is shorter than the seed length " + std::to_string(W)); return SMR_ERR_IO; }
    full_len += r.len;
  }
  {                                                          // (all cores: one thread counting 140 M letters was a third of a device build)
    std::vector<std::array<uint64_t, 4>> cnt(threads, std::array<uint64_t, 4>{0, 0, 0, 0});
    parallel_for(threads, recs.size(), [&](size_t lo, size_t hi, uint32_t t) {
      uint64_t c4[4] = {0, 0, 0, 0};
      for (size_t q = lo; q < hi; q++) { const SeqRec& r = recs[q]; for (uint32_t k = 0; k < r.len; k++) { const int c = raw[r.seq_begin + k]; if (c != 'N') c4[nt_index(c)]++; } }
      for (int k = 0; k < 4; k++) cnt[t][k] += c4[k];
    });
    for (auto& c4 : cnt) for (int k = 0; k < 4; k++) bgc[k] += (double)c4[k];
  }
  double tot = bgc[0] + bgc[1] + bgc[2] + bgc[3];
  // part split (indexdb.cpp:1384-1420)
  struct PartRange { size_t s0, s1; };
  std::vector<PartRange> pr; std::vector<PartStats> pstats;
  {
    size_t i = 0;
    while (i < recs.size()) {
      double idx_size = 0; size_t s0 = i; PartStats ps; ps.start_part = recs[i].file_start; bool any = false;
      while (i < recs.size()) {
        double est = (double)(recs[i].len - W + 1) * 9.5e-6;
        if (est > max_mb) { i++; continue; }              // sequence alone too large: skipped by the reference
        if (idx_size + est > max_mb) break;
        idx_size += est; ps.numseq_part++; ps.seq_part_size = recs[i].file_end - ps.start_part; any = true; i++;
      }
      if (!any) break;
      pr.push_back({s0, i}); pstats.push_back(ps);
    }
  }
  if (pr.empty()) { set_err(err, errcap, "no index could be created with this memory limit"); return SMR_ERR_ARG; }
  if (pr.size() > cap_parts) { set_err(err, errcap, "too many index parts for the caller's array"); return SMR_ERR_ARG; }
  std::vector<std::pair<std::string, uint32_t>> sq;
  for (auto& r : recs) sq.emplace_back(r.id, r.len);

  for (size_t pi = 0; pi < pr.size(); pi++) {
    auto ix = new smr_index();
    ix->lnwin = L; ix->part = (uint32_t)pi; ix->n_parts = (uint32_t)pr.size();
    for (int k = 0; k < 4; k++) ix->bg[k] = bgc[k] / tot;
    ix->full_len = full_len; ix->numseq = recs.size(); ix->filesize = file.size(); ix->parts = pstats; ix->sq_header = sq;
    // sequences of this part (those skipped for size are not part of it); seq number = rank within the part
    std::vector<size_t> members;
    for (size_t s = pr[pi].s0; s < pr[pi].s1; s++) if ((double)(recs[s].len - W + 1) * 9.5e-6 <= max_mb) members.push_back(s);
    // reference sequences for SW (nt_table) and nt codes of the index alphabet (map_nt) of the members, concatenated
    std::vector<uint64_t> seq_off(1, 0);
    for (size_t s : members) seq_off.push_back(seq_off.back() + recs[s].len);
    ix->ref_off = seq_off;
    ix->ref_seq.resize(seq_off.back());
    std::vector<uint8_t> codes(seq_off.back());
    parallel_for(threads, members.size(), [&](size_t lo, size_t hi, uint32_t) {
      for (size_t m = lo; m < hi; m++) {
        const SeqRec& r = recs[members[m]];
        const uint8_t* src = raw.data() + r.seq_begin;
        uint8_t* d1 = ix->ref_seq.data() + seq_off[m]; uint8_t* d2 = codes.data() + seq_off[m];
        for (uint32_t k = 0; k < r.len; k++) { d1[k] = nt_sw(src[k]); d2[k] = nt_index(src[k]); }
      }
    });
    tmd.lap("statistics, refs, codes");
    smr::IBuildInput in; in.codes = codes.data(); in.seq_off = seq_off.data(); in.n_seqs = (uint32_t)members.size(); in.L = L; in.max_pos = max_pos; in.threads = threads;
    const int rc = fn(user, in, *ix, why);
    if (rc != SMR_OK) { delete ix; set_err(err, errcap, why); return rc; }
    smr_build_lkc(*ix);                                     // (the pigeonhole layout of the tries is built on the device by smr_index_upload)
    parts_out[pi] = ix;
  }
  *n_parts_out = (uint32_t)pr.size();
  return SMR_OK;
}

extern "C" int smr_index_build(const char* ref_fasta, uint32_t L, double max_mb, uint32_t max_pos, uint32_t threads,
                               smr_index** parts_out, uint32_t cap_parts, uint32_t* n_parts_out, char* err, size_t errcap) {
  return smr_index_build_with(ref_fasta, L, max_mb, max_pos, threads, ib_part_host, nullptr, parts_out, cap_parts, n_parts_out, err, errcap);
}

// =================================================================================================
// Writer of the reference's on-disk format (so `sortmerna` itself and the test oracle can consume our index)
// =================================================================================================
namespace {
void write_bfs(const smr_index& ix, uint32_t root, std::vector<uint8_t>& out, uint32_t& mem_size) {
  // BFS over the compact arena; mem_size = nodes*64 + bucket bytes (indexdb.cpp:730-748)
  std::vector<uint32_t> q{0};
  mem_size = 0;
  auto put_flags = [&](uint32_t rel) { for (int k = 0; k < 4; k++) out.push_back((uint8_t)(ix.trie[root + rel + k] >> ELEM_FLAG_SHIFT)); };
  put_flags(0);
  for (size_t h = 0; h < q.size(); h++) {
    mem_size += 64;
    for (int k = 0; k < 4; k++) {
      uint32_t e = ix.trie[root + q[h] + k]; uint32_t fl = e >> ELEM_FLAG_SHIFT;
      if (fl == 1) { uint32_t rel = e & ELEM_OFF_MASK; put_flags(rel); q.push_back(rel); }
      else if (fl == 2) {
        uint32_t n = (e >> ELEM_NENT_SHIFT) & 0xFF, rel = e & ELEM_OFF_MASK, bytes = n * 8;
        size_t old = out.size(); out.resize(old + 4 + bytes);
        memcpy(out.data() + old, &bytes, 4); memcpy(out.data() + old + 4, &ix.trie[root + rel], bytes);
        mem_size += bytes;
      }
    }
  }
}
}  // namespace

extern "C" int smr_index_write_files(const smr_index* const* parts, uint32_t n_parts, const char* ref_fasta, const char* prefix, char* err, size_t errcap) {
  if (!parts || !n_parts || !prefix || !ref_fasta) return SMR_ERR_ARG;
  for (uint32_t p = 0; p < n_parts; p++) {
    const smr_index& ix = *parts[p];
    std::string ps = std::to_string(p);
    { std::ofstream f(std::string(prefix) + ".kmer_" + ps + ".dat", std::ios::binary);
      if (!f) { set_err(err, errcap, "cannot write kmer file"); return SMR_ERR_IO; }
      for (auto& l : ix.lookup) f.write((const char*)&l.count, 4); }
    { std::ofstream f(std::string(prefix) + ".bursttrie_" + ps + ".dat", std::ios::binary);
      if (!f) { set_err(err, errcap, "cannot write bursttrie file"); return SMR_ERR_IO; }
      std::vector<uint8_t> sf, sr;
      for (auto& l : ix.lookup) {
        uint32_t sz[2] = {0, 0}; sf.clear(); sr.clear();
        if (l.rootF != NONE) write_bfs(ix, l.rootF, sf, sz[0]);
        if (l.rootR != NONE) write_bfs(ix, l.rootR, sr, sz[1]);
        f.write((const char*)sz, 8);
        if (l.count != 0) { f.write((const char*)sf.data(), (std::streamsize)sf.size()); f.write((const char*)sr.data(), (std::streamsize)sr.size()); }
      } }
    { std::ofstream f(std::string(prefix) + ".pos_" + ps + ".dat", std::ios::binary);
      if (!f) { set_err(err, errcap, "cannot write pos file"); return SMR_ERR_IO; }
      uint32_t nid = ix.n_ids(); f.write((const char*)&nid, 4);
      for (uint32_t i = 0; i < nid; i++) {
        uint32_t sz = ix.pos_off[i + 1] - ix.pos_off[i];
        f.write((const char*)&sz, 4);
        f.write((const char*)(ix.pos_arr.data() + (size_t)ix.pos_off[i] * 2), (std::streamsize)sz * 8);
      } }
  }
  const smr_index& ix0 = *parts[0];
  std::ofstream st(std::string(prefix) + ".stats", std::ios::binary);
  if (!st) { set_err(err, errcap, "cannot write stats file"); return SMR_ERR_IO; }
  uint64_t fs = ix0.filesize; st.write((const char*)&fs, 8);
  std::string fn(ref_fasta); uint32_t fl = (uint32_t)fn.size() + 1; st.write((const char*)&fl, 4); st.write(fn.c_str(), fl);
  st.write((const char*)ix0.bg, 32); st.write((const char*)&ix0.full_len, 8); st.write((const char*)&ix0.lnwin, 4);
  st.write((const char*)&ix0.numseq, 8);
  uint16_t np = (uint16_t)n_parts; st.write((const char*)&np, 2);
  for (uint32_t p = 0; p < n_parts; p++) {
    uint8_t raw[24] = {0};
    memcpy(raw, &ix0.parts[p].start_part, 8); memcpy(raw + 8, &ix0.parts[p].seq_part_size, 8); memcpy(raw + 16, &ix0.parts[p].numseq_part, 4);
    st.write((const char*)raw, 24);
  }
  uint32_t nsq = (uint32_t)ix0.sq_header.size(); st.write((const char*)&nsq, 4);
  for (auto& s : ix0.sq_header) {
    uint32_t li = (uint32_t)s.first.size(); st.write((const char*)&li, 4); st.write(s.first.data(), li); st.write((const char*)&s.second, 4);
  }
  return SMR_OK;
}


// ------------------------------------------------------------------------------------------------
// Flat index cache.  The reference's four files per part store no stream lengths (index.cpp:176-316): loading them means one sequential
// walk over GBs to find where every mini-trie begins, then a parse.  smr_index_save writes the HOST layout of a loaded / built part as it
// lies in memory -- header, then every array 4096-byte aligned -- and smr_index_load_flat maps the file and copies the arrays with all
// cores: the index is ready at memory speed.  `stamp` is the caller's key (e.g. size and mtime of the reference FASTA and of the
// reference-format files): load fails with SMR_ERR_STATE when the file's stamp is another, and the caller falls back to the slow path.
// ------------------------------------------------------------------------------------------------
namespace {
struct FlatHeader {
  char magic[8];                                          // "SMRFLAT1"
  uint32_t version, lnwin, part, n_parts;
  uint64_t stamp;
  uint64_t n_nodes, n_buckets, n_entries, full_len, numseq, filesize;
  double bg[4];
  uint64_t n_lookup, n_trie, n_pos_off, n_pos_arr, n_ref_seq, n_ref_off, n_lkc, n_parts_stats, n_sq_bytes;      // element counts (sq: bytes)
  uint64_t off[9];                                        // byte offsets of the sections, in this order
  uint64_t total_bytes;
};
const uint64_t FLAT_ALIGN = 4096;

void par_copy(void* dst, const void* src, size_t n, uint32_t threads) {
  const size_t CH = (size_t)8 << 20;
  const size_t nch = (n + CH - 1) / CH;
  if (nch <= 1) { if (n) memcpy(dst, src, n); return; }
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  for (uint32_t t = 0; t < std::min<size_t>(threads, nch); t++)
    th.emplace_back([&] { for (size_t c; (c = next.fetch_add(1)) < nch;) memcpy((char*)dst + c * CH, (const char*)src + c * CH, std::min(CH, n - c * CH)); });
  for (auto& x : th) x.join();
}
// f(first, last) over [0, n) in chunks, by `threads` threads
template <class F> void par_for(size_t n, uint32_t threads, F f) {
  const size_t CH = (size_t)1 << 20;
  const size_t nch = (n + CH - 1) / CH;
  if (nch <= 1) { if (n) f((size_t)0, n); return; }
  std::atomic<size_t> next{0};
  std::vector<std::thread> th;
  for (uint32_t t = 0; t < std::min<size_t>(threads, nch); t++)
    th.emplace_back([&] { for (size_t c; (c = next.fetch_add(1)) < nch;) f(c * CH, std::min(n, (c + 1) * CH)); });
  for (auto& x : th) x.join();
}
}  // namespace

extern "C" int smr_index_save(const smr_index* ix, const char* path, uint64_t stamp, char* err, size_t errcap) {
  if (!ix || !path) { set_err(err, errcap, "smr_index_save: null argument"); return SMR_ERR_ARG; }
  std::string sq;
  for (auto& h : ix->sq_header) { const uint32_t l = (uint32_t)h.first.size(); sq.append((const char*)&l, 4); sq.append(h.first); sq.append((const char*)&h.second, 4); }
  FlatHeader H; memset(&H, 0, sizeof H);
  memcpy(H.magic, "SMRFLAT1", 8);
  H.version = 1; H.lnwin = ix->lnwin; H.part = ix->part; H.n_parts = ix->n_parts; H.stamp = stamp;
  H.n_nodes = ix->n_nodes; H.n_buckets = ix->n_buckets; H.n_entries = ix->n_entries; H.full_len = ix->full_len; H.numseq = ix->numseq; H.filesize = ix->filesize;
  for (int q = 0; q < 4; q++) H.bg[q] = ix->bg[q];
  H.n_lookup = ix->lookup.size(); H.n_trie = ix->trie.size(); H.n_pos_off = ix->pos_off.size(); H.n_pos_arr = ix->pos_arr.size();
  H.n_ref_seq = ix->ref_seq.size(); H.n_ref_off = ix->ref_off.size(); H.n_lkc = ix->lkc.size(); H.n_parts_stats = ix->parts.size(); H.n_sq_bytes = sq.size();
  const void* src[9] = {ix->lookup.data(), ix->trie.data(), ix->pos_off.data(), ix->pos_arr.data(), ix->ref_seq.data(), ix->ref_off.data(), ix->lkc.data(), ix->parts.data(), sq.data()};
  const uint64_t bytes[9] = {H.n_lookup * sizeof(Lookup), H.n_trie * 4, H.n_pos_off * 4, H.n_pos_arr * 4, H.n_ref_seq, H.n_ref_off * 8, H.n_lkc * 4, H.n_parts_stats * sizeof(PartStats), H.n_sq_bytes};
  uint64_t o = FLAT_ALIGN;
  for (int q = 0; q < 9; q++) { H.off[q] = o; o = (o + bytes[q] + FLAT_ALIGN - 1) & ~(FLAT_ALIGN - 1); }
  H.total_bytes = o;
  // (a name of this process's own: two processes saving the same part must not write into each other's file; the blocks are allocated before
  // the mapping is written -- a full disk is an error return here, not a SIGBUS in par_copy)
  const std::string tmp = std::string(path) + ".tmp." + std::to_string((long long)getpid());
  int fd = open(tmp.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0644);
  if (fd < 0) { set_err(err, errcap, "smr_index_save: cannot create " + tmp); return SMR_ERR_IO; }
  bool ok = posix_fallocate(fd, 0, (off_t)o) == 0;
  void* m = ok ? mmap(nullptr, o, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
  ok = ok && m != MAP_FAILED;
  if (ok) {
    const uint32_t threads = std::min(32u, smr::host_threads());
    memcpy(m, &H, sizeof H);
    for (int q = 0; q < 9; q++) par_copy((char*)m + H.off[q], src[q], bytes[q], threads);
    munmap(m, o);
  }
  close(fd);
  if (!ok || rename(tmp.c_str(), path) != 0) { unlink(tmp.c_str()); set_err(err, errcap, std::string("smr_index_save: cannot write ") + path); return SMR_ERR_IO; }
  return SMR_OK;
}

extern "C" int smr_index_load_flat(const char* path, uint64_t stamp, smr_index** out, char* err, size_t errcap) {
  if (!path || !out) { set_err(err, errcap, "smr_index_load_flat: null argument"); return SMR_ERR_ARG; }
  *out = nullptr;
  StageTimer tm;
  int fd = open(path, O_RDONLY);
  if (fd < 0) { set_err(err, errcap, std::string("no flat index at ") + path); return SMR_ERR_IO; }
  struct stat st;
  if (fstat(fd, &st) != 0 || (uint64_t)st.st_size < sizeof(FlatHeader)) { close(fd); set_err(err, errcap, std::string(path) + ": not a flat index"); return SMR_ERR_IO; }
  const size_t n = (size_t)st.st_size;
  void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { set_err(err, errcap, std::string("cannot map ") + path); return SMR_ERR_IO; }
  struct Unmap { void* p; size_t n; ~Unmap() { munmap(p, n); } } um{m, n};
  FlatHeader H; memcpy(&H, m, sizeof H);
  const uint64_t bytes[9] = {H.n_lookup * sizeof(Lookup), H.n_trie * 4, H.n_pos_off * 4, H.n_pos_arr * 4, H.n_ref_seq, H.n_ref_off * 8, H.n_lkc * 4, H.n_parts_stats * sizeof(PartStats), H.n_sq_bytes};
  bool ok = memcmp(H.magic, "SMRFLAT1", 8) == 0 && H.version == 1 && H.total_bytes == n && H.lnwin >= 8 && H.lnwin <= 20 && H.n_lookup == (1ull << H.lnwin);
  {
    // (element counts first: a count like 2^62 + k wraps to a small byte count, passes the bounds test below and throws in resize)
    const uint64_t cnt[9] = {H.n_lookup, H.n_trie, H.n_pos_off, H.n_pos_arr, H.n_ref_seq, H.n_ref_off, H.n_lkc, H.n_parts_stats, H.n_sq_bytes};
    const uint64_t esz[9] = {sizeof(Lookup), 4, 4, 4, 1, 8, 4, sizeof(PartStats), 1};
    for (int q = 0; q < 9 && ok; q++) ok = cnt[q] <= n / esz[q];
  }
  for (int q = 0; q < 9 && ok; q++) ok = H.off[q] % FLAT_ALIGN == 0 && H.off[q] <= n && bytes[q] <= n - H.off[q];
  ok = ok && H.n_lkc == H.n_lookup && H.n_pos_off >= 1 && H.n_ref_off >= 1;
  if (!ok) { set_err(err, errcap, std::string(path) + ": damaged or foreign flat index"); return SMR_ERR_IO; }
  if (H.stamp != stamp) { set_err(err, errcap, std::string(path) + ": the flat index was written for other reference files (stamp)"); return SMR_ERR_STATE; }
  (void)madvise(m, n, MADV_WILLNEED);
  std::unique_ptr<smr_index> ix(new smr_index);
  ix->lnwin = H.lnwin; ix->part = H.part; ix->n_parts = H.n_parts;
  ix->n_nodes = H.n_nodes; ix->n_buckets = H.n_buckets; ix->n_entries = H.n_entries; ix->full_len = H.full_len; ix->numseq = H.numseq; ix->filesize = H.filesize;
  for (int q = 0; q < 4; q++) ix->bg[q] = H.bg[q];
  const char* base = (const char*)m;
  try {
    // the arrays are sized side by side (a vector zero-fills what it is resized to: one thread per array, 2 MB pages), then filled by all cores
    // (an exception thrown inside a thread would end the process: every lambda reports through `failed` instead)
    std::vector<std::thread> th;
    std::atomic<bool> failed{false};
#define SIZED(...) [&] { try { __VA_ARGS__ } catch (...) { failed = true; } }
    th.emplace_back(SIZED(reserve_huge(ix->trie, H.n_trie); ix->trie.resize(H.n_trie);));
    th.emplace_back(SIZED(reserve_huge(ix->pos_arr, H.n_pos_arr); ix->pos_arr.resize(H.n_pos_arr);));
    th.emplace_back(SIZED(reserve_huge(ix->pos_off, H.n_pos_off); ix->pos_off.resize(H.n_pos_off);));
    th.emplace_back(SIZED(reserve_huge(ix->ref_seq, H.n_ref_seq); ix->ref_seq.resize(H.n_ref_seq); ix->ref_off.resize(H.n_ref_off); ix->lookup.resize(H.n_lookup); ix->lkc.resize(H.n_lkc); ix->parts.resize(H.n_parts_stats);));
#undef SIZED
    for (auto& x : th) x.join();
    if (failed) { set_err(err, errcap, std::string(path) + ": the arrays of the flat index could not be sized (out of memory?)"); return SMR_ERR_IO; }
  } catch (const std::exception& e) { set_err(err, errcap, std::string("smr_index_load_flat: ") + e.what()); return SMR_ERR_IO; }
  tm.lap("load flat: arrays sized");
  const uint32_t threads = std::min(64u, smr::host_threads());
  void* dst[8] = {ix->lookup.data(), ix->trie.data(), ix->pos_off.data(), ix->pos_arr.data(), ix->ref_seq.data(), ix->ref_off.data(), ix->lkc.data(), ix->parts.data()};
  for (int q = 0; q < 8; q++) par_copy(dst[q], base + H.off[q], bytes[q], threads);
  const char* sp = base + H.off[8]; const char* se = sp + bytes[8];
  while (sp + 4 <= se) {
    uint32_t l; memcpy(&l, sp, 4); sp += 4;
    if (sp + l + 4 > se) { set_err(err, errcap, std::string(path) + ": damaged sequence table"); return SMR_ERR_IO; }
    uint32_t ln; memcpy(&ln, sp + l, 4);
    ix->sq_header.emplace_back(std::string(sp, l), ln);
    sp += l + 4;
  }
  // what the other loaders guarantee, checked here too: offsets inside the arrays
  if (ix->pos_off.empty() || ix->pos_off.back() * 2ull != ix->pos_arr.size() || ix->ref_off.empty() || ix->ref_off.back() != ix->ref_seq.size()) {
    set_err(err, errcap, std::string(path) + ": inconsistent flat index"); return SMR_ERR_IO;
  }
  {
    // (no checksum in the file: the offsets the kernels follow are range-checked instead -- monotone position and reference offsets, every
    // mini-trie inside the arena; a damaged file with a valid header must not reach the device)
    std::atomic<bool> bad{false};
    const uint64_t ntrie = ix->trie.size();
    par_for(ix->pos_off.size() - 1, threads, [&](size_t a, size_t b) { for (size_t i = a; i < b; i++) if (ix->pos_off[i] > ix->pos_off[i + 1]) { bad = true; return; } });
    par_for(ix->ref_off.size() - 1, threads, [&](size_t a, size_t b) { for (size_t i = a; i < b; i++) if (ix->ref_off[i] > ix->ref_off[i + 1]) { bad = true; return; } });
    par_for(ix->lookup.size(), threads, [&](size_t a, size_t b) {
      for (size_t i = a; i < b; i++) { const Lookup& l = ix->lookup[i]; if ((l.wordsF && (uint64_t)l.rootF + l.wordsF > ntrie) || (l.wordsR && (uint64_t)l.rootR + l.wordsR > ntrie)) { bad = true; return; } } });
    if (bad) { set_err(err, errcap, std::string(path) + ": offsets of the flat index point outside its arrays"); return SMR_ERR_IO; }
  }
  tm.lap("load flat: arrays copied");
  *out = ix.release();
  return SMR_OK;
}
