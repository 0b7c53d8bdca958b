// smr_host.hpp -- host-side data model of libsmr_hip (index part, read batch).  Internal header.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <string>
#include <thread>
#include <mutex>
#include <vector>

#include "../../include/smr_hip.h"

namespace smr {

// host threads of the loaders / builders / parsers: the machine's cores, or fewer when SMR_HOST_THREADS says so (several ranks of one job share a
// host: bench.py --gpus N gives every rank cores / N, so that eight ranks setting up at once do not run 8 x 256 threads)
uint32_t host_threads();

// 9-mer lookup entry.  Replaces struct kmer {trie_F, trie_R, count}  (include/indexdb.hpp:98-103):
// pointers become word offsets of the mini-trie root node in the trie arena (NONE if absent).
struct Lookup {
  uint32_t count;
  uint32_t rootF;
  uint32_t rootR;
  uint32_t wordsF;   // size of the forward / reverse mini-trie in arena words (multiple of 4), for LDS staging
  uint32_t wordsR;
};
static constexpr uint32_t NONE = 0xFFFFFFFFu;

// Mini burst trie arena (u32 words).  Replaces NodeElement[4] nodes + malloc'd buckets
// (include/indexdb.hpp:67-84).  A node is 4 words, one element per nucleotide A,C,G,T:
//   bits 31..30 flag   0 empty, 1 child node, 2 bucket
//   bits 29..22 nent   number of bucket entries (flag 2)
//   bits 21..0  off    word offset RELATIVE to the mini-trie's root node
// A bucket is nent x {u32 tail (2 bit/nt, first nt in the low bits), u32 id} -- the reference's
// 8-byte ENTRYSIZE entries unchanged (indexdb.hpp:57).
static constexpr uint32_t ELEM_FLAG_SHIFT = 30;
static constexpr uint32_t ELEM_NENT_SHIFT = 22;
static constexpr uint32_t ELEM_OFF_MASK = (1u << 22) - 1;
static constexpr uint32_t ELEM_NENT_MAX = 255;

// Second device layout of the mini-tries, used by the pigeonhole seed search (k_seed_pg): per mini-trie a flat block of its complete
// candidate strings (trie path + bucket tail, pw+1 chars, char j at bits 2j).
// A string within LEV(1) of a pattern P either agrees with P on its first h = pw/2 chars, or -- the one edit being among those -- on
// chars h..pw-1 with P[h..], P[h-1..] or P[h+1..] (smr_seed_pg.hpp), so only the entries under a handful of exact keys can match:
//   dirA  4^cA + 1 offsets into TA: strings whose first cA chars (first char most significant) are < key
//   dirB  4^cB + 1 offsets into TB: the same over chars h..h+cB-1
//   TA    n strings sorted (first char most significant)          TB    the same strings sorted by chars h..pw-1
//   RA    n x {DFS rank, id} in the order of TA                    RB    ... of TB
// (strings apart from {rank, id}: a search reads the strings of its ranges -- nearly all fail -- and {rank, id} of the accepted ones only)
// "DFS rank" = the entry's position in the reference's traversal order of the mini-trie (A<C<G<T, bucket order), which is the order the
// reference meets -- and de-duplicates -- the hits in.  cA = min(h, log4 n), cB = min(pw-h, log4 2n); a block with n <= PG_SCAN entries
// has no directories and only TA, RA in DFS order (the search looks at every entry).  Blocks are 16-byte aligned;
// root3[2k + d] = {block offset / 4 words, n | cA << 24 | cB << 28} for the forward / reverse mini-trie of key k.
static constexpr uint32_t PG_SCAN = 4;
__host__ __device__ inline void pg_chars(uint32_t n, uint32_t pw, uint32_t& cA, uint32_t& cB) {
  cA = cB = 0;
  if (n <= PG_SCAN) return;
  const uint32_t h = pw / 2;
  while (cA < h && (4u << (2 * cA)) <= n) cA++;                   // floor(log4 n)
  while (cB < pw - h && (4u << (2 * cB)) <= 2 * n) cB++;          // floor(log4 2n)
}
// chars from..from+cnt-1 of a 2-bit packed string (char j at bits 2j) as a number, first char most significant
__host__ __device__ inline uint32_t pg_key(uint32_t T, uint32_t from, uint32_t cnt) {
  uint32_t k = 0;
  for (uint32_t q = 0; q < cnt; q++) k = (k << 2) | ((T >> (2 * (from + q))) & 3u);
  return k;
}

// plain u32 buffer whose allocation does not zero-fill (the GBs of the pigeonhole arena are written exactly once, by many threads)
struct WordBuf {
  uint32_t* p = nullptr; size_t n = 0;
  WordBuf() = default;
  WordBuf(const WordBuf&) = delete;
  WordBuf& operator=(const WordBuf&) = delete;
  ~WordBuf() { free(p); }
  uint32_t* data() { return p; }
  const uint32_t* data() const { return p; }
  size_t size() const { return n; }
  bool resize_uninitialized(size_t m) { free(p); p = static_cast<uint32_t*>(malloc(m * 4 + 16)); n = p ? m : 0; return p != nullptr; }
};

struct PartStats {
  uint64_t start_part = 0, seq_part_size = 0;
  uint32_t numseq_part = 0;
};

}  // namespace smr

struct smr_index {
  uint32_t lnwin = 18;
  uint32_t part = 0, n_parts = 1;
  std::vector<smr::Lookup> lookup;      // 4^(L/2)
  std::vector<uint32_t> trie;           // arena
  std::vector<uint32_t> pos_off;        // n_ids + 1
  std::vector<uint32_t> pos_arr;        // 2 * n_pos : {pos, seq}   (struct seq_pos, indexdb.hpp:87-91)
  std::vector<uint8_t> ref_seq;         // 0..4 per nt (References::convert_fix, references.cpp:162-169)
  std::vector<uint64_t> ref_off;        // n_refs + 1
  uint64_t n_nodes = 0, n_buckets = 0, n_entries = 0;
  smr::WordBuf pg;                      // pigeonhole arena (smr_build_pigeonhole)
  std::vector<uint32_t> lkc;            // per key: min(count, 2^30 - 1) | forward mini-trie present << 30 | reverse present << 31 (what the window scan needs of `lookup`, in one word)
  std::vector<uint32_t> root3;          // 2 * 2 * 4^(L/2) words: {block offset / 4, n | cA << 24 | cB << 28} of the forward / reverse mini-trie of key k at [2k], [2k+1]
  std::mutex pg_mutex;                  // smr_build_pigeonhole runs once, whichever thread / context asks first (several smr_ctx may upload the same host index)
  // whole-DB statistics (.stats)
  double bg[4] = {0.25, 0.25, 0.25, 0.25};
  uint64_t full_len = 0, numseq = 0, filesize = 0;
  std::vector<smr::PartStats> parts;                          // all parts of the DB
  std::vector<std::pair<std::string, uint32_t>> sq_header;    // (id, len) of every sequence of the DB
  uint32_t n_ids() const { return pos_off.empty() ? 0 : (uint32_t)pos_off.size() - 1; }
  uint32_t n_refs() const { return ref_off.empty() ? 0 : (uint32_t)ref_off.size() - 1; }
};

// One index part as the builders see it: the member sequences in the index alphabet (indexdb.cpp map_nt), concatenated.
namespace smr {
struct IBuildInput { const uint8_t* codes; const uint64_t* seq_off; uint32_t n_seqs, L, max_pos, threads; };
}
// fills lookup / trie / pos_off / pos_arr / n_nodes / n_buckets / n_entries of ix
typedef int (*smr_ibuild_part_fn)(void* user, const smr::IBuildInput& in, smr_index& ix, std::string& why);
int smr_index_build_with(const char* ref_fasta, uint32_t L, double max_mb, uint32_t max_pos, uint32_t threads, smr_ibuild_part_fn fn, void* user,
                         smr_index** parts_out, uint32_t cap_parts, uint32_t* n_parts_out, char* err, size_t errcap);

// builds pg/root3 from trie/lookup on the HOST (idempotent); false + message when the part is too large for the block table.  smr_index_upload
// builds the same layout on the device (smr_pgbuild.hpp); this transform is its checker (smr_index_selfcheck, smr_index_check_device) and
// what SMR_PG_HOST=1 uploads instead
bool smr_build_pigeonhole(smr_index& ix, uint32_t threads, std::string& why);
void smr_build_lkc(smr_index& ix);

// Packed read batch.  Record i = ceil(len/16) words of 2-bit codes (nt k in bits 2*(k%16) of word k/16)
// followed by ceil(len/32) words of ambiguity mask (bit k%32 of word k/32 set when the input letter
// was not ACGTU: Read::seqToIntStr stores 0 there and remembers the position, read.cpp:334-347).
struct smr_reads {
  uint32_t n = 0;
  std::vector<uint32_t> words;
  std::vector<uint64_t> rec_off;   // n + 1, in words
  std::vector<uint32_t> len;       // n
  uint64_t total_len = 0;
  uint32_t min_len = 0, max_len = 0;
  // smr_reads_load_fastx_text only: the file text (mapping or inflated copy) and where every record starts in it
  std::shared_ptr<void> text_owner;
  const char* text = nullptr; size_t text_n = 0; bool fastq = false;
  std::vector<uint64_t> hdr_off, seq_off;      // offset of the header line / of the first sequence line
};
