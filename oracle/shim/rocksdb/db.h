// TEST SCAFFOLDING for building the unmodified reference (oracle/_ref): an in-memory stand-in
// for the RocksDB API surface used by /root/reference/src/sortmerna/kvdb.cpp:43-74.
// Extra (ours): when env SMR_KVDB_DUMP=<path> is set, the final key/value map is written to
// <path> on DB destruction as: u64 n, then n x (u64 klen, key, u64 vlen, value), keys sorted.
// The values are the reference's own Read::toBinString() bytes -> golden per-read records.
#pragma once
#include <string>
#include <map>
#include <mutex>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
namespace rocksdb {
enum CompressionType { kNoCompression, kZlibCompression, kXpressCompression };
struct Options { CompressionType compression = kNoCompression; bool create_if_missing = false; void IncreaseParallelism() {} };
struct WriteOptions {};
struct ReadOptions {};
struct Status { bool ok() const { return true; } };
class DB {
  std::map<std::string, std::string> m;
  std::mutex mx;
public:
  static Status Open(const Options&, const std::string&, DB** db) { *db = new DB(); return Status(); }
  Status Put(const WriteOptions&, const std::string& k, const std::string& v) { std::lock_guard<std::mutex> l(mx); m[k] = v; return Status(); }
  Status Get(const ReadOptions&, const std::string& k, std::string* v) { std::lock_guard<std::mutex> l(mx); auto it = m.find(k); if (it != m.end()) *v = it->second; return Status(); }
  ~DB() {
    const char* p = getenv("SMR_KVDB_DUMP");
    if (!p) return;
    FILE* f = fopen(p, "wb");
    if (!f) return;
    uint64_t n = m.size();
    fwrite(&n, 8, 1, f);
    for (auto& kv : m) {
      uint64_t kl = kv.first.size(), vl = kv.second.size();
      fwrite(&kl, 8, 1, f); fwrite(kv.first.data(), 1, kl, f);
      fwrite(&vl, 8, 1, f); fwrite(kv.second.data(), 1, vl, f);
    }
    fclose(f);
  }
};
}
