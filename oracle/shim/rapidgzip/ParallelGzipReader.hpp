// TEST SCAFFOLDING (oracle/_ref build): zlib-backed stand-in for rapidgzip::ParallelGzipReader
// (API used by /root/reference/src/sortmerna/readfeed.cpp:58-63,112,918,1138-1147).
#pragma once
#include <zlib.h>
#include <memory>
#include <string>
#include <filereader/Standard.hpp>
namespace rapidgzip {
template <typename T = void> class ParallelGzipReader {
  gzFile f;
public:
  ParallelGzipReader(std::unique_ptr<StandardFileReader> fr, std::size_t) { f = gzopen(fr->path.c_str(), "rb"); }
  ~ParallelGzipReader() { if (f) gzclose(f); }
  long long read(char* buf, std::size_t n) { return gzread(f, buf, (unsigned)n); }
  long long seek(long long off) { return gzseek(f, off, SEEK_SET); }
};
}
