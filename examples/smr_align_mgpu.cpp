// smr_align_mgpu.cpp -- the C++17 multi-GPU host over the C ABI of libsmr_hip (include/smr_hip.h): one process, one host thread
// and one smr_ctx per GPU, reads sharded by record range, every GPU holds a full index replica (uploaded from ONE host copy of
// the index), no collective on the data path.  What the reference's semantics need across the shards are two tiny reductions,
// and both are RCCL all-reduces over xGMI issued from this C++ host:
//   C1 (before aligning)  sum of (reads, letters) over the shards -> the same minimal_score on every GPU
//                         (Readstats is built from the whole read file, main.cpp:77-78; refstats.cpp:247-265)
//   C2 (after aligning)   element-wise sum of the Readstats counter block, IN PLACE on the device memory that
//                         smr_counters_device hands out (readstats.hpp:77-85)
// It replaces the reference's thread fan-out over one read file (processor.cpp:248-256, split of the file into one range
// per thread readfeed.cpp:1253-1277) by a fan-out over GPUs.
//
// A rank STREAMS its shard (readfeed.cpp:776-873 streams any file size): chunks of --chunk-reads reads go through three recycled batch
// slots of the engine -- while chunk c is aligned on the GPU, a second host thread uploads chunk c+1 and a third one serialises the
// records (and report rows) of chunk c-1 from the host copy smr_results_fetch made of it.  Any number of chunks; the Readstats counters
// of the chunks are summed on the device (smr_counters_accumulate) into the block that RCCL then all-reduces.
//
// Output:  <out>/records.bin (same format as examples/smr_align.cpp: KVDB key "0_<global read number>" / Read::toBinString value,
//          shards concatenated in rank order), <out>/summary.txt (the reduced counters), one "[timing]" line per stage, and with
//          --fastx / --other / --blast / --sam the reference's report files (aligned.fq, other.fq, aligned.blast, aligned.sam): every rank
//          writes its shard's files into <out>/rank<r>/, rank order concatenation = Report::merge (report.cpp:56-97).
// Build:   hipcc -std=c++17 -O2 examples/smr_align_mgpu.cpp -Iinclude -Lsortmerna_amd/lib -lsmr_hip -lrccl -Wl,-rpath,$PWD/sortmerna_amd/lib -o smr_align_mgpu
// Options: --ref DB.fasta [--idx PREFIX] --gumbel LAMBDA K  [--ref ...]  --reads READS[.gz]  --out DIR  [--idx-flat DIR]
//          --gpus N            ranks (default: all visible devices)
//          --devices a,b,..    device of every rank (default 0,1,..,N-1)
//          --reduce rccl|host  how C1 / C2 are reduced; `host` (a mutex-protected sum between the rank threads) exists for dry
//                              runs of the N-rank path on fewer GPUs than ranks -- RCCL needs one device per rank
//          --chunk-reads M     per rank: chunks of M reads (0 = 2 M), any number of them
//          --fastx --other --blast '1 [cigar] [qcov] [qstrand]' --sam   report files (BLAST tabular, SAM without @SQ)
//          -num_alignments N, -no-best, -e EVALUE as in smr_align
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "smr_hip.h"

namespace {
[[noreturn]] void die(const std::string& m) { fprintf(stderr, "ERROR: %s\n", m.c_str()); exit(EXIT_FAILURE); }
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Db { std::string fasta, idx_prefix; double lambda = 0, K = 0; bool has_gumbel = false; std::vector<smr_index*> parts; bool from_flat = false; };

// the key of a flat index cache: size and mtime of the reference FASTA and of the first reference-format index file (when there is one)
uint64_t file_stamp(const std::string& fasta, const std::string& idx_prefix) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](uint64_t v) { for (int q = 0; q < 8; q++) { h ^= (v >> (8 * q)) & 0xFF; h *= 1099511628211ull; } };
  struct stat st;
  if (stat(fasta.c_str(), &st) == 0) { mix((uint64_t)st.st_size); mix((uint64_t)st.st_mtime); }
  if (!idx_prefix.empty() && stat((idx_prefix + ".kmer_0.dat").c_str(), &st) == 0) { mix((uint64_t)st.st_size); mix((uint64_t)st.st_mtime); }
  return h;
}

// contiguous record range of a rank (sortmerna_amd/shard.py: shard_range)
void shard_range(uint64_t n, int rank, int world, uint64_t& first, uint64_t& count) {
  const uint64_t base = n / world, rem = n % world;
  first = rank * base + std::min<uint64_t>(rank, rem);
  count = base + ((uint64_t)rank < rem ? 1 : 0);
}

struct Barrier {                       // reusable barrier for the rank threads (host reduction and stage timing)
  std::mutex m; std::condition_variable cv; int n, waiting = 0; uint64_t gen = 0;
  explicit Barrier(int n_) : n(n_) {}
  void wait() {
    std::unique_lock<std::mutex> l(m);
    const uint64_t g = gen;
    if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); } else cv.wait(l, [&] { return gen != g; });
  }
};

struct Shared {
  int world = 1;
  bool use_rccl = true;
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;
  Barrier* bar = nullptr;
  std::mutex m;
  std::vector<uint64_t> acc;           // host reduction scratch
};

// element-wise sum over the ranks of n u64 values that live in DEVICE memory at dptr (in place)
void all_reduce_sum_u64(Shared& S, int rank, void* dptr, size_t n, hipStream_t stream) {
  if (S.world == 1 && !S.use_rccl) return;
  if (S.use_rccl) {
    if (ncclAllReduce(dptr, dptr, n, ncclUint64, ncclSum, S.comms[rank], stream) != ncclSuccess) die("ncclAllReduce failed");
    if (hipStreamSynchronize(stream) != hipSuccess) die("hipStreamSynchronize after ncclAllReduce failed");
    return;
  }
  std::vector<uint64_t> h(n);
  if (hipMemcpy(h.data(), dptr, n * 8, hipMemcpyDeviceToHost) != hipSuccess) die("hipMemcpy D2H (host reduction) failed");
  S.bar->wait();
  { std::lock_guard<std::mutex> l(S.m); if (S.acc.size() != n) S.acc.assign(n, 0); for (size_t i = 0; i < n; i++) S.acc[i] += h[i]; }
  S.bar->wait();
  h = S.acc;
  S.bar->wait();
  if (rank == 0) S.acc.clear();
  S.bar->wait();
  if (hipMemcpy(dptr, h.data(), n * 8, hipMemcpyHostToDevice) != hipSuccess) die("hipMemcpy H2D (host reduction) failed");
}

struct RankOut {
  std::vector<uint8_t> records;        // concatenated (u64 klen, key, u64 vlen, value) entries of the shard
  uint64_t n_records = 0;
  std::vector<uint64_t> counters;      // reduced: identical on every rank
  double t_upload = 0, t_align = 0, t_fetch = 0, t_write = 0;
  int sw_kernel = -1;                     // smr_sw_mode of the rank's context: 0 = 32-bit kernel only (the packed kernel failed its self-check or was switched off)
  uint32_t minimal_score0 = 0;
};
}  // namespace

int main(int argc, char** argv) {
  const double t_main = now_s();
  std::vector<Db> dbs;
  std::string reads_path, out_dir = ".", reduce = "auto", devlist, flat_dir;
  smr_params base; smr_params_default(&base);
  double evalue = 1.0;
  int world = 0;
  uint64_t chunk_reads = 0;
  smr_report_opts ro; memset(&ro, 0, sizeof ro);
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto val = [&]() -> std::string { if (i + 1 >= argc) die("missing value after " + a); return argv[++i]; };
    if (a == "-ref" || a == "--ref") { Db d; d.fasta = val(); dbs.push_back(d); }
    else if (a == "-idx" || a == "--idx") { if (dbs.empty()) die("--idx before --ref"); dbs.back().idx_prefix = val(); }
    else if (a == "-idx-flat" || a == "--idx-flat") flat_dir = val();
    else if (a == "-gumbel" || a == "--gumbel") { if (dbs.empty()) die("--gumbel before --ref"); dbs.back().lambda = atof(val().c_str()); dbs.back().K = atof(val().c_str()); dbs.back().has_gumbel = true; }
    else if (a == "-reads" || a == "--reads") reads_path = val();
    else if (a == "-out" || a == "--out") out_dir = val();
    else if (a == "-e") evalue = atof(val().c_str());
    else if (a == "-num_alignments" || a == "--num_alignments") base.num_alignments = (uint32_t)atoi(val().c_str());
    else if (a == "-no-best" || a == "--no-best") base.is_best = 0;
    else if (a == "--gpus" || a == "-gpus") world = atoi(val().c_str());
    else if (a == "--devices" || a == "-devices") devlist = val();
    else if (a == "--reduce" || a == "-reduce") reduce = val();
    else if (a == "--chunk-reads" || a == "-chunk-reads") chunk_reads = strtoull(val().c_str(), nullptr, 10);
    else if (a == "-fastx" || a == "--fastx") ro.fastx = 1;
    else if (a == "-other" || a == "--other") ro.other = 1;
    else if (a == "-blast" || a == "--blast") {            // "1 [cigar] [qcov] [qstrand]" = tabular (options.cpp opt_blast)
      const std::string v = val();
      if (v.empty() || v[0] != '1') die("-blast: '1 [cigar] [qcov] [qstrand]' (tabular)");
      ro.blast_tabular = 1;
      const std::string cols = v.size() > 2 ? v.substr(2) : "";
      if (cols.size() >= sizeof ro.blast_cols) die("-blast: too many columns");
      strcpy(ro.blast_cols, cols.c_str());
    }
    else if (a == "-sam" || a == "--sam") ro.sam = 1;
    else die("unknown option " + a);
  }
  if (dbs.empty() || reads_path.empty()) die("--ref and --reads are required");
  if (const char* why = smr_params_refused(&base)) die(std::string("these options are outside what libsmr_hip aligns (the reference accepts them): ") + why);
  for (auto& d : dbs) if (!d.has_gumbel) die("--gumbel LAMBDA K is required for every --ref (the reference computes them with its vendored ALP library, refstats.cpp:194-233; minimal_score depends on them)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) die("no HIP device (libsmr_hip has no CPU fallback)");
  if (world <= 0) world = ndev;
  Shared S; S.world = world;
  if (devlist.empty()) for (int r = 0; r < world; r++) S.devices.push_back(r % ndev);
  else { size_t p = 0; while (p <= devlist.size()) { size_t q = devlist.find(',', p); if (q == std::string::npos) q = devlist.size(); S.devices.push_back(atoi(devlist.substr(p, q - p).c_str())); p = q + 1; } }
  if ((int)S.devices.size() != world) die("--devices must list one device per rank");
  // auto: RCCL when there is more than one rank (a reduction over one rank is the identity, and making a communicator takes seconds)
  if (reduce == "auto") reduce = world > 1 ? "rccl" : "host";
  S.use_rccl = reduce == "rccl";
  if (!S.use_rccl && reduce != "host") die("--reduce auto|rccl|host");
  if (S.use_rccl) {
    for (int a = 0; a < world; a++) for (int b = a + 1; b < world; b++) if (S.devices[a] == S.devices[b]) die("RCCL needs one device per rank: use --reduce host for a dry run with shared devices");
    S.comms.resize(world);
  }
  // the communicators are made while the index and the reads are being loaded (RCCL's initialisation takes seconds and nothing needs it before
  // the first reduction)
  double t_rccl = 0;
  std::thread rccl_thread([&] {
    if (!S.use_rccl) return;
    const double t = now_s();
    if (ncclCommInitAll(S.comms.data(), world, S.devices.data()) != ncclSuccess) die("ncclCommInitAll failed");
    t_rccl = now_s() - t;
  });
  Barrier bar(world); S.bar = &bar;
  char err[512] = "";

  // ---- host side, once: the reads (parsed and packed by a thread of their own, which fans out over the cores) while the index parts are
  // loaded or built (one host copy shared by all ranks) ----
  const double t0 = now_s();
  smr_reads* all = nullptr;
  const bool want_reports = ro.fastx || ro.other || ro.blast_tabular || ro.sam;
  double t_reads = 0;
  char rerr[512] = "";
  int reads_rc = SMR_OK;
  std::thread reads_thread([&] {
    // (with report files the text of the reads file stays mapped: the writers copy headers / letters / qualities from it)
    reads_rc = want_reports ? smr_reads_load_fastx_text(reads_path.c_str(), 0, &all, rerr, sizeof rerr) : smr_reads_load_fastx_mt(reads_path.c_str(), 0, &all, rerr, sizeof rerr);
    t_reads = now_s() - t0;
  });
  auto flat_path = [&](const Db& d, uint32_t part) {
    const size_t sl = d.fasta.find_last_of('/');
    return flat_dir + "/" + (sl == std::string::npos ? d.fasta : d.fasta.substr(sl + 1)) + "." + std::to_string(part) + ".flat";
  };
  for (auto& d : dbs) {
    // --idx-flat DIR: the parts as flat files of the host layout (smr_index_save), keyed by size + mtime of the reference files; loaded at
    // memory speed when they are there and current, written after this run's slow load / build otherwise
    if (!flat_dir.empty()) {
      const uint64_t stamp = file_stamp(d.fasta, d.idx_prefix);
      smr_index* p0 = nullptr;
      if (smr_index_load_flat(flat_path(d, 0).c_str(), stamp, &p0, err, sizeof err) == SMR_OK) {
        smr_index_info info; smr_index_get_info(p0, &info);
        d.parts.push_back(p0);
        bool all_ok = true;
        for (uint32_t k = 1; k < info.n_parts && all_ok; k++) {
          smr_index* pk = nullptr;
          all_ok = smr_index_load_flat(flat_path(d, k).c_str(), stamp, &pk, err, sizeof err) == SMR_OK;
          if (all_ok) d.parts.push_back(pk);
        }
        if (all_ok) { d.from_flat = true; continue; }
        for (auto* ix : d.parts) smr_index_free(ix);
        d.parts.clear();
      }
    }
    if (!d.idx_prefix.empty()) {
      smr_index* p0 = nullptr;
      if (smr_index_load_files(d.idx_prefix.c_str(), 0, d.fasta.c_str(), &p0, err, sizeof err) != SMR_OK) die(err);
      smr_index_info info; smr_index_get_info(p0, &info);
      d.parts.push_back(p0);
      for (uint32_t k = 1; k < info.n_parts; k++) {
        smr_index* pk = nullptr;
        if (smr_index_load_files(d.idx_prefix.c_str(), k, d.fasta.c_str(), &pk, err, sizeof err) != SMR_OK) die(err);
        d.parts.push_back(pk);
      }
    } else {
      // no index files: build the index, on the first rank's GPU (sorting / ids / positions / mini-tries as device kernels: seconds for a
      // 140 Mnt DB, build_index of the reference takes minutes, indexdb.cpp:1119-2095); the host copy then serves every rank
      smr_ctx* bctx = nullptr;
      if (smr_create(S.devices[0], &bctx, err, sizeof err) != SMR_OK) die(err);
      smr_index* arr[256]; uint32_t np = 0;
      if (smr_index_build_gpu(bctx, d.fasta.c_str(), 18, 3072.0, 10000, arr, 256, &np, err, sizeof err) != SMR_OK) die(err);
      smr_destroy(bctx);
      d.parts.assign(arr, arr + np);
    }
  }
  const double t_index = now_s() - t0;
  // (the flat cache of what was loaded the slow way is written beside the alignment, by a thread of its own)
  std::thread flat_thread([&] {
    if (flat_dir.empty()) return;
    mkdir(flat_dir.c_str(), 0755);
    char e2[512] = "";
    for (auto& d : dbs) {
      if (d.from_flat) continue;
      const uint64_t stamp = file_stamp(d.fasta, d.idx_prefix);
      for (size_t k = 0; k < d.parts.size(); k++)
        if (smr_index_save(d.parts[k], flat_path(d, (uint32_t)k).c_str(), stamp, e2, sizeof e2) != SMR_OK) { fprintf(stderr, "smr_align_mgpu: %s (the run goes on without the cache)\n", e2); return; }
    }
  });
  reads_thread.join();
  rccl_thread.join();
  if (reads_rc != SMR_OK) die(rerr);
  const uint64_t n = smr_reads_count(all);
  const int is_fastq = want_reports ? smr_reads_is_fastq(all) : 0;
  const double t_host = now_s() - t0;
  size_t total_parts = 0;
  for (auto& d : dbs) total_parts += d.parts.size();
  if (total_parts > 64) die("more than 64 index parts in total");

  std::vector<RankOut> outs(world);
  const double t_start = now_s();
  auto rank_main = [&](int rank) {
    RankOut& O = outs[rank];
    const int dev = S.devices[rank];
    if (hipSetDevice(dev) != hipSuccess) die("hipSetDevice failed");
    hipStream_t cstream;
    if (hipStreamCreate(&cstream) != hipSuccess) die("hipStreamCreate failed");
    char e2[512] = "";
    smr_ctx* gpu = nullptr;
    if (smr_create(dev, &gpu, e2, sizeof e2) != SMR_OK) die(e2);
    outs[rank].sw_kernel = smr_sw_mode(gpu, -1);
    // the read shard of this rank, in chunks (each chunk is one resident batch of the engine)
    uint64_t first = 0, count = 0;
    shard_range(n, rank, world, first, count);
    const uint64_t chunk = chunk_reads ? chunk_reads : 2000000;
    const size_t n_chunks = (size_t)((count + chunk - 1) / chunk);
    // C1: global read totals (this rank only knows its shard)
    uint64_t* d_tot = nullptr;
    if (hipMalloc((void**)&d_tot, 2 * 8) != hipSuccess) die("hipMalloc failed");
    {
      smr_reads* mine = nullptr;
      if (smr_reads_slice(all, first, count, &mine) != SMR_OK) die("smr_reads_slice failed");
      const uint64_t loc[2] = {smr_reads_count(mine), smr_reads_total_len(mine)};
      smr_reads_free(mine);
      if (hipMemcpy(d_tot, loc, 16, hipMemcpyHostToDevice) != hipSuccess) die("hipMemcpy failed");
    }
    all_reduce_sum_u64(S, rank, d_tot, 2, cstream);
    uint64_t tot[2];
    if (hipMemcpy(tot, d_tot, 16, hipMemcpyDeviceToHost) != hipSuccess) die("hipMemcpy failed");
    (void)hipFree(d_tot);
    // index replica: every part stays resident in its own slot
    double t = now_s();
    std::vector<std::vector<int>> slot(dbs.size());
    int next_slot = 0;
    for (size_t k = 0; k < dbs.size(); k++)
      for (size_t part = 0; part < dbs[k].parts.size(); part++) {
        if (smr_index_upload(gpu, dbs[k].parts[part], next_slot) != SMR_OK) die(smr_last_error(gpu));
        slot[k].push_back(next_slot++);
      }
    const uint32_t slots_per_read = base.num_alignments > 0 ? base.num_alignments : 256;
    std::vector<smr_params> pk(dbs.size(), base);
    for (size_t k = 0; k < dbs.size(); k++) {
      smr_index_info info; smr_index_get_info(dbs[k].parts[0], &info);
      pk[k].minimal_score = smr_minimal_score(dbs[k].lambda, dbs[k].K, info.bg, info.full_len, info.numseq, tot[0], tot[1], evalue);
      pk[k].index_num = (uint32_t)k;
    }
    O.minimal_score0 = pk[0].minimal_score;
    O.t_upload += now_s() - t;
    // the shard's report files (this rank's directory; merged in rank order afterwards)
    smr_report* rep = nullptr;
    if (want_reports) {
      const std::string rd = out_dir + "/rank" + std::to_string(rank);
      mkdir(rd.c_str(), 0777);
      if (smr_report_open(rd.c_str(), &ro, is_fastq, &rep, e2, sizeof e2) != SMR_OK) die(e2);
      for (size_t k = 0; k < dbs.size(); k++) {
        smr_index_info info; smr_index_get_info(dbs[k].parts[0], &info);
        uint64_t fr = 0, fq = 0;
        smr_refstats_corrected(dbs[k].K, info.bg, info.full_len, info.numseq, tot[0], tot[1], &fr, &fq);
        smr_report_set_db(rep, (uint32_t)k, dbs[k].lambda, dbs[k].K, fr, fq);
        for (size_t part = 0; part < dbs[k].parts.size(); part++) smr_report_set_part(rep, (uint32_t)k, (uint32_t)part, dbs[k].parts[part]);
      }
    }
    // ---- three stages over recycled batch slots: upload (thread U) -> align + fetch (this thread) -> records / report rows (thread W) ----
    constexpr int NS = 3;                                // chunk c lives in batch slot c % NS; batch 15 is what stays "selected" between chunks
    std::mutex pm; std::condition_variable pcv;
    size_t uploaded = 0, fetched = 0, written = 0;       // chunks that have completed each stage
    std::vector<smr_reads*> cr(n_chunks, nullptr);
    auto chunk_first = [&](size_t c) { return first + c * chunk; };
    auto chunk_count = [&](size_t c) { return std::min<uint64_t>(chunk, first + count - chunk_first(c)); };
    if (smr_batch_select(gpu, 15) != SMR_OK) die(smr_last_error(gpu));
    double t_up = 0, t_wr = 0;
    std::thread U([&] {
      for (size_t c = 0; c < n_chunks; c++) {
        { std::unique_lock<std::mutex> l(pm); pcv.wait(l, [&] { return c < (size_t)NS || written + NS > c; }); }        // its slot's previous chunk has been written out
        const double tu = now_s();
        if (smr_reads_slice(all, chunk_first(c), chunk_count(c), &cr[c]) != SMR_OK) die("smr_reads_slice failed");
        if (smr_reads_upload_batch(gpu, (int)(c % NS), cr[c], slots_per_read) != SMR_OK) die(smr_last_error(gpu));
        t_up += now_s() - tu;
        { std::lock_guard<std::mutex> l(pm); uploaded = c + 1; }
        pcv.notify_all();
      }
    });
    std::thread Wt([&] {
      std::vector<uint8_t> rec; std::vector<char> hh, ss, qq;
      for (size_t c = 0; c < n_chunks; c++) {
        { std::unique_lock<std::mutex> l(pm); pcv.wait(l, [&] { return fetched > c; }); }
        const double tw = now_s();
        const uint32_t cnt = smr_reads_count(cr[c]);
        for (uint32_t i = 0; i < cnt; i++) {
          // results of the shard (kvdb.put(read.id, read.toBinString()), processor.cpp:150-155), keys carry the GLOBAL read number
          const size_t len = smr_result_record_batch(gpu, (int)(c % NS), i, nullptr, 0);
          if (!len && !(rep && ro.other)) continue;
          rec.resize(len);
          if (len) smr_result_record_batch(gpu, (int)(c % NS), i, rec.data(), len);
          const uint64_t gi = chunk_first(c) + i;
          if (rep) {
            size_t tl[3];
            smr_reads_record_text(all, (uint32_t)gi, nullptr, 0, nullptr, 0, nullptr, 0, tl);
            hh.resize(tl[0] + 1); ss.resize(tl[1] + 1); qq.resize(tl[2] + 1);
            smr_reads_record_text(all, (uint32_t)gi, hh.data(), hh.size(), ss.data(), ss.size(), qq.data(), qq.size(), tl);
            if (smr_report_add(rep, hh.data(), ss.data(), is_fastq ? qq.data() : nullptr, len ? rec.data() : nullptr, len) != SMR_OK) die(smr_report_last_error(rep));
          }
          if (!len) continue;
          const std::string key = "0_" + std::to_string(gi);
          const uint64_t kl = key.size(), vl = len;
          const size_t o = O.records.size();
          O.records.resize(o + 16 + kl + vl);
          memcpy(&O.records[o], &kl, 8); memcpy(&O.records[o + 8], key.data(), kl); memcpy(&O.records[o + 8 + kl], &vl, 8); memcpy(&O.records[o + 16 + kl], rec.data(), vl);
          O.n_records++;
        }
        smr_reads_free(cr[c]); cr[c] = nullptr;
        t_wr += now_s() - tw;
        { std::lock_guard<std::mutex> l(pm); written = c + 1; }
        pcv.notify_all();
      }
    });
    // C2 accumulator: the Readstats counters of all chunks of this rank, summed on the device
    void* d_acc = nullptr; uint32_t nctr = 0;
    { void* dummy = nullptr; if (smr_counters_device(gpu, &dummy, &nctr) != SMR_OK) die(smr_last_error(gpu)); }
    if (hipMalloc(&d_acc, (size_t)nctr * 8) != hipSuccess || hipMemset(d_acc, 0, (size_t)nctr * 8) != hipSuccess) die("hipMalloc failed");
    for (size_t c = 0; c < n_chunks; c++) {
      { std::unique_lock<std::mutex> l(pm); pcv.wait(l, [&] { return uploaded > c; }); }
      t = now_s();
      if (smr_batch_select(gpu, (int)(c % NS)) != SMR_OK) die(smr_last_error(gpu));
      for (size_t k = 0; k < dbs.size(); k++)
        for (size_t part = 0; part < dbs[k].parts.size(); part++) {
          smr_params p = pk[k];
          p.part = (uint32_t)part;
          p.is_last_index_part = (k + 1 == dbs.size() && part + 1 == dbs[k].parts.size());
          if (smr_align_part(gpu, slot[k][part], &p) != SMR_OK || smr_traceback(gpu, slot[k][part], &p) != SMR_OK) die(smr_last_error(gpu));
        }
      if (smr_counters_accumulate(gpu, d_acc, nctr) != SMR_OK || smr_results_fetch(gpu) != SMR_OK) die(smr_last_error(gpu));
      if (smr_batch_select(gpu, 15) != SMR_OK) die(smr_last_error(gpu));          // nothing of slot c % NS is "selected" while the uploader refills it later
      O.t_align += now_s() - t;
      { std::lock_guard<std::mutex> l(pm); fetched = c + 1; }
      pcv.notify_all();
    }
    U.join(); Wt.join();
    O.t_upload += t_up;
    t = now_s();
    // C2: reduced over the ranks IN PLACE on the device block of sums
    all_reduce_sum_u64(S, rank, d_acc, nctr, cstream);
    {
      std::vector<uint64_t> hc(nctr);
      if (hipMemcpy(hc.data(), d_acc, (size_t)nctr * 8, hipMemcpyDeviceToHost) != hipSuccess) die("hipMemcpy failed");
      O.counters.assign(2 + dbs.size(), 0);
      O.counters[0] = hc[0]; O.counters[1] = hc[1];
      for (size_t k = 0; k < dbs.size(); k++) O.counters[2 + k] = hc[2 + k];
    }
    (void)hipFree(d_acc);
    if (rep && smr_report_close(rep) != SMR_OK) die("cannot write the report files");
    O.t_write = t_wr;
    O.t_fetch = now_s() - t;
    smr_destroy(gpu);
    (void)hipStreamDestroy(cstream);
  };
  std::vector<std::thread> th;
  for (int r = 0; r < world; r++) th.emplace_back(rank_main, r);
  for (auto& t : th) t.join();
  const double t_ranks = now_s() - t_start;
  // (the communicators are not destroyed: the process ends with _exit below)

  // ---- report files: the shards' files concatenated in rank order (Report::merge, report.cpp:56-97) ----
  if (want_reports) {
    const char* names[] = {"aligned.fq", "aligned.fa", "other.fq", "other.fa", "aligned.blast", "aligned.sam"};
    std::vector<char> buf(1 << 22);
    for (const char* nm : names) {
      FILE* o = nullptr;
      for (int r = 0; r < world; r++) {
        const std::string pth = out_dir + "/rank" + std::to_string(r) + "/" + nm;
        if (!o) {                                            // the first shard's file becomes the merged file (no copy), the others are appended
          if (rename(pth.c_str(), (out_dir + "/" + nm).c_str()) != 0) continue;
          if (!(o = fopen((out_dir + "/" + nm).c_str(), "ab"))) die(std::string("cannot write ") + nm);
          continue;
        }
        FILE* in = fopen(pth.c_str(), "rb");
        if (!in) continue;
        size_t g;
        bool first_line = true;
        while ((g = fread(buf.data(), 1, buf.size(), in)) > 0) {
          size_t off = 0;
          if (first_line && std::string(nm) == "aligned.sam") {          // one @HD / @PG header block: the later shards' header lines are dropped
            while (off < g && buf[off] == '@') { while (off < g && buf[off] != '\n') off++; if (off < g) off++; }
          }
          first_line = false;
          fwrite(buf.data() + off, 1, g - off, o);
        }
        fclose(in);
        remove(pth.c_str());
      }
      if (o) fclose(o);
    }
    for (int r = 0; r < world; r++) rmdir((out_dir + "/rank" + std::to_string(r)).c_str());
  }
  // ---- rank 0's view of the reduced counters is everybody's; shards concatenate in rank order ----
  for (int r = 1; r < world; r++) if (outs[r].counters != outs[0].counters) die("the ranks disagree on the reduced counters");
  const std::string rp = out_dir + "/records.bin", sp = out_dir + "/summary.txt";
  FILE* f = fopen(rp.c_str(), "wb");
  if (!f) die("cannot write " + rp);
  uint64_t nrec = 0;
  for (auto& o : outs) nrec += o.n_records;
  fwrite(&nrec, 8, 1, f);
  for (auto& o : outs) if (!o.records.empty()) fwrite(o.records.data(), 1, o.records.size(), f);
  fclose(f);
  f = fopen(sp.c_str(), "w");
  if (!f) die("cannot write " + sp);
  const auto& ctr = outs[0].counters;
  fprintf(f, "Total reads = %llu\nTotal reads passing E-value threshold = %llu\nToo short reads (last part) = %llu\n", (unsigned long long)n,
          (unsigned long long)ctr[0], (unsigned long long)ctr[1]);
  for (size_t k = 0; k < dbs.size(); k++) fprintf(f, "%s\t%llu\n", dbs[k].fasta.c_str(), (unsigned long long)ctr[2 + k]);
  fclose(f);
  double tu = 0, ta = 0, tf = 0, tw = 0;
  for (auto& o : outs) { tu = std::max(tu, o.t_upload); ta = std::max(ta, o.t_align); tf = std::max(tf, o.t_fetch); tw = std::max(tw, o.t_write); }
  const double t_all = now_s() - t0;
  printf("[timing] ranks %d (%s reduction), reads %llu: parse+pack %.3f s alongside index load/build %.3f s = %.3f s, per-rank max: index upload + chunk uploads %.3f s (overlapped), align+traceback+fetch %.3f s, "
         "records%s %.3f s (overlapped), counter reduce + close %.3f s; rank stage %.3f s; end to end %.3f s = %.0f reads/s (rank stage alone: %.0f reads/s)\n",
         world, S.use_rccl ? "RCCL" : "host", (unsigned long long)n, t_reads, t_index, t_host, tu, ta, want_reports ? " + report rows" : "", tw, tf, t_ranks, t_all, n / t_all, n / t_ranks);
  int swk = 2; for (auto& o : outs) swk = std::min(swk, o.sw_kernel);
  printf("[kernels] Smith-Waterman: %s\n", swk >= 1 ? "packed 16-bit (four candidate windows per wave)" : "32-bit kernel ONLY on at least one rank (packed kernel off or failed its self-check): expect about half the alignment rate");
  printf("%llu reads, %llu aligned, %llu records, minimal_score %u -> %s\n", (unsigned long long)n, (unsigned long long)ctr[0], (unsigned long long)nrec, outs[0].minimal_score0, rp.c_str());
  flat_thread.join();
  printf("[timing] process: %.3f s from main() to here (options + device discovery %.3f s, RCCL communicators %.3f s alongside the loading); the host copies of index and reads are "
         "left to the operating system\n", now_s() - t_main, t0 - t_main, t_rccl);
  fflush(stdout); fflush(stderr);
  _exit(0);                                                // (no teardown of GBs of host arrays, device memory and communicators at the end of a command-line run)
}
