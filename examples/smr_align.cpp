// smr_align.cpp -- a C++17 host driver over the C ABI of libsmr_hip (include/smr_hip.h): the shape of the reference's
// align() (/root/reference/src/sortmerna/processor.cpp:173-285) with the N x align2() thread loop of every (index, part)
// replaced by one batched GPU call, for one or more --ref databases.  It writes what the reference keeps in its KVDB:
//   <out>/records.bin   u64 n, then n x (u64 klen, key "0_<i>", u64 vlen, Read::toBinString bytes)   (reads with a record only)
//   <out>/summary.txt   total reads, reads passing the E-value threshold, per-DB counts, too-short reads
// and, on request, the reference's reports through smr_report_* (aligned/other FASTX, BLAST tabular, SAM):
//   --fastx --other --blast "1 cigar qcov qstrand" --sam
// Build:  g++ -std=c++17 -O2 examples/smr_align.cpp -Iinclude -Lsortmerna_amd/lib -lsmr_hip -Wl,-rpath,$PWD/sortmerna_amd/lib -o smr_align
// There is no CPU fallback: without a HIP device smr_create fails and the program exits like the reference does (ERR + exit 1).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "smr_hip.h"

namespace {
[[noreturn]] void die(const std::string& m) { fprintf(stderr, "ERROR: %s\n", m.c_str()); exit(EXIT_FAILURE); }
struct Db { std::string fasta, idx_prefix; double lambda = 0, K = 0; bool has_gumbel = false; std::vector<smr_index*> parts; };
}  // namespace

int main(int argc, char** argv) {
  std::vector<Db> dbs;
  std::vector<std::string> reads_paths; std::string out_dir = ".";
  int zip_out = -1;                                         // -1: like the reads file (options.hpp zip_out, report_fx_base.cpp:94)
  smr_params base; smr_params_default(&base);
  double evalue = 1.0;
  int device = 0;
  smr_report_opts ro; memset(&ro, 0, sizeof ro);
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto val = [&]() -> std::string { if (i + 1 >= argc) die("missing value after " + a); return argv[++i]; };
    if (a == "-ref" || a == "--ref") { Db d; d.fasta = val(); dbs.push_back(d); }
    else if (a == "-idx" || a == "--idx") { if (dbs.empty()) die("--idx before --ref"); dbs.back().idx_prefix = val(); }   // reference-format index files
    else if (a == "-gumbel" || a == "--gumbel") { if (dbs.empty()) die("--gumbel before --ref"); dbs.back().lambda = atof(val().c_str()); dbs.back().K = atof(val().c_str()); dbs.back().has_gumbel = true; }
    else if (a == "-reads" || a == "--reads") { if (reads_paths.size() == 2) die("at most two --reads files (mates)"); reads_paths.push_back(val()); }
    else if (a == "-out" || a == "--out") out_dir = val();
    else if (a == "-e") evalue = atof(val().c_str());
    else if (a == "-num_alignments" || a == "--num_alignments") base.num_alignments = (uint32_t)atoi(val().c_str());
    else if (a == "-no-best" || a == "--no-best") base.is_best = 0;
    else if (a == "-min_lis" || a == "--min_lis") base.min_lis = atoi(val().c_str());
    else if (a == "-num_seeds" || a == "--num_seeds") base.num_seeds = atoi(val().c_str());
    else if (a == "-edges" || a == "--edges") base.edges = atoi(val().c_str());
    else if (a == "-full_search" || a == "--full_search") base.is_full_search = 1;
    else if (a == "-F") base.is_reverse = 0;
    else if (a == "-R") base.is_forward = 0;
    else if (a == "-match") base.match = atoi(val().c_str());
    else if (a == "-mismatch") { base.mismatch = atoi(val().c_str()); base.score_N = base.mismatch; }
    else if (a == "-gap_open") base.gap_open = atoi(val().c_str());
    else if (a == "-gap_ext") base.gap_ext = atoi(val().c_str());
    else if (a == "-device") device = atoi(val().c_str());
    else if (a == "-fastx" || a == "--fastx") ro.fastx = 1;
    else if (a == "-other" || a == "--other") ro.other = 1;
    else if (a == "-sam" || a == "--sam") ro.sam = 1;
    else if (a == "-SQ" || a == "--SQ") ro.sam_sq = 1;
    else if (a == "-paired_in" || a == "--paired_in") ro.paired_in = 1;
    else if (a == "-paired_out" || a == "--paired_out") ro.paired_out = 1;
    else if (a == "-out2" || a == "--out2") ro.out2 = 1;
    else if (a == "-sout" || a == "--sout") ro.sout = 1;
    else if (a == "-zip-out" || a == "--zip-out") { const std::string v = val(); zip_out = (v == "1" || v == "true" || v == "t" || v == "yes" || v == "y") ? 1 : 0; }
    else if (a == "-blast" || a == "--blast") {            // "1" = tabular, optionally followed by cigar / qcov / qstrand (options.cpp opt_blast)
      const std::string v = val();
      if (v == "0") { ro.blast_pairwise = 1; continue; }   // BLAST-like pairwise text
      if (v.empty() || v[0] != '1') die("-blast: '0' (pairwise) or '1 [cigar] [qcov] [qstrand]' (tabular)");
      ro.blast_tabular = 1;
      const std::string cols = v.size() > 2 ? v.substr(2) : "";
      if (cols.size() >= sizeof ro.blast_cols) die("-blast: too many columns");
      strcpy(ro.blast_cols, cols.c_str());
    }
    else if (a == "-h" || a == "--help") {
      printf("usage: smr_align --ref DB.fasta --gumbel LAMBDA K [--idx PREFIX] [--ref ...] --reads READS.fa|fq[.gz] [--reads MATES] [--out DIR]\n"
             "       [-e EVALUE] [-num_alignments N] [-no-best] [-min_lis N] [-num_seeds N] [-edges N] [-full_search] [-F|-R]\n"
             "       [-match N -mismatch N -gap_open N -gap_ext N] [-device K] [--fastx] [--other] [--blast '0' | '1 cigar qcov qstrand'] [--sam [-SQ]]\n"
             "       [-zip-out 0|1] [-paired_in | -paired_out] [-out2] [-sout]     (two --reads files, or one interleaved file with -paired_in / -paired_out)\n");
      return 0;
    } else die("unknown option " + a);
  }
  if (dbs.empty() || reads_paths.empty()) die("--ref and --reads are required (see --help)");
  if (const char* why = smr_params_refused(&base)) die(std::string("these options are outside what libsmr_hip aligns (the reference accepts them): ") + why);
  // minimal_score (which reads count as aligned) and the e-values / bit scores of the BLAST report depend on the Gumbel parameters of the
  // (scoring scheme, DB background) pair.  The reference computes them per DB with its vendored NCBI ALP library (refstats.cpp:194-233),
  // which is outside this library: they must be given -- from the reference's log ("Gumbel lambda = ..", "Gumbel K = ..") for the same DB
  // and the same -match/-mismatch/-gap_open/-gap_ext.  No default is assumed: a silent guess would classify a different set of reads.
  for (auto& d : dbs) if (!d.has_gumbel) die("--gumbel LAMBDA K is required after every --ref (see --help): the values of the reference's log for this DB and scoring scheme");
  char err[512] = "";
  // reads (Readfeed::next -> Read::init, readfeed.hpp:124 / read.cpp:264-347)
  // (all cores parse and 2-bit pack the FASTA / FASTQ / .gz file; the text stays mapped for the report writers)
  // one batch per reads file (the second file holds the mates of the first: options.cpp:1591 is_paired)
  std::vector<smr_reads*> rf(reads_paths.size(), nullptr);
  uint64_t n = 0, total_len = 0; uint32_t min_len = 0xFFFFFFFFu, max_len = 0;
  for (size_t b = 0; b < rf.size(); b++) {
    if (smr_reads_load_fastx_text(reads_paths[b].c_str(), 0, &rf[b], err, sizeof err) != SMR_OK) die(err);
    n += smr_reads_count(rf[b]); total_len += smr_reads_total_len(rf[b]);
    if (smr_reads_count(rf[b])) { min_len = std::min(min_len, smr_reads_min_len(rf[b])); max_len = std::max(max_len, smr_reads_max_len(rf[b])); }
  }
  if (n == 0) min_len = 0;
  const bool is_fastq = smr_reads_is_fastq(rf[0]) != 0;
  if (zip_out == -1) {                                      // gzip reports for a gzip reads file
    unsigned char mg[2] = {0, 0};
    if (FILE* fz = fopen(reads_paths[0].c_str(), "rb")) { if (fread(mg, 1, 2, fz) != 2) mg[0] = 0; fclose(fz); }
    zip_out = (mg[0] == 0x1f && mg[1] == 0x8b) ? 1 : 0;
  }
  ro.zip_out = zip_out;
  const bool paired = rf.size() == 2 || ro.paired_in || ro.paired_out;
  if (rf.size() == 2 && smr_reads_count(rf[0]) != smr_reads_count(rf[1])) die("the two --reads files hold different numbers of reads");
  if (rf.size() == 1 && paired && (smr_reads_count(rf[0]) & 1)) die("-paired_in / -paired_out with one file: odd number of reads");

  // indexes: reference-built files when a prefix is given, else our own builder (Index ctor / build_index, index.cpp:61-107)
  for (auto& d : dbs) {
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
      smr_index* arr[256]; uint32_t np = 0;
      if (smr_index_build(d.fasta.c_str(), 18, 3072.0, 10000, 0, arr, 256, &np, err, sizeof err) != SMR_OK) die(err);
      d.parts.assign(arr, arr + np);
    }
  }

  smr_ctx* gpu = nullptr;
  if (smr_create(device, &gpu, err, sizeof err) != SMR_OK) die(err);
  if (smr_sw_mode(gpu, -1) == 0) fprintf(stderr, "smr_align: the 32-bit Smith-Waterman kernel is in use (packed kernel off or failed its self-check): expect about half the alignment rate\n");
  const uint32_t slots = base.num_alignments > 0 ? base.num_alignments : 256;
  for (size_t b = 0; b < rf.size(); b++)
    if (smr_batch_select(gpu, (int)b) != SMR_OK || smr_reads_upload(gpu, rf[b], slots) != SMR_OK) die(smr_last_error(gpu));

  // the (index, part) loop of processor.cpp:219-277
  for (size_t k = 0; k < dbs.size(); k++) {
    smr_index_info info; smr_index_get_info(dbs[k].parts[0], &info);
    smr_params p = base;
    p.minimal_score = smr_minimal_score(dbs[k].lambda, dbs[k].K, info.bg, info.full_len, info.numseq, n, total_len, evalue);
    p.index_num = (uint32_t)k;
    for (size_t part = 0; part < dbs[k].parts.size(); part++) {
      p.part = (uint32_t)part;
      p.is_last_index_part = (k + 1 == dbs.size() && part + 1 == dbs[k].parts.size());
      if (smr_index_upload(gpu, dbs[k].parts[part], 0) != SMR_OK) die(smr_last_error(gpu));
      for (size_t b = 0; b < rf.size(); b++) {
        if (smr_batch_select(gpu, (int)b) != SMR_OK || smr_align_part(gpu, 0, &p) != SMR_OK || smr_traceback(gpu, 0, &p) != SMR_OK) die(smr_last_error(gpu));
      }
      smr_index_unload(gpu, 0);
    }
  }
  for (size_t b = 0; b < rf.size(); b++) if (smr_batch_select(gpu, (int)b) != SMR_OK || smr_results_fetch(gpu) != SMR_OK) die(smr_last_error(gpu));

  // reports (writeReports, output.cpp:169-272)
  smr_report* rep = nullptr;
  std::string cmdline;
  for (int i = 0; i < argc; i++) { cmdline += argv[i]; cmdline += ' '; }       // the reference's opts.cmdline keeps the trailing blank
  if (ro.fastx || ro.other || ro.blast_tabular || ro.blast_pairwise || ro.sam) {
    if (smr_report_open(out_dir.c_str(), &ro, is_fastq, &rep, err, sizeof err) != SMR_OK) die(err);
    smr_report_set_cmdline(rep, cmdline.c_str());
    for (size_t k = 0; k < dbs.size(); k++) {
      smr_index_info info; smr_index_get_info(dbs[k].parts[0], &info);
      uint64_t fr = 0, fq = 0;
      smr_refstats_corrected(dbs[k].K, info.bg, info.full_len, info.numseq, n, total_len, &fr, &fq);
      smr_report_set_db(rep, (uint32_t)k, dbs[k].lambda, dbs[k].K, fr, fq);
      for (size_t part = 0; part < dbs[k].parts.size(); part++) smr_report_set_part(rep, (uint32_t)k, (uint32_t)part, dbs[k].parts[part]);
    }
  }
  // kvdb.put(read.id, read.toBinString()) (processor.cpp:150-155) -> records.bin ; Readstats -> summary.txt
  const std::string rp = out_dir + "/records.bin", sp = out_dir + "/summary.txt";
  FILE* f = fopen(rp.c_str(), "wb");
  if (!f) die("cannot write " + rp);
  uint64_t nrec = 0;
  fwrite(&nrec, 8, 1, f);
  struct Mate { std::vector<uint8_t> rec; std::vector<char> h, s, q; };
  auto load = [&](size_t b, uint32_t i, Mate& m) {                  // record + text of read i of batch b
    smr_batch_select(gpu, (int)b);
    const size_t len = smr_result_record(gpu, i, nullptr, 0);
    m.rec.resize(len);
    if (len) smr_result_record(gpu, i, m.rec.data(), len);
    if (rep) {
      size_t tl[3];
      smr_reads_record_text(rf[b], i, nullptr, 0, nullptr, 0, nullptr, 0, tl);
      m.h.resize(tl[0] + 1); m.s.resize(tl[1] + 1); m.q.resize(tl[2] + 1);
      smr_reads_record_text(rf[b], i, m.h.data(), m.h.size(), m.s.data(), m.s.size(), m.q.data(), m.q.size(), tl);
    }
    if (len) {
      const std::string key = std::to_string(b) + "_" + std::to_string(i);      // KVDB key: <reads file>_<read number>
      const uint64_t kl = key.size(), vl = len;
      fwrite(&kl, 8, 1, f); fwrite(key.data(), 1, kl, f); fwrite(&vl, 8, 1, f); fwrite(m.rec.data(), 1, vl, f);
      nrec++;
    }
  };
  Mate m1, m2;
  if (paired) {
    const uint32_t np = rf.size() == 2 ? smr_reads_count(rf[0]) : smr_reads_count(rf[0]) / 2;
    for (uint32_t i = 0; i < np; i++) {
      if (rf.size() == 2) { load(0, i, m1); load(1, i, m2); } else { load(0, 2 * i, m1); load(0, 2 * i + 1, m2); }
      if (rep && smr_report_add_pair(rep, m1.h.data(), m1.s.data(), is_fastq ? m1.q.data() : nullptr, m1.rec.data(), m1.rec.size(),
                                     m2.h.data(), m2.s.data(), is_fastq ? m2.q.data() : nullptr, m2.rec.data(), m2.rec.size()) != SMR_OK) die(smr_report_last_error(rep));
    }
  } else {
    for (uint32_t i = 0; i < smr_reads_count(rf[0]); i++) {
      load(0, i, m1);
      if (rep && smr_report_add(rep, m1.h.data(), m1.s.data(), is_fastq ? m1.q.data() : nullptr, m1.rec.data(), m1.rec.size()) != SMR_OK) die(smr_report_last_error(rep));
    }
  }
  std::vector<uint64_t> ctr(2 + dbs.size(), 0), cb(2 + dbs.size());
  for (size_t b = 0; b < rf.size(); b++) {
    smr_batch_select(gpu, (int)b);
    smr_counters(gpu, cb.data(), (uint32_t)dbs.size());
    for (size_t k = 0; k < ctr.size(); k++) ctr[k] += cb[k];
  }
  {  // readstats.store_to_db(kvdb) (processor.cpp:283-284): one more KVDB entry, key = hash of the read files' names
    const char* rfn[2] = {reads_paths[0].c_str(), reads_paths.size() > 1 ? reads_paths[1].c_str() : nullptr};
    char key[32];
    const uint64_t kl = smr_readstats_key(rfn, (uint32_t)reads_paths.size(), key, sizeof key);
    std::vector<uint8_t> rs(smr_readstats_record(n, total_len, min_len, max_len, ctr[0], ctr[1], ctr.data() + 2, (uint32_t)dbs.size(), nullptr, 0));
    smr_readstats_record(n, total_len, min_len, max_len, ctr[0], ctr[1], ctr.data() + 2, (uint32_t)dbs.size(), rs.data(), rs.size());
    const uint64_t vl = rs.size();
    fwrite(&kl, 8, 1, f); fwrite(key, 1, kl, f); fwrite(&vl, 8, 1, f); fwrite(rs.data(), 1, vl, f);
    nrec++;
  }
  fseek(f, 0, SEEK_SET); fwrite(&nrec, 8, 1, f); fclose(f);
  if (rep && smr_report_close(rep) != SMR_OK) die("cannot write the report files");
  f = fopen(sp.c_str(), "w");
  if (!f) die("cannot write " + sp);
  fprintf(f, "Total reads = %llu\nTotal reads passing E-value threshold = %llu\nToo short reads (last part) = %llu\n", (unsigned long long)n,
          (unsigned long long)ctr[0], (unsigned long long)ctr[1]);
  for (size_t k = 0; k < dbs.size(); k++) fprintf(f, "%s\t%llu\n", dbs[k].fasta.c_str(), (unsigned long long)ctr[2 + k]);
  fclose(f);
  {  // aligned.log (Summary::write, summary.cpp:57-100)
    std::vector<smr_summary_db> sdb(dbs.size());
    for (size_t k = 0; k < dbs.size(); k++) {
      smr_index_info info; smr_index_get_info(dbs[k].parts[0], &info);
      sdb[k].ref_file = dbs[k].fasta.c_str();
      sdb[k].skiplengths[0] = info.lnwin; sdb[k].skiplengths[1] = info.lnwin / 2; sdb[k].skiplengths[2] = 3;
      sdb[k].lambda = dbs[k].lambda; sdb[k].K = dbs[k].K;
      sdb[k].minimal_score = smr_minimal_score(dbs[k].lambda, dbs[k].K, info.bg, info.full_len, info.numseq, n, total_len, evalue);
      sdb[k].reads_matched = ctr[2 + k];
    }
    const time_t now = time(nullptr);
    const std::string stamp = ctime(&now);
    const char* rfn[2] = {reads_paths[0].c_str(), reads_paths.size() > 1 ? reads_paths[1].c_str() : nullptr};
    smr_summary sm; memset(&sm, 0, sizeof sm);
    sm.cmdline = cmdline.c_str(); sm.pid = ""; sm.timestamp = stamp.c_str();
    sm.seed_len = 18; sm.num_seeds = base.num_seeds; sm.edges = base.edges; sm.match = base.match; sm.mismatch = base.mismatch;
    sm.gap_open = base.gap_open; sm.gap_ext = base.gap_ext; sm.score_N = base.score_N; sm.sam_sq = ro.sam_sq; sm.threads = 1;
    sm.reads_files = rfn; sm.n_reads_files = (uint32_t)reads_paths.size();
    sm.total_reads = n; sm.num_aligned = ctr[0]; sm.all_reads_len = total_len;
    sm.min_read_len = min_len; sm.max_read_len = max_len;
    sm.dbs = sdb.data(); sm.n_dbs = (uint32_t)sdb.size();
    if (smr_summary_write((out_dir + "/aligned.log").c_str(), &sm) != SMR_OK) die("cannot write aligned.log");
  }
  printf("%llu reads, %llu aligned, %llu records -> %s\n", (unsigned long long)n, (unsigned long long)ctr[0], (unsigned long long)nrec, rp.c_str());
  for (auto& d : dbs) for (auto* ix : d.parts) smr_index_free(ix);
  for (auto* r : rf) smr_reads_free(r);
  smr_destroy(gpu);
  return 0;
}
