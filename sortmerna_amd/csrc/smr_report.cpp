// smr_report.cpp -- host side, SURVEY.md 8(f) N1: the report writers, fed by the per-read records of libsmr_hip.
//
// Replaces, row for row, the reference's second pass over reads + KVDB (writeReports, /root/reference/src/sortmerna/output.cpp:169-272):
//   aligned / other FASTX      ReportFxBase::write_a_read         report_fx_base.cpp:176-205, report_fastx.cpp:134-146, report_fx_other.cpp
//   BLAST tabular (+ cigar / qcov / qstrand)   ReportBlast::append   report_blast.cpp:253-354 (e-value / bit score :118-125)
//   BLAST pairwise (-blast 0)  ReportBlast::append                report_blast.cpp:130-252
//   SAM (+ @SQ header lines)   ReportSam::append / write_header   report_sam.cpp:64-152, 155-211
//   aligned.log                Summary::to_string                 summary.cpp:102-175
//   %id / mismatches / gaps    Read::calc_miss_gap_match          read.cpp:547-589
// Input per read: the original header line, letters, quality and the Read::toBinString record (read.cpp:429-462) that
// smr_result_record returns.  Rows are buffered per (index, part) and written in that order, like the reference's loop.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iomanip>
#include <map>
#include <sstream>
#include <functional>
#include <string>
#include <vector>

#include <zlib.h>

#include "smr_host.hpp"

namespace {
// one report file, plain or gzip (the reference deflates its reports when the reads file is gzip, or with -zip-out; the names get ".gz")
struct Out {
  FILE* f = nullptr; gzFile g = nullptr;
  bool open(const std::string& path, bool zip) {
    if (zip) { g = gzopen((path + ".gz").c_str(), "wb"); return g != nullptr; }
    f = fopen(path.c_str(), "wb"); return f != nullptr;
  }
  bool is_open() const { return f || g; }
  void put(const void* p, size_t n) { if (!n) return; if (g) gzwrite(g, p, (unsigned)n); else if (f) fwrite(p, 1, n, f); }
  void put(const std::string& t) { put(t.data(), t.size()); }
  void close() { if (g) gzclose(g); if (f) fclose(f); g = nullptr; f = nullptr; }
};
struct Aln {
  std::vector<uint32_t> cigar;
  uint32_t ref_num = 0; int32_t ref_begin1 = 0, ref_end1 = 0, read_begin1 = 0, read_end1 = 0; uint32_t readlen = 0;
  uint16_t score1 = 0, part = 0, index_num = 0; uint8_t strand = 0;
};
struct Db { double lambda = 0, K = 0; uint64_t full_ref = 0, full_read = 0; };

// Read::toBinString layout (read.cpp:429-462, ssw.hpp:106-140)
bool parse_record(const uint8_t* b, size_t n, bool& is_hit, std::vector<Aln>& out) {
  out.clear(); is_hit = false;
  if (n == 0) return true;
  size_t o = 0;
  auto rd = [&](void* dst, size_t k) { if (o + k > n) return false; memcpy(dst, b + o, k); o += k; return true; };
  uint32_t u32[6]; uint8_t fl[3]; uint16_t sw; int32_t na; uint32_t hs; uint64_t asz; uint32_t mn, mx; uint64_t cnt;
  if (!rd(u32, 24) || !rd(fl, 3) || !rd(&sw, 2) || !rd(&na, 4) || !rd(&hs, 4) || !rd(&asz, 8) || !rd(&mn, 4) || !rd(&mx, 4) || !rd(&cnt, 8)) return false;
  is_hit = fl[1] != 0;
  for (uint64_t k = 0; k < cnt; k++) {
    uint64_t rl, cl;
    if (!rd(&rl, 8) || !rd(&cl, 8)) return false;
    Aln a; a.cigar.resize(cl);
    if (cl && !rd(a.cigar.data(), cl * 4)) return false;
    if (!rd(&a.ref_num, 4) || !rd(&a.ref_begin1, 4) || !rd(&a.ref_end1, 4) || !rd(&a.read_begin1, 4) || !rd(&a.read_end1, 4) || !rd(&a.readlen, 4) ||
        !rd(&a.score1, 2) || !rd(&a.part, 2) || !rd(&a.index_num, 2) || !rd(&a.strand, 1)) return false;
    out.push_back(std::move(a));
  }
  return o == n;
}

inline int nt_code(unsigned char c) {
  switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': case 'U': case 'u': return 3; default: return 4; }
}
std::string cigar_text(const Aln& a, size_t readlen) {        // soft clips as report_blast.cpp:293-311 / report_sam.cpp:96-113
  std::ostringstream ss;
  if (a.read_begin1 != 0) ss << a.read_begin1 << "S";
  for (uint32_t c : a.cigar) ss << (c >> 4) << ((c & 0xF) == 0 ? "M" : ((c & 0xF) == 1 ? "I" : "D"));
  const long end_mask = (long)readlen - a.read_end1 - 1;
  if (end_mask > 0) ss << end_mask << "S";
  return ss.str();
}
}  // namespace

struct smr_report {
  std::string dir; smr_report_opts o; bool fastq = false;
  Out f_aligned[4], f_other[4];
  int num_out = 1;                                                     // ReportFxBase::set_num_out (report_fx_base.cpp:163-169)
  std::map<uint32_t, Db> dbs;
  std::map<std::pair<uint32_t, uint32_t>, const smr_index*> parts;
  std::map<std::pair<uint32_t, uint32_t>, std::string> blast, sam;     // rows per (index, part)
  std::string err, cmdline = "libsmr_hip";
};

extern "C" int smr_report_open(const char* out_dir, const smr_report_opts* opts, int is_fastq, smr_report** out, char* err, size_t errcap) {
  if (!out_dir || !opts || !out) return SMR_ERR_ARG;
  auto r = new smr_report();
  r->dir = out_dir; r->o = *opts; r->fastq = is_fastq != 0;
  const std::string ext = is_fastq ? ".fq" : ".fa";
  if ((opts->paired_in && opts->paired_out) || (opts->sout && (opts->paired_in || opts->paired_out))) {      // report_fx_base.cpp:131-136
    if (err && errcap) snprintf(err, errcap, "invalid combination of paired_in / paired_out / sout");
    delete r; return SMR_ERR_ARG;
  }
  r->num_out = (opts->out2 && opts->sout) ? 4 : (opts->out2 || opts->sout) ? 2 : 1;
  // file name suffixes (report_fx_base.cpp:71-91; the per-split files are merged into these names, report.cpp:56-97)
  auto sfx = [&](int j) -> std::string {
    if (r->num_out == 4) return j == 0 ? "_paired_fwd" : j == 1 ? "_paired_rev" : j == 2 ? "_singleton_fwd" : "_singleton_rev";
    if (r->num_out == 2) return opts->out2 ? (j == 0 ? "_fwd" : "_rev") : (j == 0 ? "_paired" : "_singleton");
    return "";
  };
  bool ok = true;
  for (int j = 0; j < r->num_out; j++) {
    if (opts->fastx) ok = r->f_aligned[j].open(r->dir + "/aligned" + sfx(j) + ext, opts->zip_out != 0) && ok;
    if (opts->other) ok = r->f_other[j].open(r->dir + "/other" + sfx(j) + ext, opts->zip_out != 0) && ok;
  }
  if (!ok) {
    if (err && errcap) snprintf(err, errcap, "cannot create report files in %s", out_dir);
    for (int j = 0; j < 4; j++) { r->f_aligned[j].close(); r->f_other[j].close(); }
    delete r; return SMR_ERR_IO;
  }
  *out = r;
  return SMR_OK;
}

extern "C" int smr_report_set_db(smr_report* r, uint32_t index_num, double lambda, double K, uint64_t full_ref_corr, uint64_t full_read_corr) {
  if (!r) return SMR_ERR_ARG;
  Db d; d.lambda = lambda; d.K = K; d.full_ref = full_ref_corr; d.full_read = full_read_corr;
  r->dbs[index_num] = d;
  return SMR_OK;
}

extern "C" int smr_report_set_part(smr_report* r, uint32_t index_num, uint32_t part, const smr_index* ix) {
  if (!r || !ix) return SMR_ERR_ARG;
  r->parts[{index_num, part}] = ix;
  return SMR_OK;
}

namespace {
void write_fx(smr_report* r, Out& f, const char* header, const char* seq, const char* qual) {      // the record as read (report_fx_base.cpp:176-181)
  if (!f.is_open()) return;
  std::string t(header); t += '\n'; t += seq; t += '\n';
  if (r->fastq) { t += "+\n"; t += qual ? qual : ""; t += '\n'; }
  f.put(t);
}
int add_rows(smr_report* r, const char* header, const char* seq, const char* qual, const std::vector<Aln>& alns);
}  // namespace

extern "C" int smr_report_add(smr_report* r, const char* header, const char* seq, const char* qual, const uint8_t* record, size_t record_len) {
  if (!r || !header || !seq) return SMR_ERR_ARG;
  bool is_hit = false;
  std::vector<Aln> alns;
  if (!parse_record(record, record_len, is_hit, alns)) { r->err = "malformed record"; return SMR_ERR_ARG; }
  write_fx(r, is_hit ? r->f_aligned[0] : r->f_other[0], header, seq, qual);
  return add_rows(r, header, seq, qual, alns);
}

extern "C" int smr_report_add_pair(smr_report* r, const char* header1, const char* seq1, const char* qual1, const uint8_t* record1, size_t record1_len,
                                   const char* header2, const char* seq2, const char* qual2, const uint8_t* record2, size_t record2_len) {
  if (!r || !header1 || !seq1 || !header2 || !seq2) return SMR_ERR_ARG;
  bool hit[2] = {false, false};
  std::vector<Aln> alns[2];
  if (!parse_record(record1, record1_len, hit[0], alns[0]) || !parse_record(record2, record2_len, hit[1], alns[1])) { r->err = "malformed record"; return SMR_ERR_ARG; }
  const char* hd[2] = {header1, header2}; const char* sq[2] = {seq1, seq2}; const char* ql[2] = {qual1, qual2};
  const bool both = hit[0] && hit[1], any = hit[0] || hit[1];
  const smr_report_opts& o = r->o;
  // aligned.* (ReportFastx::append): nothing when neither mate aligned
  if (any) {
    for (int i = 0; i < 2; i++) {
      int idx = -1;
      if (r->num_out == 1) { if (o.paired_out ? both : (o.paired_in || hit[i])) idx = 0; }
      else if (r->num_out == 2 && o.out2) { if (o.paired_out) { if (!both) break; idx = i; } else if (o.paired_in || hit[i]) idx = i; }
      else if (r->num_out == 2) { if (both) idx = 0; else if (hit[i]) idx = 1; }                    // sout: pairs | singletons
      else { if (both) idx = i; else if (hit[i]) idx = i + 2; }
      if (idx >= 0) write_fx(r, r->f_aligned[idx], hd[i], sq[i], ql[i]);
    }
  }
  // other.* (ReportFxOther::append): nothing when both mates aligned
  if (!both) {
    for (int i = 0; i < 2; i++) {
      int idx = -1;
      if (r->num_out == 1) { if (o.paired_in ? !any : (o.paired_out || !hit[i])) idx = 0; }
      else if (r->num_out == 2 && o.out2) { if (o.paired_in) { if (any) break; idx = i; } else if (o.paired_out || !hit[i]) idx = i; }
      else if (r->num_out == 2) { if (!any) idx = 0; else if (!hit[i]) idx = 1; }
      else { if (!any) idx = i; else if (!hit[i]) idx = i + 2; }
      if (idx >= 0) write_fx(r, r->f_other[idx], hd[i], sq[i], ql[i]);
    }
  }
  for (int i = 0; i < 2; i++) { const int rc = add_rows(r, hd[i], sq[i], ql[i], alns[i]); if (rc != SMR_OK) return rc; }
  return SMR_OK;
}

namespace {
int add_rows(smr_report* r, const char* header, const char* seq, const char* qual, const std::vector<Aln>& alns) {
  if (alns.empty() || (!r->o.blast_tabular && !r->o.blast_pairwise && !r->o.sam)) return SMR_OK;
  // Read::getSeqId (read.cpp:371-377)
  std::string id(header);
  id = id.substr(0, id.find(' '));
  size_t k0 = 0; while (k0 < id.size() && (id[k0] == '>' || id[k0] == '@')) k0++;
  id = id.substr(k0);
  const size_t len = strlen(seq);
  // the read in the 0..4 alphabet (flip34 to 04: ambiguous letters are 4), forward and reverse-complement
  std::string fwd(len, 0), rev(len, 0);
  for (size_t i = 0; i < len; i++) fwd[i] = (char)nt_code((unsigned char)seq[i]);
  for (size_t i = 0; i < len; i++) { const int c = fwd[len - 1 - i]; rev[i] = (char)(c == 4 ? 4 : 3 - c); }
  static const char nt_map[5] = {'A', 'C', 'G', 'T', 'N'};
  std::map<std::pair<uint32_t, uint32_t>, std::string> quals;     // ReportSam reverses read.quality IN PLACE per reverse alignment (report_sam.cpp:123-127)
  for (const Aln& a : alns) {
    const std::pair<uint32_t, uint32_t> key{a.index_num, a.part};
    auto pit = r->parts.find(key);
    auto dit = r->dbs.find(a.index_num);
    if (pit == r->parts.end() || dit == r->dbs.end()) { r->err = "alignment refers to an (index, part) that was not registered"; return SMR_ERR_STATE; }
    const smr_index* ix = pit->second;
    if (a.ref_num >= ix->n_refs()) { r->err = "ref_num out of range"; return SMR_ERR_ARG; }
    size_t first_seq = 0;
    for (uint32_t q = 0; q < a.part && q < ix->parts.size(); q++) first_seq += ix->parts[q].numseq_part;
    const std::string ref_id = first_seq + a.ref_num < ix->sq_header.size() ? ix->sq_header[first_seq + a.ref_num].first : std::string("*");
    const uint8_t* refseq = ix->ref_seq.data() + ix->ref_off[a.ref_num];
    const std::string& iseq = a.strand ? fwd : rev;           // `if (align.strand == read.reversed) read.revIntStr()`
    // Read::calc_miss_gap_match (read.cpp:547-589)
    uint32_t n_miss = 0, n_gap = 0, n_match = 0;
    {
      int64_t qb = a.ref_begin1, pb = a.read_begin1;
      for (uint32_t c : a.cigar) {
        const uint32_t letter = c & 0xF, length = c >> 4;
        if (letter == 0) { for (uint32_t u = 0; u < length; u++) { if ((char)refseq[qb] != iseq[pb]) ++n_miss; else ++n_match; ++qb; ++pb; } }
        else if (letter == 1) { pb += length; n_gap += length; }
        else { qb += length; n_gap += length; }
      }
    }
    const double idf = (double)n_match / (double)(n_miss + n_gap + n_match);
    const double cov = (double)std::abs(a.read_end1 - a.read_begin1 + 1) / (double)a.readlen;
    if (r->o.blast_pairwise && !r->o.blast_tabular) {
      // The alignment as rows of at most 60 columns: reference letters ('-' where the read has an insertion), match marks,
      // read letters ('-' where the read has a deletion); the numbers are the 1-based first / last position of the row.
      const Db& d = dit->second;
      const uint32_t bitscore = (uint32_t)((float)(d.lambda * a.score1 - std::log(d.K)) / (float)std::log(2));
      const double evalue = (double)d.K * d.full_ref * d.full_read * std::exp(-d.lambda * a.score1);
      std::ostringstream ss;
      ss << "Sequence ID: " << ref_id << "\n" << "Query ID: " << id << "\n";
      ss << "Score: " << a.score1 << " bits (" << bitscore << ")\t";
      ss.precision(3);
      ss << "Expect: " << evalue << "\t" << "strand: " << (a.strand ? '+' : '-') << "\n\n";
      std::string ops;                                     // one char per alignment column: 0 = M, 1 = I, 2 = D
      for (uint32_t c : a.cigar) ops.append(c >> 4, (char)(c & 0xF));
      int64_t q = a.ref_begin1, p = a.read_begin1;
      for (size_t c0 = 0; c0 < ops.size(); c0 += 60) {
        const size_t c1 = std::min(ops.size(), c0 + 60);
        std::string tl, ml, ql;
        const int64_t q0 = q, p0 = p;
        for (size_t c = c0; c < c1; c++) {
          const char op = ops[c];
          if (op == 0) {
            const char rc = nt_map[refseq[q]], qc = nt_map[(int)iseq[p]];
            tl += rc; ql += qc; ml += rc == qc ? '|' : '*';
            ++q; ++p;
          } else if (op == 1) { tl += '-'; ml += ' '; ql += nt_map[(int)iseq[p]]; ++p; }
          else { tl += nt_map[refseq[q]]; ml += ' '; ql += '-'; ++q; }
        }
        ss << "Target: " << std::setw(8) << q0 + 1 << "    " << tl << "    " << q << "\n";
        ss << std::setw(20) << " " << ml;
        ss << "\nQuery: " << std::setw(9) << p0 + 1 << "    " << ql << "    " << p << "\n\n";
      }
      r->blast[key] += ss.str();
    }
    if (r->o.blast_tabular) {
      const Db& d = dit->second;
      const uint32_t bitscore = (uint32_t)((float)(d.lambda * a.score1 - std::log(d.K)) / (float)std::log(2));
      const double evalue = (double)d.K * d.full_ref * d.full_read * std::exp(-d.lambda * a.score1);
      std::ostringstream ss;
      ss << id << "\t" << ref_id << "\t";
      ss.precision(3);
      ss << idf * 100 << "\t" << (a.read_end1 - a.read_begin1 + 1) << "\t" << n_miss << "\t" << n_gap << "\t" << a.read_begin1 + 1 << "\t" << a.read_end1 + 1
         << "\t" << a.ref_begin1 + 1 << "\t" << a.ref_end1 + 1 << "\t" << evalue << "\t" << bitscore;
      // optional columns in the order they were requested ("cigar", "qcov", "qstrand")
      for (const char* p = r->o.blast_cols; *p;) {
        const char* e = strchr(p, ' '); const std::string col = e ? std::string(p, e) : std::string(p);
        if (col == "cigar") ss << "\t" << cigar_text(a, len);
        else if (col == "qcov") { ss.precision(3); ss << "\t" << cov * 100; }
        else if (col == "qstrand") ss << "\t" << (a.strand ? '+' : '-');
        p = e ? e + 1 : p + col.size();
      }
      ss << "\n";
      r->blast[key] += ss.str();
    }
    if (r->o.sam) {
      std::ostringstream ss;
      ss << id << (a.strand ? "\t0\t" : "\t16\t") << ref_id << "\t" << a.ref_begin1 + 1 << "\t255\t" << cigar_text(a, len) << "\t*\t0\t0\t";
      for (size_t i = 0; i < len; i++) ss << nt_map[(int)iseq[i]];
      ss << "\t";
      if (qual && *qual) {
        auto qit = quals.find(key);
        if (qit == quals.end()) qit = quals.emplace(key, std::string(qual)).first;
        if (!a.strand) std::reverse(qit->second.begin(), qit->second.end());
        ss << qit->second;
      } else ss << "*";
      ss << "\tAS:i:" << a.score1 << "\tNM:i:" << n_miss + n_gap << "\n";
      r->sam[key] += ss.str();
    }
  }
  return SMR_OK;
}
}  // namespace

extern "C" int smr_report_close(smr_report* r) {
  if (!r) return SMR_ERR_ARG;
  int rc = SMR_OK;
  for (int j = 0; j < 4; j++) { r->f_aligned[j].close(); r->f_other[j].close(); }
  if (r->o.blast_tabular || r->o.blast_pairwise) {
    Out f;
    if (!f.open(r->dir + "/aligned.blast", r->o.zip_out != 0)) rc = SMR_ERR_IO; else { for (auto& kv : r->blast) f.put(kv.second); f.close(); }
  }
  if (r->o.sam) {
    Out f;
    if (!f.open(r->dir + "/aligned.sam", r->o.zip_out != 0)) rc = SMR_ERR_IO;
    else {
      f.put(std::string("@HD\tVN:1.0\tSO:unsorted\n"));
      if (r->o.sam_sq) {                                  // every sequence of every --ref, in --ref order (from <index>.stats)
        uint32_t last = 0xFFFFFFFFu;
        for (auto& kv : r->parts) {
          if (kv.first.first == last) continue;
          last = kv.first.first;
          for (auto& sq : kv.second->sq_header) f.put("@SQ\tSN:" + sq.first + "\tLN:" + std::to_string(sq.second) + "\n");
        }
      }
      f.put("@PG\tID:sortmerna\tVN:1.0\tCL:" + r->cmdline + "\n");
      for (auto& kv : r->sam) f.put(kv.second);
      f.close();
    }
  }
  delete r;
  return rc;
}

extern "C" int smr_report_set_cmdline(smr_report* r, const char* cmdline) {
  if (!r || !cmdline) return SMR_ERR_ARG;
  r->cmdline = cmdline;
  return SMR_OK;
}

extern "C" int smr_summary_write(const char* path, const smr_summary* s) {
  if (!path || !s || (s->n_dbs && !s->dbs) || (s->n_reads_files && !s->reads_files)) return SMR_ERR_ARG;
  std::ostringstream ss;
  ss << " Command:\n    " << (s->cmdline ? s->cmdline : "") << "\n\n" << " Process pid = " << (s->pid ? s->pid : "") << "\n\n" << " Parameters summary: \n";
  for (uint32_t i = 0; i < s->n_dbs; i++) {
    const smr_summary_db& d = s->dbs[i];
    ss << "    Reference file: " << (d.ref_file ? d.ref_file : "") << "\n"
       << "        Seed length = " << s->seed_len << "\n"
       << "        Pass 1 = " << d.skiplengths[0] << ", Pass 2 = " << d.skiplengths[1] << ", Pass 3 = " << d.skiplengths[2] << "\n"
       << "        Gumbel lambda = " << d.lambda << "\n"
       << "        Gumbel K = " << d.K << "\n"
       << "        Minimal SW score based on E-value = " << d.minimal_score << "\n";
  }
  ss << "    Number of seeds = " << s->num_seeds << "\n" << "    Edges = " << s->edges << "\n" << "    SW match = " << s->match << "\n"
     << "    SW mismatch = " << s->mismatch << "\n" << "    SW gap open penalty = " << s->gap_open << "\n"
     << "    SW gap extend penalty = " << s->gap_ext << "\n" << "    SW ambiguous nucleotide = " << s->score_N << "\n"
     << "    SQ tags are " << (s->sam_sq ? "" : "not ") << "output\n"
     << "    Number of alignment processing threads = " << s->threads << "\n";
  for (uint32_t i = 0; i < s->n_reads_files; i++) ss << "    Reads file: " << s->reads_files[i] << "\n";
  ss << "    Total reads = " << s->total_reads << "\n\n" << " Results:\n";
  const float ratio = (float)s->num_aligned / s->total_reads;
  ss << std::setprecision(2) << std::fixed
     << "    Total reads passing E-value threshold = " << s->num_aligned << " (" << (ratio * 100) << ")\n"
     << "    Total reads failing E-value threshold = " << s->total_reads - s->num_aligned << " (" << (1 - ratio) * 100 << ")\n"
     << "    Minimum read length = " << s->min_read_len << "\n" << "    Maximum read length = " << s->max_read_len << "\n"
     << "    Mean read length    = " << (s->total_reads ? s->all_reads_len / s->total_reads : 0) << "\n\n" << " Coverage by database:\n";
  for (uint32_t i = 0; i < s->n_dbs; i++) {
    const float pcn = (float)((float)s->dbs[i].reads_matched / s->total_reads) * 100;
    ss << "    " << (s->dbs[i].ref_file ? s->dbs[i].ref_file : "") << "\t\t" << pcn << "\n";
  }
  ss << "\n " << (s->timestamp ? s->timestamp : "") << "\n";
  FILE* f = fopen(path, "wb");
  if (!f) return SMR_ERR_IO;
  const std::string t = ss.str();
  fwrite(t.data(), 1, t.size(), f);
  fclose(f);
  return SMR_OK;
}

extern "C" const char* smr_report_last_error(const smr_report* r) { return r ? r->err.c_str() : "null report"; }


// ------------------------------------------------------------------------------------------------
// Readstats persistence: what the reference keeps in its KVDB next to the per-read records after the alignment stage --
// Readstats::store_to_db (readstats.cpp:291-295) puts Readstats::toBstring() (:133-174) under the decimal std::hash of the '_'-joined
// basenames of the read files (:82-91, util.cpp:216-222).  The identity / coverage counters and is_stats_calc belong to later stages of
// the reference and are zero / false here; is_set_aligned_id_cov stays false because n_yid_ycov is 0 (readstats.cpp:199-203).
// ------------------------------------------------------------------------------------------------
extern "C" size_t smr_readstats_record(uint64_t all_reads_count, uint64_t all_reads_len, uint32_t min_read_len, uint32_t max_read_len, uint64_t num_aligned,
                                       uint64_t num_short, const uint64_t* reads_matched_per_db, uint32_t n_db, uint8_t* buf, size_t cap) {
  const size_t need = 8 + 8 + 4 + 4 + 6 * 8 + 8 + (size_t)n_db * 8 + 2;
  if (!buf || cap < need) return need;
  uint8_t* p = buf;
  auto put = [&](const void* v, size_t n) { memcpy(p, v, n); p += n; };
  const uint64_t zero = 0, ndb = n_db;
  put(&all_reads_count, 8); put(&all_reads_len, 8); put(&min_read_len, 4); put(&max_read_len, 4);
  put(&num_aligned, 8);
  put(&zero, 8); put(&zero, 8); put(&zero, 8); put(&zero, 8);          // n_yid_ncov, n_nid_ycov, n_yid_ycov, num_denovo
  put(&num_short, 8);
  put(&ndb, 8);
  for (uint32_t i = 0; i < n_db; i++) put(&reads_matched_per_db[i], 8);
  const uint8_t f = 0;
  put(&f, 1); put(&f, 1);                                              // is_stats_calc, is_set_aligned_id_cov
  return need;
}

extern "C" size_t smr_readstats_key(const char* const* reads_files, uint32_t n_files, char* buf, size_t cap) {
  std::string joined;
  for (uint32_t i = 0; i < n_files; i++) {
    std::string f = reads_files[i] ? reads_files[i] : "";
    const size_t sl = f.find_last_of('/');
    if (sl != std::string::npos) f = f.substr(sl + 1);
    joined += (i ? "_" : "") + f;
  }
  const std::string key = std::to_string(std::hash<std::string>{}(joined));
  if (buf && cap > key.size()) memcpy(buf, key.c_str(), key.size() + 1);
  return key.size();
}
