// align_gpu.cpp -- TEST INFRASTRUCTURE (lives under oracle/ because it is compiled and linked WITH the reference's sources).
//
// The binding of INTEGRATION.md, compiled: a replacement of the reference's align() (/root/reference/src/sortmerna/processor.cpp:173-285)
// that keeps everything around the hot path -- the reference's own CLI and option parsing, its indexer, Readfeed, Refstats (incl. the
// ALP Gumbel parameters), KVDB, summary and report writers -- and sends the per-read work (the N x align2() threads of every
// (index, part), processor.cpp:93-168,248-256) through the C ABI of libsmr_hip (include/smr_hip.h).  oracle/Makefile (target `dropin`)
// compiles the reference's processor.cpp with its own align() renamed out of the way and links this file in its place:
//     oracle/_ref/sortmerna_gpu  =  the reference with the GPU in the middle.
// tests/test_dropin.py runs it next to the unmodified binary on the reference's test inputs and compares every output file.
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"
#include "index.hpp"
#include "kvdb.hpp"
#include "options.hpp"
#include "processor.hpp"
#include "read.hpp"
#include "readfeed.hpp"
#include "readstats.hpp"
#include "refstats.hpp"

#include "smr_hip.h"

namespace {
[[noreturn]] void die_gpu(smr_ctx* c, const char* what) { ERR(what, ": ", smr_last_error(c)); exit(EXIT_FAILURE); }   // the reference's error convention

// one chunk of the read file(s) on its way through the stages
struct Chunk {
	std::vector<std::string> ids;                            // Read::id = the KVDB key                       readfeed.cpp:793
	smr_reads* packed = nullptr;
	std::vector<std::pair<size_t, std::string>> records;     // (position in ids, Read::toBinString bytes)
};
template <class T> struct Queue {                            // hand-over between the stages, at most `cap` chunks waiting (a fast reader or a slow KVDB must
	std::mutex m; std::condition_variable cv, cv_room; std::deque<T> q; size_t cap = 4;      //  not hold the whole read set in memory); a null pointer ends it
	void push(T v) { { std::unique_lock<std::mutex> l(m); cv_room.wait(l, [&] { return q.size() < cap; }); q.push_back(std::move(v)); } cv.notify_one(); }
	T pop() { T v; { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !q.empty(); }); v = std::move(q.front()); q.pop_front(); } cv_room.notify_one(); return v; }
};
}

// The reads STREAM through the GPUs like they stream through the reference's aligner threads (readfeed.cpp:776-873, processor.cpp:248-256):
//   reader thread    Readfeed::next() exactly as align2() walks it (processor.cpp:103-104,160) -> chunks of SMR_DROPIN_CHUNK reads
//                    (default 2 M), packed 2 bit per letter
//   one worker per visible GPU (SMR_DROPIN_GPUS limits them): its own smr_ctx with every index part resident; per chunk: upload,
//                    the (index, part) loop of processor.cpp:219-277, traceback, fetch
//   writer thread    kvdb.put(read.id, record) (processor.cpp:150-155)
// A read's state crosses index parts, so a chunk stays on its GPU for all of them; nothing crosses GPUs but the Readstats counters,
// which are summed on the host (this is one process; the multi-process hosts reduce them with RCCL).
void align(Readfeed& readfeed, Readstats& readstats, Index& index, KeyValueDatabase& kvdb, Runopts& opts)
{
	(void)index;                                             // the host-side Index object is not needed: the parts are loaded by smr_index_load_files
	INFO("==== Starting alignment (libsmr_hip) ====");
	readfeed.init_reading();
	Refstats refstats(opts, readstats);                      // unchanged: .stats, Gumbel (ALP), minimal_score   refstats.cpp:81
	char err[512] = "";
	int n_gpu = smr_device_count();
	if (n_gpu <= 0) { ERR("no HIP device (libsmr_hip has no CPU fallback)"); exit(EXIT_FAILURE); }
	if (const char* e = getenv("SMR_DROPIN_GPUS")) n_gpu = std::max(1, std::min(n_gpu, atoi(e)));
	const size_t chunk_reads = getenv("SMR_DROPIN_CHUNK") ? std::max<size_t>(2, strtoull(getenv("SMR_DROPIN_CHUNK"), nullptr, 10) & ~(size_t)1) : 2000000;
	const uint32_t slots = opts.num_alignments > 0 ? (uint32_t)opts.num_alignments : 256;

	// the index parts, once on the host (replaces index.load() + refs.load()  index.cpp:143, references.cpp:55); every GPU keeps them all resident
	struct Part { smr_index* ix; smr_params p; };
	std::vector<Part> parts;
	for (size_t idx_num = 0; idx_num < opts.indexfiles.size(); ++idx_num)
		for (uint16_t part = 0; part < refstats.num_index_parts[idx_num]; ++part) {
			Part P;
			if (smr_index_load_files(opts.indexfiles[idx_num].second.c_str(), part, opts.indexfiles[idx_num].first.c_str(), &P.ix, err, sizeof err) != SMR_OK) { ERR(err); exit(EXIT_FAILURE); }
			smr_params& p = P.p;
			smr_params_default(&p);
			p.num_seeds = opts.num_seeds;      p.min_lis = opts.min_lis;          p.edges = opts.edges;   p.is_as_percent = opts.is_as_percent;
			p.match = opts.match;              p.mismatch = opts.mismatch;        p.score_N = opts.score_N;
			p.gap_open = opts.gap_open;        p.gap_ext = opts.gap_extension;
			p.num_alignments = (uint32_t)opts.num_alignments;   p.is_best = opts.is_best;   p.is_full_search = opts.is_full_search;
			p.is_forward = opts.is_forward;    p.is_reverse = opts.is_reverse;    p.minoccur = opts.minoccur;
			for (int k = 0; k < 3; ++k) p.skiplengths[k] = opts.skiplengths[idx_num][k];
			p.minimal_score = refstats.minimal_score[idx_num];   // derived from the GLOBAL read totals           refstats.cpp:261-265
			p.index_num = (uint32_t)idx_num;  p.part = part;
			p.is_last_index_part = (idx_num == opts.indexfiles.size() - 1 && part == refstats.num_index_parts[idx_num] - 1);
			if (const char* why = smr_params_refused(&p)) { ERR(std::string("options outside what libsmr_hip aligns: ") + why); exit(EXIT_FAILURE); }   // (said before any read is fed)
			parts.push_back(P);
		}
	// the engine keeps up to 64 parts resident on a GPU; a run with more (a small -m, many -ref) streams them through slot 0 per chunk instead
	const size_t max_resident = getenv("SMR_DROPIN_MAX_RESIDENT") ? (size_t)std::min(64, std::max(0, atoi(getenv("SMR_DROPIN_MAX_RESIDENT")))) : 64;   // (the variable: test aid)
	const bool resident = parts.size() <= max_resident;

	Queue<std::unique_ptr<Chunk>> to_gpu, to_db;
	to_gpu.cap = to_db.cap = (size_t)n_gpu + 2;
	std::mutex cm;
	std::vector<uint64_t> ctr(2 + opts.indexfiles.size(), 0);
	uint64_t n_reads = 0, n_chunks = 0;

	// ---- reader ----
	std::thread reader([&] {
		std::unique_ptr<Chunk> ch(new Chunk);
		std::string seqs;
		std::vector<uint64_t> offs{0};
		auto flush = [&]() {
			if (ch->ids.empty()) return;
			if (smr_reads_pack(seqs.data(), offs.data(), (uint32_t)ch->ids.size(), &ch->packed) != SMR_OK) { ERR("smr_reads_pack failed"); exit(EXIT_FAILURE); }
			n_reads += ch->ids.size(); n_chunks++;
			to_gpu.push(std::move(ch));
			ch.reset(new Chunk); seqs.clear(); offs.assign(1, 0);
		};
		for (int id = 0; id < opts.num_proc_thread; ++id) {
			int idx = id * readfeed.num_sense;
			std::string rec;
			for (; readfeed.next(idx, rec);) {
				Read read(rec);                              // parses "id\nheader\nseq[\nqual]"            read.cpp:147
				if (!read.isEmpty) { ch->ids.push_back(read.id); seqs += read.sequence; offs.push_back(seqs.size()); }
				rec.resize(0);
				if (opts.is_paired) idx ^= 1;
				if (ch->ids.size() >= chunk_reads && (!opts.is_paired || (ch->ids.size() & 1) == 0)) flush();
			}
		}
		flush();
		for (int g = 0; g < n_gpu; g++) to_gpu.push(nullptr);
	});

	// ---- one worker per GPU ----
	std::vector<std::thread> workers;
	for (int g = 0; g < n_gpu; g++) workers.emplace_back([&, g] {
		char e2[512] = "";
		smr_ctx* gpu = nullptr;
		if (smr_create(g, &gpu, e2, sizeof e2) != SMR_OK) { ERR(e2); exit(EXIT_FAILURE); }   // no CPU fallback
		if (resident) for (size_t k = 0; k < parts.size(); k++) if (smr_index_upload(gpu, parts[k].ix, (int)k) != SMR_OK) die_gpu(gpu, "smr_index_upload");
		std::vector<uint64_t> mine(ctr.size(), 0), c1(ctr.size());
		std::vector<uint8_t> buf;
		for (;;) {
			std::unique_ptr<Chunk> ch = to_gpu.pop();
			if (!ch) break;
			if (smr_reads_upload(gpu, ch->packed, slots) != SMR_OK) die_gpu(gpu, "smr_reads_upload");
			for (size_t k = 0; k < parts.size(); k++) {      // the (index, part) loop of processor.cpp:219-277 for this chunk
				const int sl = resident ? (int)k : 0;
				if (!resident && smr_index_upload(gpu, parts[k].ix, 0) != SMR_OK) die_gpu(gpu, "smr_index_upload");
				if (smr_align_part(gpu, sl, &parts[k].p) != SMR_OK) die_gpu(gpu, "smr_align_part");    // = the N x align2() threads of this part
				if (smr_traceback(gpu, sl, &parts[k].p)  != SMR_OK) die_gpu(gpu, "smr_traceback");     // CIGARs (ssw.c:577-773)
				if (!resident && smr_index_unload(gpu, 0) != SMR_OK) die_gpu(gpu, "smr_index_unload");
			}
			if (smr_results_fetch(gpu) != SMR_OK) die_gpu(gpu, "smr_results_fetch");
			for (uint32_t i = 0; i < ch->ids.size(); ++i) {
				const size_t n = smr_result_record(gpu, i, nullptr, 0);    // Read::toBinString() bytes, 0 = read has no alignment
				if (!n) continue;
				buf.resize(n);
				smr_result_record(gpu, i, buf.data(), n);
				ch->records.emplace_back(i, std::string(buf.begin(), buf.end()));
			}
			if (smr_counters(gpu, c1.data(), (uint32_t)opts.indexfiles.size()) != SMR_OK) die_gpu(gpu, "smr_counters");
			for (size_t q = 0; q < c1.size(); q++) mine[q] += c1[q];
			smr_reads_free(ch->packed); ch->packed = nullptr;
			to_db.push(std::move(ch));
		}
		{ std::lock_guard<std::mutex> l(cm); for (size_t q = 0; q < ctr.size(); q++) ctr[q] += mine[q]; }
		smr_destroy(gpu);
	});

	// ---- writer: the same KVDB values the reference writes (processor.cpp:150-155) ----
	std::thread writer([&] {
		for (;;) {
			std::unique_ptr<Chunk> ch = to_db.pop();
			if (!ch) break;
			for (auto& r : ch->records) kvdb.put(ch->ids[r.first], r.second);
		}
	});
	reader.join();
	for (auto& w : workers) w.join();
	to_db.push(nullptr);
	writer.join();

	readstats.num_aligned = ctr[0];
	readstats.num_short = ctr[1];
	for (size_t i = 0; i < opts.indexfiles.size(); ++i) readstats.reads_matched_per_db[i] = ctr[2 + i];
	INFO("==== Done alignment on ", n_gpu, " GPU(s): ", n_reads, " reads in ", n_chunks, " chunk(s), ", ctr[0], " aligned ====\n");
	readstats.set_is_set_aligned_id_cov();
	readstats.store_to_db(kvdb);                             // processor.cpp:283-284
	for (auto& P : parts) smr_index_free(P.ix);
}
