// align_gpu.cpp -- TEST INFRASTRUCTURE (lives under oracle/ because it is compiled and linked WITH the reference's sources).
//
// The binding of INTEGRATION.md, compiled: a replacement of the reference's align() (/root/reference/src/sortmerna/processor.cpp:173-285)
// that keeps everything around the hot path -- the reference's own CLI and option parsing, its indexer, Readfeed, Refstats (incl. the
// ALP Gumbel parameters), KVDB, summary and report writers -- and sends the per-read work (the N x align2() threads of every
// (index, part), processor.cpp:93-168,248-256) through the C ABI of libsmr_hip (include/smr_hip.h).  oracle/Makefile (target `dropin`)
// compiles the reference's processor.cpp with its own align() renamed out of the way and links this file in its place:
//     oracle/_ref/sortmerna_gpu  =  the reference with the GPU in the middle.
// tests/test_dropin_gpu.py runs it next to the unmodified binary on the reference's test inputs and compares every output file.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
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
}

void align(Readfeed& readfeed, Readstats& readstats, Index& index, KeyValueDatabase& kvdb, Runopts& opts)
{
	(void)index;                                             // the host-side Index object is not needed: the parts are loaded by smr_index_load_files
	INFO("==== Starting alignment (libsmr_hip) ====");
	readfeed.init_reading();
	Refstats refstats(opts, readstats);                      // unchanged: .stats, Gumbel (ALP), minimal_score   refstats.cpp:81
	char err[512] = "";
	smr_ctx* gpu = nullptr;
	if (smr_create(/*device*/0, &gpu, err, sizeof err) != SMR_OK) { ERR(err); exit(EXIT_FAILURE); }   // no CPU fallback

	// 1. reads: Readfeed::next() exactly as align2() walks it (processor.cpp:103-104,160), so read ids / KVDB keys are the reference's
	std::vector<std::string> ids;
	std::string seqs;
	std::vector<uint64_t> offs{0};
	for (int id = 0; id < opts.num_proc_thread; ++id) {
		int idx = id * readfeed.num_sense;
		std::string rec;
		for (; readfeed.next(idx, rec);) {
			Read read(rec);                                  // parses "id\nheader\nseq[\nqual]"            read.cpp:147
			if (!read.isEmpty) { ids.push_back(read.id); seqs += read.sequence; offs.push_back(seqs.size()); }
			rec.resize(0);
			if (opts.is_paired) idx ^= 1;
		}
	}
	smr_reads* batch = nullptr;
	if (smr_reads_pack(seqs.data(), offs.data(), (uint32_t)ids.size(), &batch) != SMR_OK) { ERR("smr_reads_pack failed"); exit(EXIT_FAILURE); }
	const uint32_t slots = opts.num_alignments > 0 ? (uint32_t)opts.num_alignments : 256;
	if (smr_reads_upload(gpu, batch, slots) != SMR_OK) die_gpu(gpu, "smr_reads_upload");

	// 2. the (index, part) loop of processor.cpp:219-277
	for (size_t idx_num = 0; idx_num < opts.indexfiles.size(); ++idx_num)
		for (uint16_t part = 0; part < refstats.num_index_parts[idx_num]; ++part) {
			smr_index* ix = nullptr;                         // replaces index.load() + refs.load()       index.cpp:143, references.cpp:55
			if (smr_index_load_files(opts.indexfiles[idx_num].second.c_str(), part, opts.indexfiles[idx_num].first.c_str(), &ix, err, sizeof err) != SMR_OK) { ERR(err); exit(EXIT_FAILURE); }
			if (smr_index_upload(gpu, ix, /*slot*/0) != SMR_OK) die_gpu(gpu, "smr_index_upload");
			smr_params p;
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
			if (smr_align_part(gpu, 0, &p) != SMR_OK) die_gpu(gpu, "smr_align_part");    // = the N x align2() threads of this part
			if (smr_traceback(gpu, 0, &p)  != SMR_OK) die_gpu(gpu, "smr_traceback");     // CIGARs (ssw.c:577-773)
			smr_index_unload(gpu, 0);
			smr_index_free(ix);
			INFO("done index: ", idx_num, " part: ", part + 1, " on the GPU");
		}

	// 3. results -> the same KVDB values the reference writes (processor.cpp:150-155), counters -> Readstats
	if (smr_results_fetch(gpu) != SMR_OK) die_gpu(gpu, "smr_results_fetch");
	std::vector<uint8_t> buf;
	for (uint32_t i = 0; i < ids.size(); ++i) {
		const size_t n = smr_result_record(gpu, i, nullptr, 0);    // Read::toBinString() bytes, 0 = read has no alignment
		if (!n) continue;
		buf.resize(n);
		smr_result_record(gpu, i, buf.data(), n);
		kvdb.put(ids[i], std::string(buf.begin(), buf.end()));
	}
	std::vector<uint64_t> ctr(2 + opts.indexfiles.size());
	if (smr_counters(gpu, ctr.data(), (uint32_t)opts.indexfiles.size()) != SMR_OK) die_gpu(gpu, "smr_counters");
	readstats.num_aligned = ctr[0];
	readstats.num_short = ctr[1];
	for (size_t i = 0; i < opts.indexfiles.size(); ++i) readstats.reads_matched_per_db[i] = ctr[2 + i];
	INFO("==== Done alignment on the GPU: ", ids.size(), " reads, ", ctr[0], " aligned ====\n");
	readstats.set_is_set_aligned_id_cov();
	readstats.store_to_db(kvdb);                             // processor.cpp:283-284
	smr_reads_free(batch);
	smr_destroy(gpu);
}
