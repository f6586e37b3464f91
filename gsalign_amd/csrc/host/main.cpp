// gsalign_amd/csrc/host/main.cpp -- GSAlign_hip: command-line front end with the
// reference's surface (reference src/main.cpp:14-334): same flags, same defaults
// (the CODE's defaults: -clr 200, -alen 200), same output files, with the per-contig
// hot path handed to libgsa_hip.so (gsa_align_contig) instead of GenomeComparison's
// pthread stages.  Extra flags: -gpu LIST (device ordinals, comma separated: the query contigs shard over them with the
// index replicated), -ctx N (contexts per GPU: gsa_clone, contigs overlap on one device) and -timing
// (one JSON line on stderr: where the wall time of the run went).
// The contigs go through gsa_align_many (the per-contig loop of GSAlign.cpp:483-548).  Round 5: a GPU worker thread only COPIES a finished
// contig out of the library's memory; ONE formatter thread takes the contigs in contig order (OutputMAF appends per contig, VarVec grows in
// contig order: GSAlign.cpp:543-546) and formats each with the host pool's threads (par.h: the text lines of a block, the variants of a
// block's records), a writer thread writes the buffers in order -- so the output bytes depend neither on GPUs / contexts nor on -t.
#include <errno.h>
#include <malloc.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <fcntl.h>
#include <unistd.h>
#include "gsa_host.h"
#include "par.h"

static void usage(const char *prog, int t, const gsa_params &p, int fmt)
{
	fprintf(stderr, "\nGenAlign v%s\n", "1.0.22");
	fprintf(stderr, "Usage: %s [-i IndexFile Prefix / -r Reference file] -q QueryFile[Fasta]\n\n", prog);
	fprintf(stderr, "Options: -t     INT     number of threads [%d] (host side: FASTA / index loading, MAF / VCF formatting; the hot path runs on the GPU)\n", t);
	fprintf(stderr, "         -o     STR     Set the prefix of the output files [output]\n");
	fprintf(stderr, "         -fmt   INT     Set the output format 1:maf, 2:aln [%d]\n", fmt);
	fprintf(stderr, "         -idy   INT     Set the minimal sequence identity (0-100) of a local alignment [%d]\n", p.min_identity);
	fprintf(stderr, "         -slen  INT     Set the minimal seed length [%d]\n", p.min_seed_len);
	fprintf(stderr, "         -alen  INT     Set the minimal alignment length [%d]\n", p.min_aln_len);
	fprintf(stderr, "         -ind   INT     Set the maximal indel size [%d]\n", p.max_indel);
	fprintf(stderr, "         -clr   INT     Set the minimal cluster size [%d]\n", p.min_block_score);
	fprintf(stderr, "         -unique        Output unique alignment only [false]\n");
	fprintf(stderr, "         -sen           Sensitive mode [False]\n");
	fprintf(stderr, "         -dp            Output Dot-plots\n");
	fprintf(stderr, "         -one           set one on one aligment mode[false]\n");
	fprintf(stderr, "         -gp    STR     Specify the path of gnuplot\n");
	fprintf(stderr, "         -no_vcf        do not write the VCF file\n");
	fprintf(stderr, "         -gpu   LIST    GPU ordinals, comma separated [0]\n");
	fprintf(stderr, "         -ctx   INT     contexts per GPU working on different query sequences [2]\n");
	fprintf(stderr, "         -timing        print where the wall time went (one JSON line on stderr)\n\n");
}

static bool check_prefix(const char *p)                     // CheckOutputPrefix (main.cpp:116-138)
{
	if (strcmp(p, "/dev/null") == 0) return true;
	for (size_t i = 0; i < strlen(p); i++) {
		const int c = (unsigned char)p[i];
		if (!isprint(c) || (c >= 32 && c <= 44) || (c >= 58 && c <= 64) || (c >= 123 && c <= 127)) { fprintf(stderr, "FatalError: Please specify a valid prefix name\n"); return false; }
	}
	return true;
}

static bool first_char_is_header(const char *path)          // CheckInputFile (main.cpp:49-64)
{
	FILE *fp = fopen(path, "r"); if (!fp) return false;
	int c = fgetc(fp); fclose(fp);
	return c == '>';
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char *argv[])
{
	// -ctx contexts x 5 streams each; the runtime's default is 4 hardware queues per process.  bench.py uses the same value (--hwq); see INTEGRATION.md
	setenv("GPU_MAX_HW_QUEUES", "16", 0);
	// (finished contigs are copied into fresh ~100 MB blocks and freed after they are written: keep that memory in the heap instead of handing it back to the
	//  kernel and faulting it in again for the next contig)
	mallopt(M_MMAP_THRESHOLD, 1 << 30); mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_TOP_PAD, 256 << 20);
	gsa_params prm; gsa_default_params(&prm);
	int threads = HostPool::default_threads(), fmt = 1, n_ctx_per_gpu = 2; bool vcf = true, allow_dup = true, dotplot = false, timing = getenv("GSA_TIMING") != NULL;
	std::vector<int> gpus;
	const char *index_prefix = NULL, *ref_fa = NULL, *query_fa = NULL, *out_prefix = NULL, *gnuplot_arg = NULL;
	if (argc == 1 || strcmp(argv[1], "-h") == 0) { usage(argv[0], threads, prm, fmt); return 0; }
	if (strcmp(argv[1], "index") == 0) {
		if (argc == 4) { std::string e; if (!gsah_build_index(argv[2], argv[3], e)) { fprintf(stderr, "%s\n", e.c_str()); return 1; } }
		else fprintf(stderr, "usage: %s index ref.fa prefix\n", argv[0]);
		return 0;
	}
	for (int i = 1; i < argc; i++) {
		const std::string a = argv[i];
		if (a == "-i") index_prefix = argv[++i];
		else if (a == "-r" && i + 1 < argc) ref_fa = argv[++i];
		else if (a == "-q" && i + 1 < argc) query_fa = argv[++i];
		else if (a == "-t" && i + 1 < argc) { if ((threads = atoi(argv[++i])) < 0) { fprintf(stderr, "Warning! Thread number should be greater than 0!\n"); threads = 16; } }
		else if (a == "-slen" && i + 1 < argc) { prm.min_seed_len = atoi(argv[++i]); if (prm.min_seed_len < 10 || prm.min_seed_len > 30) { fprintf(stderr, "Warning! minimal seed length is between 10~20!\n"); return 0; } }
		else if (a == "-ind" && i + 1 < argc) { prm.max_indel = atoi(argv[++i]); if (prm.max_indel < 10 || prm.max_indel > 100) { fprintf(stderr, "Warning! maximal indel size is between 10~100!\n"); return 0; } }
		else if (a == "-sen" || a == "-sensitive") { prm.sensitive = 1; prm.min_aln_len = 200; prm.min_block_score = 50; }
		else if (a == "-unique") allow_dup = false;
		else if (a == "-no_vcf") vcf = false;
		else if (a == "-one") prm.one_on_one = 1;
		else if (a == "-idy" && i + 1 < argc) prm.min_identity = atoi(argv[++i]);
		else if (a == "-alen" && i + 1 < argc) prm.min_aln_len = atoi(argv[++i]);
		else if (a == "-clr" && i + 1 < argc) prm.min_block_score = atoi(argv[++i]);
		else if (a == "-fmt" && i + 1 < argc) fmt = atoi(argv[++i]);
		else if (a == "-o") out_prefix = argv[++i];
		else if (a == "-gpu" && i + 1 < argc) { for (const char *p = argv[++i]; *p;) { gpus.push_back(atoi(p)); while (*p && *p != ',') p++; if (*p == ',') p++; } }
		else if (a == "-ctx" && i + 1 < argc) { n_ctx_per_gpu = atoi(argv[++i]); if (n_ctx_per_gpu < 1) n_ctx_per_gpu = 1; }
		else if (a == "-timing") timing = true;
		else if (a == "-dp") dotplot = true;
		else if (a == "-gp" && i + 1 < argc) gnuplot_arg = argv[++i];      // main.cpp:285: the path of gnuplot, used as given
		else if (a == "-d" || a == "-debug") { /* debug printers: not reproduced */ }
		else if (a == "-obr" && i + 1 < argc) ++i;
		else fprintf(stderr, "Warning! Unknow parameter: %s\n", argv[i]);
	}
	if ((index_prefix == NULL && ref_fa == NULL) || query_fa == NULL) { usage(argv[0], threads, prm, fmt); return 0; }
	if (out_prefix == NULL) out_prefix = "output"; else if (!check_prefix(out_prefix)) return 0;
	HostPool::global().resize(threads < 1 ? 1 : (threads > 256 ? 256 : threads));

	const time_t t0 = time(NULL);
	const double T0 = now_s();
	double t_query = 0, t_index = 0, t_create = 0, t_align = 0, t_drain = 0, t_vcf = 0, t_maf_fmt = 0, t_var = 0, t_copy = 0, t_build = 0;
	fprintf(stderr, "Step1. Load the two genome sequences...\n");
	std::string err, qerr; std::vector<QueryContig> qs; bool q_ok = false;
	if (!first_char_is_header(query_fa)) { fprintf(stderr, "Please check the query file: %s\n", query_fa); return 0; }
	// the query FASTA is read beside the index (both use the pool: their parallel sections take turns) and beside gsa_create, which is GPU work
	std::thread q_loader([&] { const double t = now_s(); q_ok = gsah_load_query(query_fa, qs, qerr); t_query = now_s() - t; });
	struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } q_join{ q_loader };
	HostIndex idx; std::string prefix;
	if (index_prefix != NULL && gsah_index_files_exist(index_prefix)) prefix = index_prefix;
	else if (ref_fa != NULL && first_char_is_header(ref_fa)) {
		prefix = ref_fa; size_t p = prefix.find_last_of('.'); if (p != std::string::npos && p > 0) prefix.resize(p);
		const double t = now_s();
		if (!gsah_build_index(ref_fa, prefix, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
		t_build = now_s() - t;
	} else { q_loader.join(); if (!q_ok) fprintf(stderr, "Please check the query file: %s\n", query_fa); else fprintf(stderr, "Please specify a valid reference genome\n"); return 0; }
	// while the files are read: the HIP runtime comes up and the device memory of the two largest tables (their sizes follow from the text length in the .bwt header)
	// is set aside -- ~0.4 s of hipMalloc for a human index that gsa_create would otherwise spend AFTER the files are in
	if (gpus.empty()) gpus.push_back(0);
	double t_reserve = 0, t_reserve_wait = 0, t_unpack_wait = 0, t_at_align = 0;
	std::thread reserver([&] {
		const double t = now_s();
		uint64_t hdr[5] = { 0, 0, 0, 0, 0 };
		FILE *fb = fopen((prefix + ".bwt").c_str(), "rb");
		const bool ok = fb && fread(hdr, 8, 5, fb) == 5; if (fb) fclose(fb);
		const char *fw = getenv("GSA_FORCE_WIDE");
		if (ok && hdr[4] > 0) (void)gsa_reserve_index(gpus[0], hdr[4], (fw && *fw && *fw != '0') ? GSA_CREATE_WIDE : 0u);
		t_reserve = now_s() - t;
	});
	struct Joiner3 { std::thread &t; ~Joiner3() { if (t.joinable()) t.join(); } } reserve_join{ reserver };
	// the index files as they lie on disk first (.bwt, .sa, .ann, the raw .pac bytes); RefSequence -- which only the emitters of THIS program read -- is unpacked further
	// down, beside gsa_create: the device gets the .pac bytes and unpacks its own copy (GSA_CREATE_REF_PAC)
	{ const double t = now_s(); const bool ok = gsah_load_index_files(prefix, idx, err); t_index = now_s() - t; if (!ok) { fprintf(stderr, "\n\nError! Please check your input! (%s)\n", err.c_str()); return 1; } }

	// FindGnuPlotPath (main.cpp:169-191): -dp needs a gnuplot binary -- the one -gp names, else the first one on PATH; without one the
	// reference plots nothing either
	std::string gnuplot = gnuplot_arg ? gnuplot_arg : "";
	if (dotplot && gnuplot.empty()) {
		const char *path = getenv("PATH");
		for (std::string p = path ? path : ""; !p.empty();) {
			const size_t k = p.find(':'); const std::string dir = p.substr(0, k);
			FILE *fp = fopen((dir + "/gnuplot").c_str(), "r");
			if (fp) { fclose(fp); gnuplot = dir + "/gnuplot"; break; }
			if (k == std::string::npos) break;
			p = p.substr(k + 1);
		}
	}
	gsa_index_view view; idx.fill_view(&view);
	view.ref = (const char *)idx.pac.data();      // (GSA_CREATE_REF_PAC below)
	if (gpus.empty()) gpus.push_back(0);
	double t_unpack = 0; bool unpack_ok = false; std::string unpack_err;
	std::thread unpacker([&] { const double t = now_s(); unpack_ok = gsah_unpack_ref(idx, unpack_err, true); t_unpack = now_s() - t; });      // (RestoreReferenceInfo for the emitters, beside gsa_create)
	struct Joiner2 { std::thread &t; ~Joiner2() { if (t.joinable()) t.join(); } } unpack_join{ unpacker };
	if (getenv("GSA_BIND")) (void)gsa_bind_host_thread(gpus[0]);      // (opt-in: small contigs gain from a near-socket thread, chromosome-sized ones lose 5 %)
	// one context per GPU owns that device's copy of the index; the others borrow it (gsa_clone).  The number of contexts wanted depends on
	// the number of query sequences: the owners are created first (GPU work, beside the FASTA loader), the clones once the count is known
	std::vector<gsa_ctx *> ctxs;
	{ const double t = now_s(); reserver.join(); t_reserve_wait = now_s() - t; }
	{
		const double t = now_s();
		for (size_t g = 0; g < gpus.size(); g++) {
			gsa_ctx *owner = NULL;
			// (the HOST PROGRAM honours a few environment variables and hands them to the library as options -- the library itself reads none:
			//  GSA_FORCE_WIDE=1: the >= 2^32-row device layout on any index; GSA_SPLIT_MIN / GSA_BUNDLE_CONTIG / GSA_BUNDLE_CAP: gsa_align_many's policy)
			const char *fw = getenv("GSA_FORCE_WIDE");
			// (the first GPU gets the index from the host -- upload + table builds; every further GPU gets a device-to-device copy of the finished tables:
			//  gsa_clone_to_device, no second pass over PCIe, nothing rebuilt)
			const int rc_c = g == 0 ? gsa_create_opts(gpus[g], &view, &prm, ((fw && *fw && *fw != '0') ? GSA_CREATE_WIDE : 0u) | GSA_CREATE_REF_PAC, &owner) : gsa_clone_to_device(ctxs[0], gpus[g], &owner);
			if (rc_c != GSA_OK) { fprintf(stderr, "GPU initialisation failed (device %d): %s\n", gpus[g], gsa_last_error(NULL)); return 2; }
			for (const char *nm : { "split_min", "bundle_contig", "bundle_cap" }) {
				std::string ev = std::string("GSA_") + nm; for (char &ch : ev) ch = (char)toupper((unsigned char)ch);
				if (const char *v = getenv(ev.c_str())) (void)gsa_set_option(owner, nm, atoll(v));
			}
			ctxs.push_back(owner);
		}
		t_create = now_s() - t;
	}
	{ const double t = now_s(); unpacker.join(); t_unpack_wait = now_s() - t; }
	if (!unpack_ok) { fprintf(stderr, "\n\nError! Please check your input! (%s)\n", unpack_err.c_str()); return 1; }
	idx.pac.resize(0);
	q_loader.join();
	if (!q_ok) { fprintf(stderr, "Please check the query file: %s\n", query_fa); return 0; }
	fprintf(stderr, "\tLoad the query sequences (%d %s)\n", (int)qs.size(), qs.size() > 1 ? "chromosomes" : "chromosome");
	fprintf(stderr, "\tLoad the reference sequences (%d %s)\n", (int)idx.chr_len.size(), idx.chr_len.size() > 1 ? "chromosomes" : "chromosome");
	{
		// (at least one context per listed GPU: with fewer query sequences than GPUs gsa_align_many seeds a long sequence on several of them)
		const size_t want_ctx = std::min(std::max(qs.size(), gpus.size()), gpus.size() * (size_t)n_ctx_per_gpu);
		const double t = now_s();
		for (int k = 1; k < n_ctx_per_gpu; k++)
			for (size_t g = 0; g < gpus.size() && ctxs.size() < want_ctx; g++) {
				gsa_ctx *cl = NULL;
				if (gsa_clone(ctxs[g], &cl) != GSA_OK) { fprintf(stderr, "GPU initialisation failed (device %d): %s\n", gpus[g], gsa_last_error(NULL)); return 2; }
				ctxs.push_back(cl);
			}
		t_create += now_s() - t;
	}

	const std::string maf = std::string(out_prefix) + ".maf", aln = std::string(out_prefix) + ".aln", vcfn = std::string(out_prefix) + ".vcf";
	Emitter em; em.idx = &idx; em.allow_dup = allow_dup;
	long long n_aln = 0, tot_len = 0, tot_match = 0, n_dup = 0;
	fprintf(stderr, "Step2. Sequence analysis for all query chromosomes\n");
	std::vector<const char *> qptr(qs.size()); std::vector<int32_t> qlen(qs.size());
	for (size_t ci = 0; ci < qs.size(); ci++) { qptr[ci] = qs[ci].seq.data(); qlen[ci] = (int32_t)qs[ci].seq.size(); }
	// the sequences are page-locked where the loader put them (10 ms per GB): their uploads are DMA transfers instead of staged copies
	double t_pin = 0;
	{ const double t = now_s(); for (size_t ci = 0; ci < qs.size(); ci++) if (qs[ci].seq.size() >= ((size_t)1 << 20)) (void)gsa_host_register(&qs[ci].seq[0], qs[ci].seq.size()); t_pin = now_s() - t; }
	// finished contigs: copied by the GPU worker that produced them (on_result), taken in contig order by the formatter thread
	struct Sink {
		std::mutex mu; std::condition_variable cv; std::vector<ContigResult> res; std::vector<char> ready; bool abort = false; double copy_s = 0;
	} sink;
	sink.res.resize(qs.size()); sink.ready.assign(qs.size(), 0);
	// the MAF file: created by the first contig that has alignments ("w" for contig 0, "a" afterwards: tools.cpp:158-163 -- a run whose
	// first contig aligns nowhere appends to whatever the file held, as the reference does)
	int maf_fd = -1; std::unique_ptr<OrderedWriter> maf_w;
	bool out_failed = false;          // an output file that could not be opened or written: reported, exit status 1 (a full disk must not look like success)
	size_t written = 0;
	auto write_contig = [&](size_t ci, ContigResult &cr) {
		fprintf(stderr, "\tProcess query chromsomoe: %s...\n", qs[ci].name.c_str());
		if (cr.blocks.empty()) return;
		long long len = 0, score = 0;
		for (size_t b = 0; b < cr.blocks.size(); b++) { len += cr.blocks[b].aln_len; score += cr.blocks[b].score; if (cr.blocks[b].bdup) n_dup++; }
		n_aln += (long long)cr.blocks.size(); tot_len += len; tot_match += score;
		fprintf(stderr, "\t\tProduce %d local alignments (length = %lld), ANI=%.2f%%\n", (int)cr.blocks.size(), len, 100 * (1.0 * score / len));
		if (fmt == 1) {
			const double t = now_s();
			if (maf_fd < 0 || ci == 0) {
				if (maf_w) { if (!maf_w->close()) out_failed = true; maf_w.reset(); }
				if (maf_fd >= 0) close(maf_fd);
				maf_fd = open(maf.c_str(), O_WRONLY | O_CREAT | (ci == 0 ? O_TRUNC : O_APPEND), 0644);
				if (maf_fd >= 0) maf_w.reset(new OrderedWriter(maf_fd));
				else if (!out_failed) { out_failed = true; fprintf(stderr, "Error! cannot open [%s] for writing: %s\n", maf.c_str(), strerror(errno)); }
			}
			if (maf_w) em.maf_text(ci == 0, qs[ci], cr, [&](OutBuf &&o) { maf_w->push(std::move(o)); }, [&](size_t c) { return maf_w->take(c); });
			t_maf_fmt += now_s() - t;
		}
		if (fmt == 2) {
			FILE *fp = fopen(aln.c_str(), ci == 0 ? "w" : "a");
			if (fp) { em.aln(fp, qs[ci], cr); if (ferror(fp) | fclose(fp)) out_failed = true; }
			else if (!out_failed) { out_failed = true; fprintf(stderr, "Error! cannot open [%s] for writing: %s\n", aln.c_str(), strerror(errno)); }
		}
		if (vcf) { const double t = now_s(); em.variants((int)ci, qs[ci], cr); t_var += now_s() - t; }
		if (dotplot && !gnuplot.empty()) {                               // GSAlign.cpp:546: only when gnuplot was found (main.cpp:324)
			const std::string gp = std::string(out_prefix) + ".gp";
			std::vector<std::string> data_files;
			if (em.dotplot(gp, out_prefix, qs[ci], cr, &data_files)) {
				fprintf(stderr, "\t\tGenerate dotplot for query sequence (%s-%s.ps)\n", out_prefix, qs[ci].name.c_str());
				if (system((gnuplot + " " + gp).c_str()) != 0) fprintf(stderr, "\t\tgnuplot failed\n");
				// (the reference shells out to `rm <prefix>.<contig name>*` with the raw FASTA header name, DotPloting.cpp:70: the
				//  files it means are exactly the data files written above)
				for (const std::string &fn : data_files) (void)remove(fn.c_str());
			}
		}
	};
	std::thread formatter([&] {
		for (size_t k = 0; k < qs.size(); k++) {
			{ std::unique_lock<std::mutex> lk(sink.mu); sink.cv.wait(lk, [&] { return sink.abort || sink.ready[k]; }); if (!sink.ready[k]) return; }
			write_contig(k, sink.res[k]);
			ContigResult().blocks.swap(sink.res[k].blocks); sink.res[k].recs.reset();
			sink.res[k].aln1.reset(); sink.res[k].aln2.reset();
			written = k + 1;
		}
	});
	auto on_result = [](void *user, int32_t ci, const gsa_result *res) -> int {
		Sink &sk = *(Sink *)user;
		const double t = now_s();
		sk.res[(size_t)ci].assign(*res);
		const double dt = now_s() - t;
		{ std::lock_guard<std::mutex> lk(sk.mu); sk.ready[(size_t)ci] = 1; sk.copy_s += dt; }
		sk.cv.notify_all();
		return 0;
	};
	const double ta = now_s(); t_at_align = ta - T0;
	const int rc_many = gsa_align_many(ctxs.data(), (int32_t)ctxs.size(), qptr.data(), qlen.data(), (int32_t)qs.size(), 0, on_result, &sink);
	t_align = now_s() - ta;
	// where the contexts' host threads spent that time (gsa_get_wall_sums: [0] query set-up / wait for the upload, [s] stage s) and what growing buffers cost them
	if (timing && getenv("GSA_DUMP_BUFFERS")) for (gsa_ctx *c : ctxs) (void)gsa_debug_buffers(c, 24);
	double wall_sum[10] = { 0 }; double alloc_ms = 0; long long alloc_n = 0, alloc_bytes = 0;
	for (gsa_ctx *c : ctxs) {
		double w[10]; int64_t wn = 0; if (gsa_get_wall_sums(c, w, &wn) == GSA_OK) for (int k = 0; k < 9; k++) wall_sum[k] += w[k];
		double am = 0; int64_t an = 0, ab = 0; if (gsa_get_alloc_stats(c, &am, &an, &ab) == GSA_OK) { alloc_ms += am; alloc_n += an; alloc_bytes += ab; }
	}
	{ std::lock_guard<std::mutex> lk(sink.mu); if (rc_many != GSA_OK) sink.abort = true; }
	sink.cv.notify_all();
	const double td = now_s();
	// (the contexts are done with -- every result was copied by on_result: their teardown, 0.2 s of hipFree at human scale, runs beside the output's tail)
	double t_destroy = 0;
	std::thread destroyer([&] { if (rc_many != GSA_OK) return; const double t = now_s(); for (size_t k = ctxs.size(); k-- > 0;) gsa_destroy(ctxs[k]); t_destroy = now_s() - t; });      // clones before the owners of their index (after an error the contexts stay: their messages are printed below)
	formatter.join();
	double maf_write_s = 0; unsigned long long maf_bytes = 0;
	// (the MAF writer still has its queue to write -- ~1 s at human scale: the VCF is sorted, formatted and written beside it, the MAF file is closed behind)
	auto close_maf = [&] {
		if (maf_w) { if (!maf_w->close()) out_failed = true; maf_write_s = maf_w->write_seconds(); maf_bytes = maf_w->bytes(); maf_w.reset(); }
		if (maf_fd >= 0) { if (close(maf_fd) != 0) out_failed = true; maf_fd = -1; }
	};
	t_copy = sink.copy_s;
	if (rc_many != GSA_OK) {
		close_maf(); destroyer.join();
		for (gsa_ctx *c : ctxs) if (*gsa_last_error(c)) fprintf(stderr, "GPU error: %s\n", gsa_last_error(c));
		fprintf(stderr, "\t%d of %d query sequences were written before the error\n", (int)written, (int)qs.size());
		return 2;
	}
	if (n_aln > 0) fprintf(stderr, "\tAlignment#=%d (total alignment length=%lld) ANI=%.2f%%, unique alignment#=%d\n", (int)n_aln, tot_len, 100 * (1.0 * tot_match / tot_len), (int)(n_aln - n_dup));
	fprintf(stderr, "\tIt took %lld seconds for genome sequence alignment.\n", (long long)(time(NULL) - t0));
	unsigned long long vcf_bytes = 0; double vcf_write_s = 0;
	if (vcf) {
		const double t = now_s();
		fprintf(stderr, "\nGSAlign identifies %d SNVs, %d insertions, and %d deletions [%s].\n\n", em.n_snv, em.n_ins, em.n_del, vcfn.c_str());
		const int fd = open(vcfn.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
		if (fd >= 0) {
			OrderedWriter w(fd);
			em.vcf_text(index_prefix != NULL ? index_prefix : ref_fa, [&](OutBuf &&o) { w.push(std::move(o)); });
			if (!w.close()) out_failed = true;
			vcf_bytes = w.bytes(); vcf_write_s = w.write_seconds();
			if (close(fd) != 0) out_failed = true;
		} else { out_failed = true; fprintf(stderr, "Error! cannot open [%s] for writing: %s\n", vcfn.c_str(), strerror(errno)); }
		t_vcf = now_s() - t;
	}
	close_maf();
	t_drain = now_s() - td;      // (from the last contig's alignment to both files closed; the VCF's time lies inside it now)
	destroyer.join();
	if (timing) {
		long long qbp = 0; for (const QueryContig &q : qs) qbp += (long long)q.seq.size();
		const double total = now_s() - T0;
		fprintf(stderr, "GSA_TIMING {\"total_s\": %.3f, \"index_build_s\": %.3f, \"index_load_s\": %.3f, \"gsa_create_s\": %.3f, \"query_load_s\": %.3f, \"query_pin_s\": %.3f, \"align_many_s\": %.3f, "
		        "\"result_copy_s_sum\": %.3f, \"maf_format_s\": %.3f, \"variants_s\": %.3f, \"output_drain_after_align_s\": %.3f, \"maf_write_s\": %.3f, \"maf_bytes\": %llu, "
		        "\"vcf_s\": %.3f, \"vcf_write_s\": %.3f, \"vcf_bytes\": %llu, \"destroy_s\": %.3f, \"host_threads\": %d, \"contexts\": %d, \"query_bp\": %lld, \"contigs\": %d, \"gbp_per_s_excl_index_build\": %.4f, "
		        "\"ref_unpack_s\": %.3f, \"reserve_s\": %.3f, \"reserve_wait_s\": %.3f, \"unpack_wait_s\": %.3f, \"align_starts_at_s\": %.3f, \"ctx_wall_ms_sum\": [%.1f, %.1f, %.1f, %.1f, %.1f, %.1f, %.1f, %.1f, %.1f], \"alloc_ms_sum\": %.1f, \"alloc_n\": %lld, \"alloc_gb\": %.2f}\n",
		        total, t_build, t_index, t_create, t_query, t_pin, t_align, t_copy, t_maf_fmt, t_var, t_drain, maf_write_s, maf_bytes, t_vcf, vcf_write_s, vcf_bytes, t_destroy,
		        HostPool::global().threads(), (int)ctxs.size(), qbp, (int)qs.size(), (double)qbp / (total - t_build) / 1e9,
		        t_unpack, t_reserve, t_reserve_wait, t_unpack_wait, t_at_align, wall_sum[0], wall_sum[1], wall_sum[2], wall_sum[3], wall_sum[4], wall_sum[5], wall_sum[6], wall_sum[7], wall_sum[8], alloc_ms, alloc_n, (double)alloc_bytes / 1e9);
	}
	// (everything is on disk and the GPU is released: the process ends here -- unwinding 20 GB of host buffers and the HIP runtime's own
	//  teardown cost a second or two of wall time at human scale and produce nothing)
	if (out_failed) fprintf(stderr, "Error! an output file could not be written completely (disk full?)\n");
	fflush(stdout); fflush(stderr);
	_exit(out_failed ? 1 : 0);
}
