// tests/host_tsan/host_tsan_main.cpp -- the host-side components (gsalign_amd/csrc/host: thread pool, loaders, index builder, MAF / VCF emitters, ordered writer) driven
// the way GSAlign_hip's main() drives them -- several threads at once -- in a build with -fsanitize=thread.  Test infrastructure (tests/test_host_tsan.py compiles and runs it);
// needs no GPU: the "results" are made up here (a query that is its reference with a substitution every ~100 bases: seed, one-base gap, seed, ...), the point is the
// thread interplay, not the bytes (the bytes are the business of tests/test_host_components.py and the CLI goldens).
//   1. HostPool::run back to back (the round-5 advisor's race) and the GLOBAL pool entered from three threads at once (callers serialise on its gate)
//   2. exact_sort on the pool
//   3. index build -> gsah_load_index_files, then RefSequence unpacked on one thread while another loads the query FASTA (main.cpp's start-up)
//   4. two "GPU worker" threads handing finished contigs over (ContigResult::assign: copies on the pool) while the formatter thread writes MAF through an OrderedWriter
//      and collects variants; then the VCF
// usage: host_tsan <work dir> [chromosome length] [race]
// Exit status 0 and "HOST_TSAN_OK"; ThreadSanitizer reports go to stderr and make the exit status 66.
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "gsa_host.h"
#include "par.h"

extern "C" int gsah_c_pool_stress(int threads, int runs, unsigned seed);
extern "C" int gsah_c_exact_sort_check(long long n, int distinct, unsigned seed, int pattern, long long grain);

static unsigned long long rs = 88172645463325252ull;
static unsigned rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (unsigned)(rs >> 11); }

struct Made { std::vector<gsa_block> blocks; std::vector<gsa_rec> recs; std::string a1, a2; };

// contig = chromosome `chr` of the index with a substitution every 60 - 140 bases; one forward block per `blk` bases
static void make_result(const HostIndex &idx, int chr, std::string &qseq, Made &m, int blk)
{
	const int64_t r0 = idx.chr_fwd[(size_t)chr]; const int len = idx.chr_len[(size_t)chr];
	qseq.assign(idx.ref.data() + r0, (size_t)len);
	for (int b0 = 0; b0 < len; b0 += blk) {
		const int b1 = std::min(len, b0 + blk);
		gsa_block B; memset(&B, 0, sizeof(B));
		B.frag_off = (int64_t)m.recs.size(); B.bdir = 1; B.chr = chr; B.gpos = b0 + 1; B.bdup = 0;
		int p = b0, score = 0;
		while (p < b1) {
			int s = p + 60 + (int)(rnd() % 80); if (s >= b1 - 1) s = b1;
			gsa_rec r; r.seed.qpos = p; r.seed.len = s - p; r.seed.rpos = r0 + p; m.recs.push_back(r); score += s - p;
			if (s < b1) {
				const char rc = qseq[(size_t)s]; const char qc = rc == 'A' ? 'C' : 'A'; qseq[(size_t)s] = qc;
				gsa_rec g; g.gap.nqlen = -1 - 1; g.gap.rlen = 1; g.gap.aln_len = 1; g.gap.aln_off = (uint32_t)m.a1.size(); m.recs.push_back(g);
				m.a1.push_back(rc); m.a2.push_back(qc);
				p = s + 1;
			} else p = s;
		}
		// (a block ends with a seed: the loop's last record is one -- when the last step was a gap, p == b1 only after a seed of length >= 1 follows; make sure)
		if (!gsa_rec_is_seed(&m.recs.back())) { m.recs.pop_back(); m.a1.pop_back(); m.a2.pop_back(); }
		B.n_frag = (int32_t)((int64_t)m.recs.size() - B.frag_off); B.score = score; B.aln_len = b1 - b0;
		m.blocks.push_back(B);
	}
}

int main(int argc, char **argv)
{
	const std::string dir = argc > 1 ? argv[1] : "/tmp";
	const int chr_len = argc > 2 ? atoi(argv[2]) : 600000;
	if (argc > 3 && !strcmp(argv[3], "race")) {      // positive control: an unsynchronised counter -- the ThreadSanitizer build must report it (exit status 66)
		static long long racy = 0;
		std::thread a([] { for (int i = 0; i < 100000; i++) racy++; }), b([] { for (int i = 0; i < 100000; i++) racy++; });
		a.join(); b.join(); printf("racy %lld\n", racy); return 0;
	}
	setenv("GSA_HOST_PAR_MIN", "4096", 1);      // the parallel forms at this size
	setenv("GSA_HOST_THREADS", "6", 0);

	// 1
	if (int r = gsah_c_pool_stress(6, 1500, 7)) { fprintf(stderr, "pool stress: run %d\n", r); return 1; }
	{
		std::vector<std::thread> th; std::atomic<long long> sum{0};
		for (int t = 0; t < 3; t++) th.emplace_back([&, t] {
			for (int k = 0; k < 200; k++) { const size_t n = 1000 + 37 * (size_t)((k + t) % 50); std::atomic<long long> s{0};
				par_ranges(n, 16, [&](size_t b, size_t e) { long long x = 0; for (size_t i = b; i < e; i++) x += (long long)i; s += x; });
				if (s.load() != (long long)n * (long long)(n - 1) / 2) { fprintf(stderr, "par_ranges: wrong sum\n"); exit(1); }
				sum += s.load(); } });
		for (auto &x : th) x.join();
	}
	// 2
	for (int pat = 0; pat <= 6; pat++) if (gsah_c_exact_sort_check(120000, 50, 3u + (unsigned)pat, pat, 2048)) { fprintf(stderr, "exact_sort pattern %d\n", pat); return 1; }

	// 3
	const std::string fa = dir + "/tsan_ref.fa", px = dir + "/tsan_ref", qfa = dir + "/tsan_qry.fa";
	{
		FILE *f = fopen(fa.c_str(), "w"); if (!f) { perror("ref.fa"); return 1; }
		for (int c = 0; c < 3; c++) { fprintf(f, ">chr%d\n", c + 1); for (int i = 0; i < chr_len + 1000 * c; i++) { fputc("ACGT"[rnd() & 3], f); if (i % 70 == 69) fputc('\n', f); } fputc('\n', f); }
		fclose(f);
	}
	std::string err;
	if (!gsah_build_index(fa, px, err)) { fprintf(stderr, "build_index: %s\n", err.c_str()); return 1; }
	HostIndex idx;
	if (!gsah_load_index_files(px, idx, err)) { fprintf(stderr, "load_index_files: %s\n", err.c_str()); return 1; }
	// (the query file must exist before it is loaded beside the unpacking: written from a first, serial unpack of a second index object)
	std::vector<std::string> qseq(3); std::vector<Made> made(3);
	{
		HostIndex i2; if (!gsah_load_index(px, i2, err)) { fprintf(stderr, "load_index: %s\n", err.c_str()); return 1; }
		for (int c = 0; c < 3; c++) make_result(i2, c, qseq[(size_t)c], made[(size_t)c], 150000);
		FILE *f = fopen(qfa.c_str(), "w"); if (!f) { perror("qry.fa"); return 1; }
		for (int c = 0; c < 3; c++) { fprintf(f, ">q%d some description\n", c + 1); for (size_t i = 0; i < qseq[(size_t)c].size(); i += 60) { fwrite(qseq[(size_t)c].data() + i, 1, std::min<size_t>(60, qseq[(size_t)c].size() - i), f); fputc('\n', f); } }
		fclose(f);
	}
	std::vector<QueryContig> qs; std::string e1, e2; bool ok1 = false, ok2 = false;
	{
		std::thread unpacker([&] { ok1 = gsah_unpack_ref(idx, e1, true); });
		std::thread loader([&] { ok2 = gsah_load_query(qfa, qs, e2); });
		unpacker.join(); loader.join();
	}
	if (!ok1 || !ok2 || qs.size() != 3) { fprintf(stderr, "unpack / load_query: %s %s\n", e1.c_str(), e2.c_str()); return 1; }
	for (int c = 0; c < 3; c++) if (qs[(size_t)c].seq != qseq[(size_t)c]) { fprintf(stderr, "query %d read back differently\n", c); return 1; }

	// 4
	struct Sink { std::mutex mu; std::condition_variable cv; std::vector<ContigResult> res; std::vector<char> ready; } sink;
	sink.res.resize(3); sink.ready.assign(3, 0);
	Emitter em; em.idx = &idx;
	const std::string maf = dir + "/tsan_out.maf", vcf = dir + "/tsan_out.vcf";
	const int fd = open(maf.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644); if (fd < 0) { perror("maf"); return 1; }
	unsigned long long maf_bytes = 0;
	{
		OrderedWriter w(fd, (size_t)1 << 20);      // (a small budget: push() has to wait for the writer)
		std::thread formatter([&] {
			for (size_t k = 0; k < 3; k++) {
				{ std::unique_lock<std::mutex> lk(sink.mu); sink.cv.wait(lk, [&] { return sink.ready[k] != 0; }); }
				em.maf_text(k == 0, qs[k], sink.res[k], [&](OutBuf &&o) { w.push(std::move(o)); }, [&](size_t c) { return w.take(c); });
				em.variants((int)k, qs[k], sink.res[k]);
				ContigResult().blocks.swap(sink.res[k].blocks); sink.res[k].recs.reset(); sink.res[k].aln1.reset(); sink.res[k].aln2.reset();
			}
		});
		auto deliver = [&](size_t k) {
			gsa_result r; memset(&r, 0, sizeof(r)); const Made &m = made[k];
			r.n_blocks = (int32_t)m.blocks.size(); r.n_frags = (int64_t)m.recs.size(); r.n_aln = (int64_t)m.a1.size();
			r.blocks = m.blocks.data(); r.recs = m.recs.data(); r.aln1 = m.a1.data(); r.aln2 = m.a2.data();
			sink.res[k].assign(r);
			{ std::lock_guard<std::mutex> lk(sink.mu); sink.ready[k] = 1; }
			sink.cv.notify_all();
		};
		std::thread w0([&] { deliver(1); deliver(0); }), w1([&] { deliver(2); });      // out of order, as contexts finish
		w0.join(); w1.join(); formatter.join();
		if (!w.close()) { fprintf(stderr, "MAF writer failed\n"); return 1; }
		maf_bytes = w.bytes();
	}
	close(fd);
	{
		const int vfd = open(vcf.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644); if (vfd < 0) { perror("vcf"); return 1; }
		OrderedWriter w(vfd);
		em.vcf_text("tsan_ref", [&](OutBuf &&o) { w.push(std::move(o)); });
		if (!w.close()) return 1;
		close(vfd);
	}
	size_t n_gaps = 0; for (const Made &m : made) n_gaps += m.a1.size();
	if (em.n_snv != (int)n_gaps || maf_bytes < (unsigned long long)(2 * 3 * chr_len)) { fprintf(stderr, "SNVs %d (made %zu), MAF bytes %llu\n", em.n_snv, n_gaps, maf_bytes); return 1; }
	printf("HOST_TSAN_OK %d SNVs, %llu MAF bytes\n", em.n_snv, maf_bytes);
	return 0;
}
