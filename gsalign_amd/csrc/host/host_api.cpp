// gsalign_amd/csrc/host/host_api.cpp -- C entry points of libgsa_host.so, so that
// the CPU test-suite can drive the host-side components (index builder, loader,
// emitters) without a GPU.  The CLI uses the C++ interface directly.
#include <atomic>
#include <algorithm>
#include <cstring>
#include "gsa_host.h"
#include "exact_sort.h"

extern "C" {

// LoadQueryFile through the C API (tests): the contigs stay in a static list until the next call
static std::vector<QueryContig> g_query;
int gsah_c_load_query(const char *path, char *err)
{
	g_query.clear(); std::string e;
	if (gsah_load_query(path, g_query, e)) return (int)g_query.size();
	if (err) { strncpy(err, e.c_str(), 255); err[255] = 0; }
	g_query.clear();
	return -1;
}
const char *gsah_c_query_name(int i) { return g_query[(size_t)i].name.c_str(); }
long long gsah_c_query_len(int i) { return (long long)g_query[(size_t)i].seq.size(); }
const char *gsah_c_query_seq(int i) { return g_query[(size_t)i].seq.data(); }

// exact_sort (exact_sort.h) against std::sort itself on n keys {a, b, original index} compared on (a, b) only -- the VCF sort's shape.
// pattern 0: random with `distinct` values per field (ties), 1: ascending, 2: descending, 3: all equal, 4: organ pipe, 5: a few long runs, 6: the
// median-of-three killer (quicksort's depth budget runs out: the heapsort fallback).  Returns 0 when every element sits where std::sort put it.
int gsah_c_exact_sort_check(long long n, int distinct, unsigned seed, int pattern, long long grain)
{
	struct Key { int32_t a, b; uint32_t idx, pad; };
	struct ByAB { bool operator()(const Key &x, const Key &y) const { return x.a == y.a ? x.b < y.b : x.a < y.a; } };
	std::vector<Key> v((size_t)n);
	unsigned long long st = seed * 2654435761ull + 88172645463325252ull;
	auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
	for (long long i = 0; i < n; i++) {
		Key k; k.idx = (uint32_t)i; k.pad = 0;
		switch (pattern) {
		case 1: k.a = 0; k.b = (int32_t)(i / 3); break;
		case 2: k.a = 0; k.b = (int32_t)((n - i) / 3); break;
		case 3: k.a = 7; k.b = 7; break;
		case 4: k.a = 0; k.b = (int32_t)(i < n / 2 ? i : n - i); break;
		case 5: k.a = (int32_t)(i * 5 / (n > 0 ? n : 1)); k.b = (int32_t)(rnd() % 3); break;
		default: k.a = (int32_t)(rnd() % (unsigned)(distinct < 1 ? 1 : distinct)) / 97; k.b = (int32_t)(rnd() % (unsigned)(distinct < 1 ? 1 : distinct)); break;
		}
		v[(size_t)i] = k;
	}
	if (pattern == 6 && n >= 4) {      // Musser's median-of-three killer on b
		const long long k2 = n / 2;
		for (long long i = 0; i < k2; i++) { v[(size_t)i].a = 0; v[(size_t)i].b = (i % 2 == 0) ? (int32_t)(i + 1) : (int32_t)(k2 + i + (k2 % 2 ? 0 : 1)); }
		for (long long i = k2; i < n; i++) { v[(size_t)i].a = 0; v[(size_t)i].b = (int32_t)((i - k2 + 1) * 2); }
	}
	std::vector<Key> w(v);
	std::sort(v.begin(), v.end(), ByAB());
	exact_sort(w.data(), w.data() + w.size(), ByAB(), grain > 0 ? (size_t)grain : (size_t)1 << 16);
	for (size_t i = 0; i < v.size(); i++) if (v[i].idx != w[i].idx) return 1;
	return 0;
}

// returns 0 on success; err (>= 256 bytes) receives the message otherwise
// HostPool::run back to back with short jobs of changing size (the shape maf_block produces): every index of every run exactly once.
// Returns 0, or the 1-based run in which an index ran twice or not at all.  (Round-5 advisor finding: a worker that woke late could carry an
// index of the previous job into the next one.)
int gsah_c_pool_stress(int threads, int runs, unsigned seed)
{
	HostPool pool(threads);
	std::vector<std::atomic<int>> hit(4096);
	unsigned x = seed * 2654435761u + 1u;
	for (int r = 0; r < runs; r++) {
		x = x * 1664525u + 1013904223u;
		const size_t n = 2 + (x >> 20) % 4094;
		for (size_t i = 0; i < n; i++) hit[i].store(0, std::memory_order_relaxed);
		pool.run(n, [&](size_t i) { hit[i].fetch_add(1, std::memory_order_relaxed); });
		for (size_t i = 0; i < n; i++) if (hit[i].load(std::memory_order_relaxed) != 1) return r + 1;
	}
	return 0;
}

int gsah_c_build_index(const char *fasta, const char *prefix, char *err)
{
	std::string e;
	if (gsah_build_index(fasta, prefix, e)) return 0;
	if (err) { strncpy(err, e.c_str(), 255); err[255] = 0; }
	return -1;
}

// Emit MAF + VCF for a whole query FASTA given, per contig, a finished gsa_result.
// get_result(user, contig_index, seq, len, &result) is called once per contig, in order.
typedef int (*gsah_result_cb)(void *user, int contig, const char *seq, int len, gsa_result *out);

int gsah_c_emit_fmt(const char *index_prefix, const char *query_fa, const char *maf_path, const char *vcf_path, const char *reference_label,
                    int allow_dup, int fmt, gsah_result_cb cb, void *user, char *err);

int gsah_c_emit(const char *index_prefix, const char *query_fa, const char *maf_path, const char *vcf_path, const char *reference_label,
                int allow_dup, gsah_result_cb cb, void *user, char *err)
{
	return gsah_c_emit_fmt(index_prefix, query_fa, maf_path, vcf_path, reference_label, allow_dup, 1, cb, user, err);
}

// fmt 1: MAF (OutputMAF), fmt 2: ALN (OutputAlignment) -- the -fmt flag of the CLI (main.cpp:284)
int gsah_c_emit_fmt(const char *index_prefix, const char *query_fa, const char *maf_path, const char *vcf_path, const char *reference_label,
                    int allow_dup, int fmt, gsah_result_cb cb, void *user, char *err)
{
	std::string e; HostIndex idx; std::vector<QueryContig> qs;
	if (!gsah_load_index(index_prefix, idx, e) || !gsah_load_query(query_fa, qs, e)) { if (err) { strncpy(err, e.c_str(), 255); err[255] = 0; } return -1; }
	Emitter em; em.idx = &idx; em.allow_dup = allow_dup != 0;
	for (size_t ci = 0; ci < qs.size(); ci++) {
		gsa_result res; memset(&res, 0, sizeof(res));
		if (cb(user, (int)ci, qs[ci].seq.data(), (int)qs[ci].seq.size(), &res) != 0) { if (err) strcpy(err, "result callback failed"); return -2; }
		if (res.n_blocks == 0) continue;                         // GSAlign.cpp:541
		ContigResult cr; cr.assign(res);
		FILE *fp = fopen(maf_path, ci == 0 ? "w" : "a");         // tools.cpp:158-163
		if (!fp) { if (err) strcpy(err, "cannot open MAF output"); return -3; }
		if (fmt == 2) em.aln(fp, qs[ci], cr); else em.maf(fp, ci == 0, qs[ci], cr);
		fclose(fp);
		em.variants((int)ci, qs[ci], cr);
	}
	FILE *fp = fopen(vcf_path, "w");
	if (!fp) { if (err) strcpy(err, "cannot open VCF output"); return -3; }
	em.vcf(fp, reference_label); fclose(fp);
	return 0;
}

// OutputDotplot for one contig: script + data files (no gnuplot run).  Returns 1 if something was written.
// The reference plots AFTER OutputMAF (GSAlign.cpp:543-546), which shortens the last record of a block that runs over the end of
// its reference sequence (tools.cpp:149-220), and the plot shows the shortened block: the MAF emitter runs first here too.
int gsah_c_dotplot(const char *index_prefix, const char *query_fa, int contig, const char *gp_path, const char *out_prefix, gsah_result_cb cb, void *user, char *err)
{
	std::string e; HostIndex idx; std::vector<QueryContig> qs;
	if (!gsah_load_index(index_prefix, idx, e) || !gsah_load_query(query_fa, qs, e) || contig < 0 || contig >= (int)qs.size()) { if (err) { strncpy(err, e.c_str(), 255); err[255] = 0; } return -1; }
	gsa_result res; memset(&res, 0, sizeof(res));
	if (cb(user, contig, qs[(size_t)contig].seq.data(), (int)qs[(size_t)contig].seq.size(), &res) != 0) return -2;
	ContigResult cr; cr.assign(res);
	Emitter em; em.idx = &idx;
	if (!cr.blocks.empty()) { FILE *nul = fopen("/dev/null", "w"); if (nul) { em.maf(nul, false, qs[(size_t)contig], cr); fclose(nul); } }
	return em.dotplot(gp_path, out_prefix, qs[(size_t)contig], cr) ? 1 : 0;
}

} // extern "C"
