// gsalign_amd/csrc/host/host_api.cpp -- C entry points of libgsa_host.so, so that
// the CPU test-suite can drive the host-side components (index builder, loader,
// emitters) without a GPU.  The CLI uses the C++ interface directly.
#include <cstring>
#include "gsa_host.h"

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

// returns 0 on success; err (>= 256 bytes) receives the message otherwise
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
