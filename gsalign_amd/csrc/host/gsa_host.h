// gsalign_amd/csrc/host/gsa_host.h -- CPU-side parts of the aligner that
// north_star keeps on the host: index build/load, query FASTA loading, MAF / ALN
// / VCF emission.  Plain C++ (g++), no HIP.  Built twice: into the CLI
// (GSAlign_hip) and into libgsa_host.so, whose C entry points let the CPU tests
// drive the emitters and the index builder without a GPU.
#ifndef GSA_HOST_H
#define GSA_HOST_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <functional>
#include <string>
#include <vector>
#include "gsa_hip.h"

// A buffer that is NOT zero-filled when it is sized: the index arrays of a 3.08 Gbp reference are 11 GB that the loader overwrites anyway
// (std::vector / std::string would first clear them on one thread, ~1 s; the loader's threads then fault the pages in themselves).
template <class T> struct RawBuf {
	T *p = nullptr; size_t n = 0;
	RawBuf() {}
	RawBuf(const RawBuf &) = delete; RawBuf &operator=(const RawBuf &) = delete;
	~RawBuf() { free(p); }
	void resize(size_t m) { free(p); p = m ? (T *)malloc(m * sizeof(T)) : nullptr; n = p ? m : 0; }
	T *data() { return p; } const T *data() const { return p; }
	size_t size() const { return n; }
	T &operator[](size_t i) { return p[i]; } const T &operator[](size_t i) const { return p[i]; }
};

struct HostIndex {
	uint64_t primary = 0, L2[5] = {0, 0, 0, 0, 0};
	RawBuf<uint32_t> bwt;
	RawBuf<uint64_t> sa;
	int64_t G = 0;
	std::vector<std::string> chr_name;
	std::vector<int32_t> chr_len;
	std::vector<int64_t> chr_fwd, chr_rev;        // FowardLocation / ReverseLocation (bwt_index.cpp:247-248)
	std::vector<int64_t> end_key; std::vector<int32_t> end_chr;   // ChrLocMap as sorted arrays
	RawBuf<char> ref;                             // RefSequence: 2G ASCII
	RawBuf<uint8_t> pac;                          // the .pac bytes (between gsah_load_index_files and gsah_unpack_ref)

	void fill_view(gsa_index_view *v) const;
	// GenCoordinateInfo (tools.cpp:120-140)
	void coordinate(int64_t rpos, int *bdir, int *chr, int *gpos) const;
};

struct QueryContig { std::string name, seq; };

// index files (reference src/bwt_index.cpp:25-264, src/GetData.cpp:8-24)
bool gsah_index_files_exist(const std::string &prefix);
bool gsah_load_index(const std::string &prefix, HostIndex &idx, std::string &err);
// the same in two steps: the files as they are (idx.pac = the raw .pac bytes, idx.ref still empty), then RestoreReferenceInfo's unpacking (bwt_index.cpp:229-264) --
// a host hands idx.pac to gsa_create_opts(GSA_CREATE_REF_PAC) after the first step and unpacks its own RefSequence while the device builds its tables
bool gsah_load_index_files(const std::string &prefix, HostIndex &idx, std::string &err);
bool gsah_unpack_ref(HostIndex &idx, std::string &err, bool keep_pac = false);
// bwa_idx_build (reference src/BWT_Index/bwtindex.c:77-149): byte-identical .bwt .sa .pac .ann .amb
bool gsah_build_index(const std::string &fasta, const std::string &prefix, std::string &err);
// LoadQueryFile / TrimChromosomeName / CheckQuerySeq (reference src/main.cpp:35-114)
bool gsah_load_query(const std::string &path, std::vector<QueryContig> &out, std::string &err);

// A plain array that is allocated WITHOUT being touched (std::vector / std::string write every byte once before the copy does, or fault their pages on one
// thread): a 250 Mb contig's records and gapped strings are ~150 MB, and a GPU worker thread is inside the result callback while they are copied.
template <class T> struct RawArr {
	T *p = nullptr; size_t n = 0;
	RawArr() {}
	RawArr(const RawArr &) = delete; RawArr &operator=(const RawArr &) = delete;
	RawArr(RawArr &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
	RawArr &operator=(RawArr &&o) noexcept { if (this != &o) { free(p); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
	~RawArr() { free(p); }
	void reset() { free(p); p = nullptr; n = 0; }
	void assign(const T *src, size_t count);            // emit.cpp: large arrays are copied by the pool's threads (first touch spread over them)
	T *data() { return p; } const T *data() const { return p; }
	size_t size() const { return n; }
	T &operator[](size_t i) { return p[i]; } const T &operator[](size_t i) const { return p[i]; }
};

// one finished contig, as delivered by gsa_align_contig
struct ContigResult {
	std::vector<gsa_block> blocks;
	RawArr<gsa_rec> recs;                  // the 16-byte records as they arrived; frag(i) is record i as a FragPair_t
	RawArr<char> aln1, aln2;
	void assign(const gsa_result &r);      // the copies only: what a GPU worker thread does inside the result callback
	gsa_frag frag(int64_t i) const { gsa_frag f; gsa_rec_expand(recs.data(), i, &f); return f; }
	void trim(int64_t i, int ext);         // iExtension (tools.cpp:192-202): record i loses its last `ext` bases
};

// Variant_t (structure.h:124-132); the two alleles are pieces of RefSequence / of the query sequence and are pointed at, not copied:
// both outlive the Emitter's list (the index and the query contigs are loaded once per run)
struct Variant { int pos, chr_idx, query_idx, type; const char *ref_p, *alt_p; uint32_t ref_n, alt_n; };

struct OutBuf;                             // par.h

struct Emitter {
	const HostIndex *idx = nullptr;
	bool allow_dup = true;                 // !-unique
	std::vector<Variant> vars;             // VarVec ...
	std::vector<std::vector<Variant> > var_chunks;   // ... kept as the lists the pool's threads filled, in the serial order (vars first)
	int n_snv = 0, n_ins = 0, n_del = 0;
	// OutputMAF (tools.cpp:149-220).  first = (QueryChrIdx == 0).  May shorten the last record of a block (iExtension).
	void maf(FILE *fp, bool first, const QueryContig &q, ContigResult &r) const;
	// the same bytes handed to `sink` buffer after buffer, in order; a block's two text lines are filled by the pool's threads (par.h);
	// `take(n)` supplies the large buffers (OrderedWriter::take recycles them)
	void maf_text(bool first, const QueryContig &q, ContigResult &r, const std::function<void(OutBuf &&)> &sink, const std::function<OutBuf(size_t)> &take) const;
	void maf_block(const QueryContig &q, ContigResult &r, gsa_block &b, OutBuf &small, const std::function<void(OutBuf &&)> &sink, const std::function<OutBuf(size_t)> &take) const;
	// OutputAlignment (tools.cpp:222-286)
	void aln(FILE *fp, const QueryContig &q, ContigResult &r) const;
	// OutputDotplot (DotPloting.cpp:10-71): gnuplot script `gp_path` + one data file per plotted reference sequence
	// (`<prefix>.<query>vs<chr>`); returns false when there is nothing to plot.  Running gnuplot on the script and removing
	// the data files afterwards (DotPloting.cpp:69-70) is the caller's business: `data_files` receives their names.
	bool dotplot(const std::string &gp_path, const std::string &out_prefix, const QueryContig &q, const ContigResult &r, std::vector<std::string> *data_files = nullptr) const;
	// VariantIdentification (SeqVariant.cpp:12-119); long blocks are dealt to the pool in record ranges
	void variants(int query_idx, const QueryContig &q, ContigResult &r);
	// OutputSequenceVariants (SeqVariant.cpp:121-143)
	void vcf(FILE *fp, const std::string &reference_label);
	void vcf_text(const std::string &reference_label, const std::function<void(OutBuf &&)> &sink);
};

#endif
