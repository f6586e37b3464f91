// gsalign_amd/csrc/host/emit.cpp -- query FASTA loader and the MAF / ALN / VCF
// emitters.  CPU code by design (north_star keeps MAF/VCF emission on the host).
//
// Follows, quirk for quirk (SURVEY.md App. A.7, App. B #4,#15-#21):
//   LoadQueryFile / TrimChromosomeName / CheckQuerySeq   reference src/main.cpp:35-114
//   OutputMAF / OutputAlignment / SelfComplementarySeq   src/tools.cpp:3-44,149-286
//   VariantIdentification / OutputSequenceVariants       src/SeqVariant.cpp:6-143
#include <algorithm>
#include <array>
#include <atomic>
#include <cstring>
#include <fstream>
#include <memory>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include "gsa_host.h"
#include "par.h"
#include "exact_sort.h"

namespace {

// blocks / lists / files below this size are handled by the calling thread alone.  GSA_HOST_PAR_MIN (read at every call: the CPU tests switch it)
// forces the parallel forms on the small goldens
size_t par_min_bytes() { const char *e = getenv("GSA_HOST_PAR_MIN"); return e ? (size_t)atoll(e) : (size_t)1 << 20; }

inline int nt4(unsigned char c)
{
	switch (c | 0x20) { case 'a': return 0; case 'c': return 1; case 'g': return 2; case 't': return 3; default: return 4; }
}

// tools.cpp:3-31: ACGTN/acgtn -> upper-case complement, U/u -> A, '-' stays, everything else NUL
inline char rev_map(char c)
{
	switch (c) {
	case 'A': case 'a': return 'T'; case 'C': case 'c': return 'G'; case 'G': case 'g': return 'C'; case 'T': case 't': return 'A';
	case 'U': case 'u': return 'A'; case 'N': case 'n': return 'N'; case '-': return '-';
	default: return '\0';
	}
}

// TrimChromosomeName (main.cpp:35-47): '|' -> '-', cut at space # : = tab
std::string trim_name(const char *p, size_t n)
{
	std::string name(p, n); size_t i;
	for (i = 0; i < name.size(); i++) {
		if (name[i] == '|') name[i] = '-';
		else if (name[i] == ' ' || name[i] == '#' || name[i] == ':' || name[i] == '=' || name[i] == '\t') break;
	}
	return name.substr(0, i);
}

int count_gaps(const std::string &s, int i, int stop) { int n = 0; for (; i < stop; i++) if (s[i] == '-') n++; return n; }

// the two text lines of a block, built exactly like OutputMAF does (tools.cpp:168-184)
void block_text(const QueryContig &q, const ContigResult &r, const gsa_block &b, std::string &t1, std::string &t2)
{
	t1.clear(); t2.clear();
	for (int k = 0; k < b.n_frag; k++) {
		const gsa_frag f = r.frag(b.frag_off + k);
		if (f.bseed) { t1.append(q.seq, f.qpos, f.qlen); t2.append(q.seq, f.qpos, f.qlen); }     // seeds print the QUERY text on both lines (App. B #4)
		else { t1.append(r.aln1.data() + f.aln_off, (size_t)f.aln_len); t2.append(r.aln2.data() + f.aln_off, (size_t)f.aln_len); }
	}
}

// iExtension (tools.cpp:192-202): a block running past the end of its chromosome copy is trimmed
int extension(const HostIndex &ix, const gsa_block &b, const gsa_frag &last)
{
	const int64_t end = last.rpos + last.rlen;
	const int64_t lim = (b.bdir ? ix.chr_fwd[b.chr] : ix.chr_rev[b.chr]) + ix.chr_len[b.chr];
	return end > lim ? (int)(end - lim) : 0;
}

} // namespace

// what a GPU worker thread does inside gsa_align_many's callback: the result is valid during the call only, so its bytes are copied -- and
// nothing else.  The records stay in their 16-byte form: the emitters expand the one they look at (frag()), 70 M records of a human genome
// are never materialised as 40-byte FragPair_t copies

template <class T> void RawArr<T>::assign(const T *src, size_t count)
{
	reset();
	if (count == 0) return;
	p = (T *)malloc(count * sizeof(T)); if (!p) abort();
	n = count;
	const size_t bytes = count * sizeof(T);
	char *d = (char *)p; const char *sp = (const char *)src;
	if (bytes < ((size_t)8 << 20)) { memcpy(d, sp, bytes); return; }
	par_ranges(bytes, (size_t)2 << 20, [&](size_t b, size_t e) { memcpy(d + b, sp + b, e - b); });
}
template struct RawArr<gsa_rec>;
template struct RawArr<char>;
void ContigResult::assign(const gsa_result &r)
{
	blocks.assign(r.blocks, r.blocks + r.n_blocks);
	recs.assign(r.recs, (size_t)r.n_frags);
	aln1.assign(r.aln1, (size_t)r.n_aln); aln2.assign(r.aln2, (size_t)r.n_aln);
}

// iExtension's trim of a block's last record (tools.cpp:192-202): a seed's one length, or a gap's two
void ContigResult::trim(int64_t i, int ext)
{
	gsa_rec &x = recs[(size_t)i];
	// A block's records are seed [gap] seed ... seed (gsa_hip.h): the last one is a seed, whose tag is its qpos -- its length may go to
	// zero or below exactly as the reference's qLen / rLen do.  A gap's tag is the SIGN of nqlen (= -1 - qLen): it never crosses zero here,
	// so a record can not change its kind under a later gsa_rec_expand.
	if (x.seed.qpos >= 0) x.seed.len -= ext;
	else { x.gap.rlen -= ext; x.gap.nqlen = (int64_t)x.gap.nqlen + ext > -1 ? -1 : x.gap.nqlen + ext; }
}

// LoadQueryFile (main.cpp:82-114) line by line -- getline on '\n', empty lines skipped, a line that starts with '>' opens a sequence
// (TrimChromosomeName), every other line loses ONE trailing '\r', must be all isalpha (CheckQuerySeq) and is appended -- but on the whole
// file at once: the file is read in slices by the pool's threads, cut into segments at line starts, every segment validates and counts
// its lines (pass 1), the sequences get their sizes, every segment copies its lines to where they belong (pass 2).  3 GB of FASTA: the
// reference's getline loop takes ~10 s, this takes the time of two passes over the page cache.
bool gsah_load_query(const std::string &path, std::vector<QueryContig> &out, std::string &err)
{
	const int fd = open(path.c_str(), O_RDONLY);
	if (fd < 0) { err = "cannot open " + path; return false; }
	struct stat st;
	if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
		// not a regular file (a pipe): the serial reader
		close(fd);
		std::ifstream file(path.c_str());
		if (!file.is_open()) { err = "cannot open " + path; return false; }
		std::string str; bool first = true;
		while (std::getline(file, str)) {
			if (first) { first = false; if (str.empty() || str[0] != '>') { err = "not a FASTA file: " + path; return false; } }
			if (str.empty()) continue;
			if (str[0] == '>') { QueryContig qc; qc.name = trim_name(str.data() + 1, str.size() - 1); out.push_back(qc); }
			else {
				if (str[str.size() - 1] == '\r') str.resize(str.size() - 1);
				for (size_t i = 0; i < str.size(); i++) if (!isalpha((unsigned char)str[i])) { err = "The query sequence contains non-alphabet characters!"; return false; }
				if (out.empty()) { err = "sequence before the first header"; return false; }
				out.back().seq.append(str);
			}
		}
		if (out.empty()) { err = "no sequence in " + path; return false; }
		return true;
	}
	const size_t n = (size_t)st.st_size;
	if (n == 0) { close(fd); err = "not a FASTA file: " + path; return false; }
	std::unique_ptr<char[]> buf(new char[n]);
	std::atomic<bool> io_ok(true);
	par_ranges(n, (size_t)8 << 20, [&](size_t b, size_t e) {
		while (b < e) { const ssize_t r = pread(fd, buf.get() + b, e - b, (off_t)b); if (r <= 0) { io_ok = false; return; } b += (size_t)r; }
	});
	close(fd);
	if (!io_ok) { err = "cannot read " + path; return false; }
	const char *d = buf.get();
	if (d[0] != '>') { err = "not a FASTA file: " + path; return false; }      // (the first line is empty or no header: CheckInputFile, main.cpp:49-64)
	// segments: [seg[k], seg[k + 1]) each starting at a line start
	const size_t want = std::min<size_t>((size_t)HostPool::global().threads() * 4, n / std::max<size_t>(par_min_bytes(), 1) + 1);
	std::vector<size_t> seg(1, 0);
	for (size_t k = 1; k < want; k++) {
		size_t p = n * k / want; if (p <= seg.back()) continue;
		const char *nl = (const char *)memchr(d + p, '\n', n - p);
		if (!nl) break;
		p = (size_t)(nl - d) + 1;
		if (p > seg.back() && p < n) seg.push_back(p);
	}
	seg.push_back(n);
	const size_t S = seg.size() - 1;
	struct Piece { size_t hdr_off, hdr_len; size_t bases; };         // a header line (hdr_len = 0 for the headless first piece of a segment) and the bases that follow it in the segment
	std::vector<std::vector<Piece> > pieces(S);
	std::atomic<bool> bad(false);
	static const struct AlphaTab { bool ok[256]; AlphaTab() { for (int c = 0; c < 256; c++) ok[c] = isalpha(c) != 0; } } alpha;
	// calls line(ptr, len) for every line of the segment (without its '\n')
	auto each_line = [&](size_t k, auto &&line) {
		size_t p = seg[k]; const size_t e = seg[k + 1];
		while (p < e) {
			const char *nl = (const char *)memchr(d + p, '\n', e - p);
			const size_t le = nl ? (size_t)(nl - d) : e;
			line(p, le - p);
			p = le + 1;
		}
	};
	HostPool::global().run(S, [&](size_t k) {
		std::vector<Piece> &pc = pieces[k];
		pc.push_back(Piece{ 0, 0, 0 });
		each_line(k, [&](size_t p, size_t len) {
			if (len == 0) return;
			if (d[p] == '>') { pc.push_back(Piece{ p, len, 0 }); return; }
			if (d[p + len - 1] == '\r') len--;
			bool ok = true;
			for (size_t i = 0; i < len; i++) ok &= alpha.ok[(unsigned char)d[p + i]];
			if (!ok) bad = true;
			pc.back().bases += len;
		});
	});
	if (bad) { err = "The query sequence contains non-alphabet characters!"; return false; }
	// sequences and the place of every piece in them
	const size_t base = out.size();                  // (a caller may hand in a vector that already holds contigs)
	std::vector<std::vector<size_t> > dst(S), cidx(S);
	std::vector<size_t> total;
	for (size_t k = 0; k < S; k++) {
		dst[k].assign(pieces[k].size(), 0); cidx[k].assign(pieces[k].size(), 0);
		for (size_t j = 0; j < pieces[k].size(); j++) {
			const Piece &pc = pieces[k][j];
			if (pc.hdr_len) { QueryContig qc; qc.name = trim_name(d + pc.hdr_off + 1, pc.hdr_len - 1); out.push_back(qc); total.push_back(0); }
			if (total.empty()) { if (pc.bases) { err = "sequence before the first header"; return false; } continue; }
			cidx[k][j] = total.size() - 1; dst[k][j] = total.back(); total.back() += pc.bases;
		}
	}
	if (total.empty()) { err = "no sequence in " + path; return false; }
	HostPool::global().run(total.size(), [&](size_t c) { out[base + c].seq.resize(total[c]); });
	HostPool::global().run(S, [&](size_t k) {
		size_t j = 0;
		char *w = pieces[k][0].bases ? &out[base + cidx[k][0]].seq[0] + dst[k][0] : nullptr;
		each_line(k, [&](size_t p, size_t len) {
			if (len == 0) return;
			if (d[p] == '>') { j++; w = pieces[k][j].bases ? &out[base + cidx[k][j]].seq[0] + dst[k][j] : nullptr; return; }
			if (d[p + len - 1] == '\r') len--;
			if (len) { memcpy(w, d + p, len); w += len; }
		});
	});
	return true;
}

// One block of OutputMAF (tools.cpp:164-216) as bytes.  The two text lines are filled piece by piece -- a seed prints the QUERY text on both
// lines, a gap its two gapped strings -- by the pool's threads, each piece at the offset the serial concatenation gives it; trimming
// (iExtension), the '-' counts, the reverse-complement of a reverse-strand block (SelfComplementarySeq: in place, pairs (i, L-1-i) are
// independent) and the NUL quirk (an unknown byte maps to '\0' and "%s" stops there) are the serial code's.
void Emitter::maf_block(const QueryContig &q, ContigResult &r, gsa_block &b, OutBuf &small, const std::function<void(OutBuf &&)> &sink, const std::function<OutBuf(size_t)> &take) const
{
	const int64_t f0 = b.frag_off;
	const size_t nf = (size_t)b.n_frag;
	std::vector<size_t> off(nf + 1); off[0] = 0;
	auto piece = [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; k++) { const gsa_frag f = r.frag(f0 + (int64_t)k); off[k + 1] = (size_t)(f.bseed ? f.qlen : f.aln_len); } };
	if (nf * 16 >= par_min_bytes()) par_ranges(nf, (size_t)1 << 16, piece); else piece(0, nf);
	for (size_t k = 0; k < nf; k++) off[k + 1] += off[k];
	const size_t total = off[nf];
	gsa_frag last = r.frag(f0 + (int64_t)nf - 1);
	const int ext = extension(*idx, b, last);
	if (ext > 0) { b.aln_len -= ext; b.score -= ext; last.rlen -= ext; last.qlen -= ext; r.trim(f0 + (int64_t)nf - 1, ext); }
	const size_t L = (size_t)(b.aln_len > 0 ? b.aln_len : 0);
	const bool big = 2 * L >= par_min_bytes();
	OutBuf t1 = big ? take(L + 1) : OutBuf(L + 1), t2 = big ? take(L + 1) : OutBuf(L + 1);
	const size_t fillable = total < L ? total : L;
	auto fill = [&](size_t pb, size_t pe) {      // text positions [pb, pe)
		size_t k = (size_t)(std::upper_bound(off.begin(), off.end(), pb) - off.begin()) - 1;
		for (size_t p = pb; p < pe; k++) {
			const size_t o = p - off[k], n = std::min(off[k + 1], pe) - p;
			if (n == 0) continue;
			const gsa_frag f = r.frag(f0 + (int64_t)k);
			if (f.bseed) { memcpy(t1.p + p, q.seq.data() + f.qpos + o, n); memcpy(t2.p + p, q.seq.data() + f.qpos + o, n); }
			else { memcpy(t1.p + p, r.aln1.data() + f.aln_off + o, n); memcpy(t2.p + p, r.aln2.data() + f.aln_off + o, n); }
			p += n;
		}
	};
	std::atomic<long long> g1(0), g2(0);
	auto gaps = [&](size_t pb, size_t pe) { long long a = 0, c = 0; for (size_t i = pb; i < pe; i++) { a += t1.p[i] == '-'; c += t2.p[i] == '-'; } g1 += a; g2 += c; };
	auto revpair = [&](char *s, size_t ib, size_t ie) { for (size_t i = ib; i < ie; i++) { const size_t j = L - 1 - i; const char a = s[i], c = s[j]; s[i] = rev_map(c); s[j] = rev_map(a); } };
	if (big) {
		par_ranges(fillable, (size_t)1 << 18, fill);
		if (L > fillable) { memset(t1.p + fillable, 0, L - fillable); memset(t2.p + fillable, 0, L - fillable); }      // (std::string::resize pads with NULs)
		par_ranges(L, (size_t)1 << 18, gaps);
		if (!b.bdir) {
			par_ranges(L / 2, (size_t)1 << 18, [&](size_t ib, size_t ie) { revpair(t1.p, ib, ie); revpair(t2.p, ib, ie); });
			if (L & 1) { t1.p[L / 2] = rev_map(t1.p[L / 2]); t2.p[L / 2] = rev_map(t2.p[L / 2]); }
		}
	} else {
		if (fillable) fill(0, fillable);
		if (L > fillable) { memset(t1.p + fillable, 0, L - fillable); memset(t2.p + fillable, 0, L - fillable); }
		gaps(0, L);
		if (!b.bdir) { revpair(t1.p, 0, L / 2); revpair(t2.p, 0, L / 2); if (L & 1) { t1.p[L / 2] = rev_map(t1.p[L / 2]); t2.p[L / 2] = rev_map(t2.p[L / 2]); } }
	}
	// "%s": the line ends at the first NUL
	const size_t n1 = (size_t)((const char *)memchr(t1.p, 0, L) ? (const char *)memchr(t1.p, 0, L) - t1.p : (ptrdiff_t)L);
	const size_t n2 = (size_t)((const char *)memchr(t2.p, 0, L) ? (const char *)memchr(t2.p, 0, L) - t2.p : (ptrdiff_t)L);
	const std::string &rname = idx->chr_name[b.chr];
	std::string qname = q.name;
	if (qname.size() <= rname.size()) qname.append(rname.size() - qname.size(), ' ');           // only the query name is printed padded (App. B #18)
	const int clen = idx->chr_len[b.chr]; const unsigned qlen = (unsigned)q.seq.size();
	std::vector<char> h1(rname.size() + 128), h2(qname.size() + 128);
	int l1, l2;
	if (b.bdir) {
		l1 = snprintf(h1.data(), h1.size(), "a score=%d\ns ref.%s %d %d + %d ", b.bdup ? 1 : b.score, rname.c_str(), b.gpos - 1, b.aln_len - (int)g1.load(), clen);
		l2 = snprintf(h2.data(), h2.size(), "\ns qry.%s %d %d + %d ", qname.c_str(), r.frag(f0).qpos, b.aln_len - (int)g2.load(), qlen);
	} else {
		const int64_t rpos = last.rpos + last.rlen - 1;
		int d, c, g; idx->coordinate(rpos, &d, &c, &g);
		l1 = snprintf(h1.data(), h1.size(), "a score=%d\ns ref.%s %d %d + %d ", b.bdup ? 1 : b.score, rname.c_str(), g - 1, b.aln_len - (int)g1.load(), clen);
		l2 = snprintf(h2.data(), h2.size(), "\ns qry.%s %d %d - %d ", qname.c_str(), qlen - (last.qpos + last.qlen), b.aln_len - (int)g2.load(), qlen);
	}
	if (!big) {
		small.append(h1.data(), (size_t)l1); small.append(t1.p, n1); small.append(h2.data(), (size_t)l2); small.append(t2.p, n2); small.append("\n\n", 2);
		if (small.n >= ((size_t)4 << 20)) { OutBuf out = std::move(small); small = OutBuf(); sink(std::move(out)); }
		return;
	}
	small.append(h1.data(), (size_t)l1);
	{ OutBuf out = std::move(small); small = OutBuf(); sink(std::move(out)); }
	t1.n = n1; sink(std::move(t1));
	small.append(h2.data(), (size_t)l2);
	{ OutBuf out = std::move(small); small = OutBuf(); sink(std::move(out)); }
	t2.n = n2; sink(std::move(t2));
	small.append("\n\n", 2);
}

void Emitter::maf_text(bool first, const QueryContig &q, ContigResult &r, const std::function<void(OutBuf &&)> &sink, const std::function<OutBuf(size_t)> &take) const
{
	OutBuf small;
	if (first) small.append("##maf version=1\n", 16);
	for (size_t bi = 0; bi < r.blocks.size(); bi++) {
		gsa_block &b = r.blocks[bi];
		if (!allow_dup && b.bdup) continue;
		maf_block(q, r, b, small, sink, take);
	}
	if (small.n) sink(std::move(small));
}

void Emitter::maf(FILE *fp, bool first, const QueryContig &q, ContigResult &r) const
{
	maf_text(first, q, r, [&](OutBuf &&o) { if (o.n) fwrite(o.p, 1, o.n, fp); }, [](size_t c) { return OutBuf(c); });
}

void Emitter::aln(FILE *fp, const QueryContig &q, ContigResult &r) const
{
	std::string t1, t2;
	for (size_t bi = 0; bi < r.blocks.size(); bi++) {
		gsa_block &b = r.blocks[bi];
		if (!allow_dup && b.bdup) continue;
		block_text(q, r, b, t1, t2);
		const unsigned full = (unsigned)t1.size();
		const gsa_frag last = r.frag(b.frag_off + b.n_frag - 1);
		const int ext = extension(*idx, b, last);
		if (ext > 0) { b.aln_len -= ext; b.score -= ext; r.trim(b.frag_off + b.n_frag - 1, ext); t1[b.aln_len] = t2[b.aln_len] = '\0'; }
		std::string rname = idx->chr_name[b.chr], qname = q.name;
		if (qname.size() > rname.size()) rname.append(qname.size() - rname.size(), ' '); else qname.append(rname.size() - qname.size(), ' ');
		fprintf(fp, "#Identity = %d / %d (%.2f%%) Orientation = %s\n\n", b.score, b.aln_len, (int)(1000 * (1.0 * b.score / b.aln_len)) / 10.0, b.bdir ? "Forward" : "Reverse");
		unsigned pos = 0; int qp = r.frag(b.frag_off).qpos + 1; long long rp = b.gpos;
		while (pos < full) {                                         // the reference loops over the UNtrimmed length (tools.cpp:275)
			const unsigned stop = pos + 80 > full ? full : pos + 80;
			const int p = 80 - count_gaps(t1, pos, stop), qn = 80 - count_gaps(t2, pos, stop);
			fprintf(fp, "ref.%s\t%12lld\t%.80s\nqry.%s\t%12d\t%.80s\n\n", rname.c_str(), rp, t1.c_str() + pos, qname.c_str(), qp, t2.c_str() + pos);
			pos += 80; rp += (b.bdir ? p : 0 - p); qp += qn;
		}
		fprintf(fp, "%s\n", std::string(100, '*').c_str());
	}
}

bool Emitter::dotplot(const std::string &gp_path, const std::string &out_prefix, const QueryContig &q, const ContigResult &r, std::vector<std::string> *data_files) const
{
	static const char *colors[10] = { "red", "blue", "web-green", "dark-magenta", "orange", "yellow", "turquoise", "dark-yellow", "violet", "dark-grey" };
	if (r.blocks.empty()) return false;
	const int nchr = (int)idx->chr_len.size();
	std::vector<int> sum((size_t)nchr, 0);
	for (const gsa_block &b : r.blocks) if (b.score > 0) sum[(size_t)b.chr] += b.score;
	std::vector<std::pair<int, int64_t> > top;                          // reference sequences with >= 1000 identical columns, best five
	for (int i = 0; i < nchr; i++) if (sum[(size_t)i] >= 1000) top.push_back(std::make_pair(i, (int64_t)sum[(size_t)i]));
	if (top.empty()) return false;
	std::sort(top.begin(), top.end(), [](const std::pair<int, int64_t> &a, const std::pair<int, int64_t> &b) { return a.second > b.second; });
	if (top.size() > 5) top.resize(5);
	FILE *gp = fopen(gp_path.c_str(), "w"); if (!gp) return false;
	const std::string data = out_prefix + "." + q.name;
	std::vector<FILE *> fh((size_t)nchr, (FILE *)NULL);
	for (size_t i = 0; i < top.size(); i++) {
		const std::string fn = data + "vs" + idx->chr_name[(size_t)top[i].first];
		fh[(size_t)top[i].first] = fopen(fn.c_str(), "w");
		if (fh[(size_t)top[i].first]) { fprintf(fh[(size_t)top[i].first], "0 0\n0 0\n\n"); if (data_files) data_files->push_back(fn); }
	}
	fprintf(gp, "set terminal postscript color solid 'Courier' 15\nset output '%s-%s.ps'\nset grid\nset border 1\n", out_prefix.c_str(), q.name.c_str());
	for (size_t i = 0; i < top.size(); i++) fprintf(gp, "set style line %d lw 4 pt 0 ps 0.5 lc '%s'\n", (int)i + 1, colors[i]);
	fprintf(gp, "set xrange[1:*]\nset yrange[1:*]\nset xlabel 'Query (%s)'\nset ylabel 'Ref'\n", q.name.c_str());
	fprintf(gp, "plot ");
	for (size_t i = 0; i < top.size(); i++) {
		const std::string &cn = idx->chr_name[(size_t)top[i].first];
		fprintf(gp, "'%svs%s' title '%s' with lp ls %d%s", data.c_str(), cn.c_str(), cn.c_str(), (int)i + 1, i + 1 != top.size() ? ", " : "\n\n");
	}
	for (const gsa_block &b : r.blocks) {
		if (b.score <= 0 || !fh[(size_t)b.chr]) continue;
		const gsa_frag first = r.frag(b.frag_off), last = r.frag(b.frag_off + b.n_frag - 1);
		int d, c, g0, g1;
		idx->coordinate(first.rpos, &d, &c, &g0); idx->coordinate(last.rpos + last.rlen - 1, &d, &c, &g1);
		fprintf(fh[(size_t)b.chr], "%d %d\n%d %d\n\n", first.qpos + 1, g0, last.qpos + last.qlen, g1);
	}
	for (FILE *f : fh) if (f) fclose(f);
	fclose(gp);
	return true;
}

// VariantIdentification for the records [kb, ke) of one block (SeqVariant.cpp:27-117): a record's variants depend on that record only
static void variants_of(const HostIndex *idx, int query_idx, const QueryContig &q, const ContigResult &r, const gsa_block &b, size_t kb, size_t ke, std::vector<Variant> &vars, int cnt[3])
{
	// Every allele the reference copies into a Variant (SeqVariant.cpp: ref_frag / alt_frag) is a piece of RefSequence or of the query sequence --
	// also the single columns it takes from the gapped strings, which hold exactly those bytes -- so a Variant points there instead of owning
	// two std::strings: 30 M variants of a human genome are 40 bytes each, written once.
	const char *ref = idx->ref.data(), *qs = q.seq.data();
	Variant v; v.chr_idx = b.chr; v.query_idx = query_idx;
	int d, c, g;
	for (size_t k = kb; k < ke; k++) {
		const gsa_frag f = r.frag(b.frag_off + (int64_t)k);
		if (f.bseed) continue;
		if (f.qlen == 0 && f.rlen == 0) continue;
		if (f.qlen == 0) {                                        // pure deletion (:36-45)
			cnt[2]++;
			v.type = 2; idx->coordinate(f.rpos - 1, &d, &c, &g); v.pos = g;
			v.ref_p = ref + f.rpos - 1; v.ref_n = (uint32_t)f.rlen + 1; v.alt_p = qs + f.qpos - 1; v.alt_n = 1;
			vars.push_back(v);
		} else if (f.rlen == 0) {                                 // pure insertion (:46-55)
			cnt[1]++;
			v.type = 1; idx->coordinate(f.rpos - 1, &d, &c, &g); v.pos = g;
			v.ref_p = ref + f.rpos - 1; v.ref_n = 1; v.alt_p = qs + f.qpos - 1; v.alt_n = (uint32_t)f.qlen + 1;
			vars.push_back(v);
		} else if (f.qlen == 1 && f.rlen == 1) {                  // 1x1 (:56-67)
			const char a1 = r.aln1[f.aln_off], a2 = r.aln2[f.aln_off];
			if (nt4(a1) != nt4(a2) && nt4(a2) != 4) {
				cnt[0]++;
				v.type = 0; idx->coordinate(f.rpos, &d, &c, &g); v.pos = g;
				v.ref_p = ref + f.rpos; v.ref_n = 1; v.alt_p = qs + f.qpos; v.alt_n = 1;      // (the column a1 / a2 = these two bytes)
				vars.push_back(v);
			}
		} else {                                                  // walk the aligned columns (:68-115)
			const char *x1 = r.aln1.data() + f.aln_off, *x2 = r.aln2.data() + f.aln_off;
			const int L = f.aln_len; int64_t rp = f.rpos; int qp = f.qpos;
			for (int i = 0; i < L; i++) {
				if (x1[i] == '-') {
					cnt[1]++;
					int n = 1; while (i + n < L && x1[i + n] == '-') n++;
					v.type = 1; idx->coordinate(rp - 1, &d, &c, &g); v.pos = g;
					v.ref_p = qs + qp - 1; v.ref_n = 1; v.alt_p = qs + qp - 1; v.alt_n = (uint32_t)n + 1;           // REF anchor comes from the QUERY (App. B #15)
					vars.push_back(v);
					qp += n; i += n - 1;
				} else if (x2[i] == '-') {
					cnt[2]++;
					int n = 1; while (i + n < L && x2[i + n] == '-') n++;
					v.type = 2; idx->coordinate(rp - 1, &d, &c, &g); v.pos = g;
					v.ref_p = ref + rp - 1; v.ref_n = (uint32_t)n + 1; v.alt_p = ref + rp - 1; v.alt_n = 1;
					vars.push_back(v);
					rp += n; i += n - 1;
				} else if (nt4(x1[i]) != nt4(x2[i])) {
					if (nt4(x2[i]) != 4) {
						cnt[0]++;
						v.type = 0; idx->coordinate(rp, &d, &c, &g); v.pos = g;
						v.ref_p = ref + rp; v.ref_n = 1; v.alt_p = qs + qp; v.alt_n = 1;      // (the literal column characters = these two bytes)
						vars.push_back(v);
					}
					rp++; qp++;
				} else { rp++; qp++; }
			}
		}
	}
}

// VarVec grows block after block, record after record (SeqVariant.cpp:12-119); here a long block's records are dealt to the pool in ranges
// and the ranges' lists are kept IN THAT ORDER (var_chunks): the sequence the final sort sees is the serial one.
void Emitter::variants(int query_idx, const QueryContig &q, ContigResult &r)
{
	for (size_t bi = 0; bi < r.blocks.size(); bi++) {
		const gsa_block &b = r.blocks[bi];
		if (b.bdup) continue;
		const size_t nf = (size_t)b.n_frag;
		if (nf * 16 < par_min_bytes()) {
			if (var_chunks.empty() || var_chunks.back().size() > ((size_t)1 << 20)) var_chunks.emplace_back();
			int cnt[3] = { 0, 0, 0 };
			variants_of(idx, query_idx, q, r, b, 0, nf, var_chunks.back(), cnt);
			n_snv += cnt[0]; n_ins += cnt[1]; n_del += cnt[2];
			continue;
		}
		const size_t parts = std::min<size_t>((size_t)HostPool::global().threads() * 4, nf / 4096 + 1);
		const size_t at = var_chunks.size();
		var_chunks.resize(at + parts);
		std::vector<std::array<int, 3> > cnts(parts, std::array<int, 3>{ { 0, 0, 0 } });
		HostPool::global().run(parts, [&](size_t k) {
			variants_of(idx, query_idx, q, r, b, nf * k / parts, nf * (k + 1) / parts, var_chunks[at + k], cnts[k].data());
		});
		for (size_t k = 0; k < parts; k++) { n_snv += cnts[k][0]; n_ins += cnts[k][1]; n_del += cnts[k][2]; }
	}
}

namespace {
inline char *put_int(char *p, long long v)
{
	char tmp[24]; int n = 0; bool neg = v < 0; unsigned long long u = neg ? 0ull - (unsigned long long)v : (unsigned long long)v;
	do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
	if (neg) *p++ = '-';
	while (n) *p++ = tmp[--n];
	return p;
}
}

// OutputSequenceVariants (SeqVariant.cpp:121-143).  The reference std::sorts VarVec on (chr_idx, pos) -- an incomplete key: the order of
// variants that share a position is whatever libstdc++'s introsort leaves (App. B #17).  What introsort does depends on the comparator's
// answers only, never on what else an element carries, so sorting 16-byte keys {chr, pos, where the variant lies} with the same comparator
// on the same initial sequence makes the same comparisons and the same moves: the permutation is the reference's, without dragging two
// std::strings per element through every swap; and introsort's partition steps split the array into parts that never meet again, so they run
// on the pool's threads (exact_sort.h: std::sort's permutation, element for element).  The lines are then formatted by the pool in ranges and
// written in order.
void Emitter::vcf_text(const std::string &reference_label, const std::function<void(OutBuf &&)> &sink)
{
	static const char *MutType[3] = { "SUBSTITUTE", "INSERT", "DELETE" };
	struct Key { int32_t chr, pos; uint32_t chunk, at; };
	struct ByPos { bool operator()(const Key &a, const Key &b) const { return a.chr == b.chr ? a.pos < b.pos : a.chr < b.chr; } };
	if (!vars.empty()) { var_chunks.insert(var_chunks.begin(), std::vector<Variant>()); var_chunks.front().swap(vars); }      // (variants pushed into `vars` directly come first)
	std::vector<size_t> base(var_chunks.size() + 1, 0);
	for (size_t c = 0; c < var_chunks.size(); c++) base[c + 1] = base[c] + var_chunks[c].size();
	const size_t n = base.back();
	std::vector<Key> keys(n);
	HostPool::global().run(var_chunks.size(), [&](size_t c) {
		const std::vector<Variant> &vc = var_chunks[c]; Key *k = keys.data() + base[c];
		for (size_t i = 0; i < vc.size(); i++) { k[i].chr = vc[i].chr_idx; k[i].pos = vc[i].pos; k[i].chunk = (uint32_t)c; k[i].at = (uint32_t)i; }
	});
	exact_sort(keys.data(), keys.data() + keys.size(), ByPos());      // std::sort's permutation (same incomplete key, same initial order), on the pool: exact_sort.h
	{
		OutBuf h;
		h.append("##fileformat=VCFv4.1\n"); h.append("##reference=" + reference_label + "\n"); h.append("##source=GSAlign 1.0.22\n");
		h.append("##INFO=<ID=TYPE,Number=1,Type=String,Description=\"The type of allele, either SUBSTITUTE, INSERT, or DELETE.\">\n");
		for (size_t i = 0; i < idx->chr_name.size(); i++) h.append("##contig=<ID=" + idx->chr_name[i] + ",length=" + std::to_string(idx->chr_len[i]) + ">\n");
		h.append("#CHROM	POS	ID	REF	ALT	QUAL	FILTER	INFO\n");
		sink(std::move(h));
	}
	const size_t per = (size_t)1 << 16, parts = (n + per - 1) / per;
	// (ranges are formatted a batch at a time so that the writer gets them in order while the next batch is formatted)
	const size_t batch = (size_t)HostPool::global().threads() * 2;
	for (size_t p0 = 0; p0 < parts; p0 += batch) {
		const size_t pn = std::min(batch, parts - p0);
		std::vector<OutBuf> outs(pn);
		auto one = [&](size_t j) {
			const size_t b = (p0 + j) * per, e = std::min(n, b + per);
			OutBuf &o = outs[j]; o.reserve((e - b) * 48);
			for (size_t i = b; i < e; i++) {
				const Variant &v = var_chunks[keys[i].chunk][keys[i].at];
				const std::string &cn = idx->chr_name[(size_t)v.chr_idx];
				const char *mt = MutType[v.type]; const size_t ml = strlen(mt);
				const size_t rl = v.ref_n, al = v.alt_n;      // (RefSequence and a validated query hold no NUL: "%s" prints the whole fragment)
				const size_t need = cn.size() + rl + al + ml + 48;
				if (o.n + need > o.cap) o.reserve((o.n + need) * 2);
				char *w = o.p + o.n;
				memcpy(w, cn.data(), cn.size()); w += cn.size(); *w++ = '\t';
				w = put_int(w, v.pos); memcpy(w, "\t.\t", 3); w += 3;
				memcpy(w, v.ref_p, rl); w += rl; *w++ = '\t';
				memcpy(w, v.alt_p, al); w += al;
				memcpy(w, "\t100\t*\tTYPE=", 12); w += 12;
				memcpy(w, mt, ml); w += ml; *w++ = '\n';
				o.n = (size_t)(w - o.p);
			}
		};
		if (n * 48 < par_min_bytes()) for (size_t j = 0; j < pn; j++) one(j); else HostPool::global().run(pn, one);
		for (size_t j = 0; j < pn; j++) sink(std::move(outs[j]));
	}
}

void Emitter::vcf(FILE *fp, const std::string &reference_label)
{
	vcf_text(reference_label, [&](OutBuf &&o) { if (o.n) fwrite(o.p, 1, o.n, fp); });
}
