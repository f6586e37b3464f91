// gsalign_amd/csrc/host/index_io.cpp -- BWA-style index files: loader and builder.
//
// Loader  : reference src/bwt_index.cpp:25-264 (bwa_idx_load, RestoreReferenceInfo),
//           src/GetData.cpp:8-24 (CheckBWAIndexFiles).
// Builder : reference src/BWT_Index/bwtindex.c:77-149 (bwa_idx_build),
//           bntseq.c:59-88,110-211 (pack FASTA, N -> lrand48()&3 with seed 11,
//           .ann/.amb text), bwt.c:101-123 (sampled SA).
// The reference grows the BWT incrementally (BWT-SW, bwt_gen.c); the files it
// produces are mathematically determined by the FASTA, so this
// builder sorts all suffixes with a linear-time SA-IS and derives BWT, Occ and
// the SA samples from that -- the output is byte-identical.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include "gsa_host.h"
#include "par.h"

namespace {

inline int nt4(unsigned char c)
{
	switch (c | 0x20) { case 'a': return 0; case 'c': return 1; case 'g': return 2; case 't': return 3; default: return 4; }
}

// ---- SA-IS (Nong, Zhang, Chan): suffix array of T[0..n), T[n-1] = unique smallest ----
// I = index type: int32_t up to 2^31 - 2 suffixes (a reference of 1 Gbp: the text is forward + reverse complement + '$'),
// int64_t beyond (a 3.1 Gbp human reference = 6.2 G suffixes, 8 bytes each).
template <class Ch, class I>
void induce(const Ch *T, I *SA, I n, I K, const std::vector<bool> &isS, std::vector<I> &bkt, const std::vector<I> &cnt)
{
	I sum = 0;
	for (I c = 0; c < K; c++) { bkt[c] = sum; sum += cnt[c]; }                 // bucket starts
	for (I i = 0; i < n; i++) { I j = SA[i] - 1; if (SA[i] > 0 && !isS[j]) SA[bkt[T[j]]++] = j; }
	sum = 0;
	for (I c = 0; c < K; c++) { sum += cnt[c]; bkt[c] = sum; }                 // bucket ends
	for (I i = n - 1; i >= 0; i--) { I j = SA[i] - 1; if (SA[i] > 0 && isS[j]) SA[--bkt[T[j]]] = j; }
}

template <class Ch, class I>
void sais(const Ch *T, I *SA, I n, I K)
{
	if (n == 1) { SA[0] = 0; return; }
	std::vector<bool> isS((size_t)n);
	isS[n - 1] = true;
	for (I i = n - 2; i >= 0; i--) isS[i] = T[i] < T[i + 1] || (T[i] == T[i + 1] && isS[i + 1]);
	auto isLMS = [&](I i) { return i > 0 && isS[i] && !isS[i - 1]; };
	std::vector<I> cnt((size_t)K, 0), bkt((size_t)K);
	for (I i = 0; i < n; i++) cnt[T[i]]++;
	auto ends = [&]() { I s = 0; for (I c = 0; c < K; c++) { s += cnt[c]; bkt[c] = s; } };
	// 1. sort LMS substrings
	std::fill(SA, SA + n, (I)-1);
	ends();
	for (I i = 1; i < n; i++) if (isLMS(i)) SA[--bkt[T[i]]] = i;
	induce(T, SA, n, K, isS, bkt, cnt);
	I n1 = 0;
	for (I i = 0; i < n; i++) if (isLMS(SA[i])) SA[n1++] = SA[i];
	std::fill(SA + n1, SA + n, (I)-1);
	I name = 0, prev = -1;
	for (I i = 0; i < n1; i++) {
		I pos = SA[i]; bool diff = false;
		if (prev < 0) diff = true;
		else for (I d = 0;; d++) {
			if (pos + d >= n || prev + d >= n || T[pos + d] != T[prev + d] || isS[pos + d] != isS[prev + d]) { diff = true; break; }
			if (d > 0 && (isLMS(pos + d) || isLMS(prev + d))) { diff = !(isLMS(pos + d) && isLMS(prev + d)); break; }
		}
		if (diff) { name++; prev = pos; }
		SA[n1 + (pos >> 1)] = name - 1;
	}
	for (I i = n - 1, j = n - 1; i >= n1; i--) if (SA[i] >= 0) SA[j--] = SA[i];
	I *s1 = SA + n - n1, *SA1 = SA;
	// 2. order of the LMS suffixes
	if (name < n1) sais<I, I>(s1, SA1, n1, name);
	else for (I i = 0; i < n1; i++) SA1[s1[i]] = i;
	// 3. induce everything from the sorted LMS suffixes
	for (I i = 1, j = 0; i < n; i++) if (isLMS(i)) s1[j++] = i;
	for (I i = 0; i < n1; i++) SA1[i] = s1[SA1[i]];
	std::fill(SA + n1, SA + n, (I)-1);
	ends();
	for (I i = n1 - 1; i >= 0; i--) { I j = SA[i]; SA[i] = -1; SA[--bkt[T[j]]] = j; }
	induce(T, SA, n, K, isS, bkt, cnt);
}

// ---- host threads ----
// GSA_INDEX_THREADS (default: all hardware threads, at most 128); 1 = the serial builder (SA-IS)
int index_threads()
{
	if (const char *e = getenv("GSA_INDEX_THREADS")) { const int v = atoi(e); if (v >= 1) return v; }
	unsigned h = std::thread::hardware_concurrency();
	return (int)(h < 1 ? 1 : (h > 128 ? 128 : h));
}
// fn(thread, begin, end) over [0, n) cut into one contiguous range per thread
template <class F> void parallel_ranges(int64_t n, int nt, F fn)
{
	if (nt <= 1 || n < 65536) { fn(0, (int64_t)0, n); return; }
	std::vector<std::thread> th;
	for (int t = 0; t < nt; t++) { const int64_t a = n * t / nt, b = n * (t + 1) / nt; th.emplace_back([=] { fn(t, a, b); }); }
	for (std::thread &x : th) x.join();
}

// ---- parallel suffix sorter (round 3) ----
// The files of the index are determined by the suffix ORDER, not by how it was found, so references above a few Mb are
// sorted on all host cores: suffixes are dealt into 4^8 buckets by their first 8 bases (two parallel passes over the text),
// every bucket is sorted by one thread -- first on the next 32 bases as one 64-bit key, then the runs of equal keys by
// comparing the 2-bit packed texts 32 bases per word -- and threads take buckets from a shared counter.  A suffix that runs
// into the end of the text is SMALLER than a longer one with the same prefix ('$' sorts first); the packed text is padded
// with A = 0, the smallest letter, so padded keys order such pairs correctly or tie, and ties are settled by length.
// Cost grows with the depth of the comparisons (long exact repeats); a 3.1 Gbp reference (6.2 G suffixes) takes about a minute
// on 128 threads where SA-IS on one thread took 17.
struct PackedText {
	std::vector<uint64_t> w; int64_t S = 0;          // base p (code 0..3) at bits 62 - 2 (p & 31) of word p >> 5
	// Round 4 (ADVICE): a comparison that runs deeper than `depth_limit` bases gives the sorter up -- exact repeats of megabase length (identical
	// haplotigs, a duplicated chromosome) make the comparisons of a bucket cost copies x L^2 / 32 word compares, hours where SA-IS is linear --
	// and the caller falls back to SA-IS for the whole text.
	int64_t depth_limit = 1ll << 22; std::atomic<bool> *give_up = nullptr;
	inline uint64_t get32(int64_t p) const           // the 32 bases from p on, zero beyond the end
	{
		if (p >= S) return 0;
		const int64_t wi = p >> 5; const int sh = (int)(p & 31) * 2;
		return sh ? (w[(size_t)wi] << sh) | (w[(size_t)wi + 1] >> (64 - sh)) : w[(size_t)wi];
	}
	// suffix i < suffix j, both known equal on their first d bases
	inline bool less_from(int64_t i, int64_t j, int64_t d) const
	{
		for (;; d += 32) {
			if (d > depth_limit) { if (give_up) give_up->store(true, std::memory_order_relaxed); return i < j; }      // (the order no longer matters: the result is thrown away)
			const int64_t ri = S - (i + d), rj = S - (j + d);
			if (ri <= 0 || rj <= 0) return ri < rj;
			uint64_t a = get32(i + d), b = get32(j + d);
			const int64_t m = ri < rj ? (ri < 32 ? ri : 32) : (rj < 32 ? rj : 32);
			if (m < 32) { const uint64_t mask = ~0ull << (64 - 2 * m); a &= mask; b &= mask; }
			if (a != b) return a < b;
			if (m < 32) return ri < rj;
		}
	}
};

template <class I>
bool parallel_suffix_sort(const std::vector<uint8_t> &T, int64_t S, I *SA, int nt)
{
	PackedText P; P.S = S; P.w.assign((size_t)(S / 32 + 3), 0);
	std::atomic<bool> give_up(false); P.give_up = &give_up;
	if (const char *e = getenv("GSA_INDEX_DEPTH")) { const long long v = atoll(e); if (v >= 64) P.depth_limit = v; }      // (tests: force the fallback)
	parallel_ranges((S + 31) / 32, nt, [&](int, int64_t a, int64_t b) {
		for (int64_t wi = a; wi < b; wi++) {
			uint64_t v = 0; const int64_t p0 = wi * 32, p1 = p0 + 32 < S ? p0 + 32 : S;
			for (int64_t p = p0; p < p1; p++) v |= (uint64_t)(T[(size_t)p] - 1) << (62 - 2 * (p - p0));
			P.w[(size_t)wi] = v;
		}
	});
	const int NB = 1 << 16;
	std::vector<std::vector<int64_t> > hist((size_t)nt, std::vector<int64_t>((size_t)NB, 0));
	parallel_ranges(S, nt, [&](int t, int64_t a, int64_t b) { int64_t *h = hist[(size_t)t].data(); for (int64_t i = a; i < b; i++) h[P.get32(i) >> 48]++; });
	std::vector<int64_t> base((size_t)NB + 1);
	{
		int64_t at = 1;                                   // SA[0] = the suffix that is '$' alone
		for (int bkt = 0; bkt < NB; bkt++) { base[(size_t)bkt] = at; for (int t = 0; t < nt; t++) { const int64_t c = hist[(size_t)t][(size_t)bkt]; hist[(size_t)t][(size_t)bkt] = at; at += c; } }
		base[(size_t)NB] = at;
	}
	SA[0] = (I)S;
	parallel_ranges(S, nt, [&](int t, int64_t a, int64_t b) { int64_t *o = hist[(size_t)t].data(); for (int64_t i = a; i < b; i++) SA[o[P.get32(i) >> 48]++] = (I)i; });
	{ std::vector<std::vector<int64_t> >().swap(hist); }
	// the buckets, largest first (a bucket is one thread's job: the heavy ones must not come last)
	std::vector<int> order((size_t)NB);
	for (int b = 0; b < NB; b++) order[(size_t)b] = b;
	std::sort(order.begin(), order.end(), [&](int x, int y) { const int64_t sx = base[(size_t)x + 1] - base[(size_t)x], sy = base[(size_t)y + 1] - base[(size_t)y]; return sx != sy ? sx > sy : x < y; });
	std::atomic<int> next(0);
	auto worker = [&]() {
		struct KI { uint64_t key; I idx; };
		std::vector<KI> buf;
		for (;;) {
			const int o = next.fetch_add(1); if (o >= NB || give_up.load(std::memory_order_relaxed)) return;
			const int bkt = order[(size_t)o];
			const int64_t lo = base[(size_t)bkt], m = base[(size_t)bkt + 1] - lo;
			if (m < 2) continue;
			buf.resize((size_t)m);
			for (int64_t k = 0; k < m; k++) { const I i = SA[lo + k]; buf[(size_t)k].idx = i; buf[(size_t)k].key = P.get32((int64_t)i + 8); }
			std::sort(buf.begin(), buf.end(), [](const KI &x, const KI &y) { return x.key < y.key; });
			for (int64_t k = 0; k < m;) {
				int64_t e = k + 1; while (e < m && buf[(size_t)e].key == buf[(size_t)k].key) e++;
				if (e - k > 1) std::sort(buf.begin() + k, buf.begin() + e, [&](const KI &x, const KI &y) { return P.less_from((int64_t)x.idx, (int64_t)y.idx, 8); });
				k = e;
			}
			for (int64_t k = 0; k < m; k++) SA[lo + k] = buf[(size_t)k].idx;
		}
	};
	std::vector<std::thread> th;
	for (int t = 1; t < nt; t++) th.emplace_back(worker);
	worker();
	for (std::thread &x : th) x.join();
	return !give_up.load();
}

// BWT without '$' (2 bits per symbol, MSB first in each word), primary row and the SA samples of every 32nd row, from the
// suffix array of T = text + '$' (symbols 1..4, T[S] = 0)
template <class I>
void derive_bwt_sa(const std::vector<uint8_t> &T, int64_t S, std::vector<uint32_t> &packed, uint64_t &primary, std::vector<uint64_t> &sa)
{
	const I n = (I)(S + 1);
	std::vector<I> SA((size_t)n);
	const int nt = index_threads();
	// (GSA_INDEX_PAR_MIN: the tests send small fixtures through the parallel sorter too)
	const char *pm = getenv("GSA_INDEX_PAR_MIN");
	bool sorted = false;
	if (nt > 1 && S >= (pm ? atoll(pm) : (1ll << 20))) sorted = parallel_suffix_sort<I>(T, S, SA.data(), nt);
	if (!sorted) sais<uint8_t, I>(T.data(), SA.data(), n, (I)5);      // (small texts, one thread, or comparisons that ran too deep: linear time whatever the repeats)
	std::atomic<int64_t> prim(-1);
	parallel_ranges((int64_t)n, nt, [&](int, int64_t a, int64_t b) { for (int64_t i = a; i < b; i++) if (SA[(size_t)i] == 0) prim.store(i); });
	primary = (uint64_t)prim.load();
	// row i of the matrix gives BWT symbol k = i - (i > primary); one output word (16 symbols) per iteration: no word is shared
	parallel_ranges((S + 15) / 16, nt, [&](int, int64_t a, int64_t b) {
		for (int64_t kw = a; kw < b; kw++) {
			uint32_t v = 0; const int64_t k1 = kw * 16 + 16 < S ? kw * 16 + 16 : S;
			for (int64_t k = kw * 16; k < k1; k++) {
				const int64_t i = k + (k >= (int64_t)primary ? 1 : 0);
				v |= (uint32_t)(T[(size_t)SA[(size_t)i] - 1] - 1) << ((~k & 15) << 1);
			}
			packed[(size_t)kw] = v;
		}
	});
	parallel_ranges((int64_t)sa.size(), nt, [&](int, int64_t a, int64_t b) { for (int64_t i = a < 1 ? 1 : a; i < b; i++) sa[(size_t)i] = (uint64_t)SA[(size_t)(32 * i)]; });
}

struct FaRec { std::string name, comment, seq; };

// kseq-style FASTA parsing (BWT_Index/kseq.h:176-215): name up to the first white space,
// the rest of the header line is the comment; plain or gzip'd input.
bool read_fasta_gz(const std::string &path, std::vector<FaRec> &recs)
{
	gzFile fp = gzopen(path.c_str(), "r");
	if (!fp) return false;
	std::string data; char buf[1 << 16]; int n;
	while ((n = gzread(fp, buf, sizeof(buf))) > 0) data.append(buf, (size_t)n);
	gzclose(fp);
	size_t p = 0, N = data.size();
	while (p < N && data[p] != '>' && data[p] != '@') p++;
	while (p < N) {
		p++;                                   // skip '>'
		FaRec r;
		size_t e = p; while (e < N && !isspace((unsigned char)data[e])) e++;
		r.name = data.substr(p, e - p);
		if (e < N && data[e] != '\n') { size_t le = data.find('\n', e + 1); if (le == std::string::npos) le = N; r.comment = data.substr(e + 1, le - e - 1); e = le; }
		while (!r.comment.empty() && r.comment.back() == '\r') r.comment.pop_back();
		p = e < N ? e + 1 : N;
		while (p < N && data[p] != '>') {
			size_t le = data.find('\n', p); if (le == std::string::npos) le = N;
			size_t ce = le; while (ce > p && data[ce - 1] == '\r') ce--;
			r.seq.append(data, p, ce - p);
			p = le < N ? le + 1 : N;
		}
		recs.push_back(r);
	}
	return true;
}

void fput(FILE *fp, const void *p, size_t n) { if (n) fwrite(p, 1, n, fp); }

} // namespace

void HostIndex::fill_view(gsa_index_view *v) const
{
	v->primary = primary; for (int i = 0; i < 5; i++) v->L2[i] = L2[i];
	v->bwt = bwt.data(); v->bwt_words = bwt.size(); v->sa = sa.data(); v->n_sa = sa.size();
	v->ref = ref.data(); v->G = G; v->chr_len = chr_len.data(); v->n_chr = (int32_t)chr_len.size();
}

void HostIndex::coordinate(int64_t rpos, int *bdir, int *chr, int *gpos) const
{
	size_t k = std::lower_bound(end_key.begin(), end_key.end(), rpos) - end_key.begin();
	*chr = end_chr[k];
	if (rpos < G) { *bdir = 1; *gpos = (int)(rpos + 1 - chr_fwd[*chr]); }
	else { *bdir = 0; *gpos = (int)(end_key[k] - rpos + 1); }
}

bool gsah_index_files_exist(const std::string &prefix)
{
	for (const char *ext : { ".ann", ".amb", ".pac" }) { FILE *fp = fopen((prefix + ext).c_str(), "r"); if (!fp) return false; fclose(fp); }
	return true;
}

// bytes [off, off + n) of a file into dst, in slices read by the pool's threads (page cache -> memory at the speed of many memcpys)
static bool pread_all(const std::string &path, size_t off, void *dst, size_t n)
{
	const int fd = open(path.c_str(), O_RDONLY);
	if (fd < 0) return false;
	std::atomic<bool> ok(true);
	par_ranges(n, (size_t)16 << 20, [&](size_t b, size_t e) {
		while (b < e) { const ssize_t r = pread(fd, (char *)dst + b, e - b, (off_t)(off + b)); if (r <= 0) { ok = false; return; } b += (size_t)r; }
	});
	close(fd);
	return ok;
}
static int64_t file_size(const std::string &path) { struct stat st; return stat(path.c_str(), &st) == 0 ? (int64_t)st.st_size : -1; }

// The index files as they lie on disk: .bwt, .sa, .ann and the RAW .pac bytes (idx.pac) -- everything gsa_create needs (with GSA_CREATE_REF_PAC the device unpacks the
// text itself), so a host program can start the device-side table builds and unpack RefSequence for its own emitters (gsah_unpack_ref) beside them.
bool gsah_load_index_files(const std::string &prefix, HostIndex &idx, std::string &err)
{
	// (round 5: every file goes straight into the vector that keeps it -- the first version read it into a scratch buffer and copied -- in
	//  slices read by the pool's threads, and RestoreReferenceInfo's unpacking loop runs on the pool as well: 12 s -> ~2 s for a 3.08 Gbp index)
	const int64_t bwt_sz = file_size(prefix + ".bwt");
	uint64_t hdr[5];
	if (bwt_sz < 40 || !pread_all(prefix + ".bwt", 0, hdr, 40)) { err = "cannot read " + prefix + ".bwt"; return false; }
	idx.primary = hdr[0]; idx.L2[0] = 0; memcpy(&idx.L2[1], &hdr[1], 32);
	idx.bwt.resize((size_t)(bwt_sz - 40) / 4);
	if (!pread_all(prefix + ".bwt", 40, idx.bwt.data(), idx.bwt.size() * 4)) { err = "cannot read " + prefix + ".bwt"; return false; }
	const uint64_t seq_len = idx.L2[4];
	const int64_t sa_sz = file_size(prefix + ".sa");
	uint64_t sa_hdr[7];
	if (sa_sz < 56 || !pread_all(prefix + ".sa", 0, sa_hdr, 56)) { err = "cannot read " + prefix + ".sa"; return false; }
	if ((sa_hdr[5] & 0xffffffffu) != 32) { err = "unexpected SA interval"; return false; }
	const uint64_t n_sa = (seq_len + 32) / 32;
	if ((uint64_t)sa_sz < 56 + (n_sa - 1) * 8) { err = prefix + ".sa is truncated"; return false; }
	idx.sa.resize(n_sa); idx.sa[0] = (uint64_t)-1;
	if (!pread_all(prefix + ".sa", 56, idx.sa.data() + 1, (n_sa - 1) * 8)) { err = "cannot read " + prefix + ".sa"; return false; }
	FILE *fp = fopen((prefix + ".ann").c_str(), "r");
	if (!fp) { err = "cannot read " + prefix + ".ann"; return false; }
	long long G; int n_seqs; unsigned seed;
	if (fscanf(fp, "%lld%d%u", &G, &n_seqs, &seed) != 3) { fclose(fp); err = "bad .ann"; return false; }
	idx.G = G; idx.chr_name.clear(); idx.chr_len.clear();
	char str[10240];
	for (int i = 0; i < n_seqs; i++) {
		unsigned gi; long long off; int len, nambs, ch;
		if (fscanf(fp, "%u%10239s", &gi, str) != 2) { fclose(fp); err = "bad .ann"; return false; }
		idx.chr_name.push_back(str);
		while ((ch = fgetc(fp)) != '\n' && ch != EOF);
		if (fscanf(fp, "%lld%d%d", &off, &len, &nambs) != 3) { fclose(fp); err = "bad .ann"; return false; }
		idx.chr_len.push_back(len);
	}
	fclose(fp);
	if (seq_len != (uint64_t)(2 * idx.G)) { err = "index is not forward+reverse"; return false; }
	{
		const int64_t pac_sz = file_size(prefix + ".pac");
		if (pac_sz < idx.G / 4 + (idx.G % 4 ? 1 : 0)) { err = "cannot read " + prefix + ".pac"; return false; }
		idx.pac.resize((size_t)pac_sz);
		if (!idx.pac.data() || !pread_all(prefix + ".pac", 0, idx.pac.data(), idx.pac.size())) { err = "cannot read " + prefix + ".pac"; return false; }
	}
	if (!idx.bwt.data() || !idx.sa.data()) { err = "out of memory"; return false; }
	const int64_t G2 = 2 * idx.G;
	idx.chr_fwd.clear(); idx.chr_rev.clear(); idx.end_key.clear(); idx.end_chr.clear();
	int64_t tot = 0; std::vector<std::pair<int64_t, int32_t> > ends;
	for (int i = 0; i < n_seqs; i++) {
		idx.chr_fwd.push_back(tot); tot += idx.chr_len[i]; idx.chr_rev.push_back(G2 - tot);
		ends.push_back(std::make_pair(idx.chr_fwd[i] + idx.chr_len[i] - 1, i)); ends.push_back(std::make_pair(idx.chr_rev[i] + idx.chr_len[i] - 1, i));
	}
	std::sort(ends.begin(), ends.end());
	for (size_t i = 0; i < ends.size(); i++) { idx.end_key.push_back(ends[i].first); idx.end_chr.push_back(ends[i].second); }
	return true;
}

// RestoreReferenceInfo (bwt_index.cpp:229-264): forward strand, then its reverse complement, from idx.pac; idx.pac is released afterwards unless keep_pac
bool gsah_unpack_ref(HostIndex &idx, std::string &err, bool keep_pac)
{
	idx.ref.resize((size_t)(2 * idx.G));
	if (!idx.ref.data() || !idx.pac.data()) { err = "out of memory"; return false; }
	const int64_t G2 = 2 * idx.G;
	{
		char *ref = idx.ref.data(); const uint8_t *pac = idx.pac.data();
		par_ranges((size_t)((idx.G + 3) / 4), (size_t)1 << 20, [&](size_t b4, size_t e4) {
			const int64_t fe = std::min<int64_t>((int64_t)e4 * 4, idx.G);
			for (int64_t f = (int64_t)b4 * 4; f < fe; f++) {
				const int b = pac[f >> 2] >> ((~f & 3) << 1) & 3;
				ref[f] = "ACGT"[b]; ref[G2 - 1 - f] = "TGCA"[b];
			}
		});
	}
	if (!keep_pac) idx.pac.resize(0);
	return true;
}

bool gsah_load_index(const std::string &prefix, HostIndex &idx, std::string &err)
{
	return gsah_load_index_files(prefix, idx, err) && gsah_unpack_ref(idx, err, false);
}


bool gsah_build_index(const std::string &fasta, const std::string &prefix, std::string &err)
{
	std::vector<FaRec> recs;
	if (!read_fasta_gz(fasta, recs) || recs.empty()) { err = "cannot read FASTA " + fasta; return false; }
	// ---- pack: bns_fasta2bntseq / add1 (bntseq.c:110-211) ----
	struct Hole { int64_t off; int32_t len; char amb; };
	std::vector<Hole> holes; std::vector<int64_t> offs; std::vector<int32_t> nambs;
	std::vector<uint8_t> codes;
	srand48(11);
	for (size_t s = 0; s < recs.size(); s++) {
		offs.push_back((int64_t)codes.size()); nambs.push_back(0);
		int lasts = 0;
		for (size_t i = 0; i < recs[s].seq.size(); i++) {
			const unsigned char ch = (unsigned char)recs[s].seq[i];
			int c = nt4(ch);
			if (c >= 4) {
				if (lasts == ch) holes.back().len++;
				else { Hole h = { offs[s] + (int64_t)i, 1, (char)ch }; holes.push_back(h); nambs[s]++; }
				c = (int)(lrand48() & 3);
			}
			lasts = ch;
			codes.push_back((uint8_t)c);
		}
	}
	const int64_t G = (int64_t)codes.size();
	if (G <= 0) { err = "empty reference"; return false; }
	// .pac (forward only; bntseq.c:192-201)
	{
		std::vector<uint8_t> pac((size_t)((G >> 2) + ((G & 3) ? 1 : 0)), 0);
		for (int64_t l = 0; l < G; l++) pac[l >> 2] |= codes[l] << ((~l & 3) << 1);
		FILE *fp = fopen((prefix + ".pac").c_str(), "wb"); if (!fp) { err = "cannot write " + prefix + ".pac"; return false; }
		fput(fp, pac.data(), pac.size());
		uint8_t ct = 0; if (G % 4 == 0) fput(fp, &ct, 1);
		ct = (uint8_t)(G % 4); fput(fp, &ct, 1);
		fclose(fp);
	}
	// .ann / .amb (bns_dump, bntseq.c:59-88)
	{
		FILE *fp = fopen((prefix + ".ann").c_str(), "w"); if (!fp) { err = "cannot write .ann"; return false; }
		fprintf(fp, "%lld %d %u\n", (long long)G, (int)recs.size(), 11u);
		for (size_t s = 0; s < recs.size(); s++) {
			const std::string anno = recs[s].comment.empty() ? "(null)" : recs[s].comment;
			fprintf(fp, "%d %s", 0, recs[s].name.c_str());
			if (!anno.empty()) fprintf(fp, " %s\n", anno.c_str()); else fprintf(fp, "\n");
			fprintf(fp, "%lld %d %d\n", (long long)offs[s], (int)recs[s].seq.size(), nambs[s]);
		}
		fclose(fp);
		fp = fopen((prefix + ".amb").c_str(), "w"); if (!fp) { err = "cannot write .amb"; return false; }
		fprintf(fp, "%lld %d %u\n", (long long)G, (int)recs.size(), (unsigned)holes.size());
		for (size_t h = 0; h < holes.size(); h++) fprintf(fp, "%lld %d %c\n", (long long)holes[h].off, holes[h].len, holes[h].amb);
		fclose(fp);
	}
	// ---- text = forward + reverse complement, then '$' ----
	const int64_t S = 2 * G;
	std::vector<uint8_t> T((size_t)S + 1);
	const int nt = index_threads();
	parallel_ranges(G, nt, [&](int, int64_t a, int64_t b) { for (int64_t i = a; i < b; i++) { T[(size_t)i] = codes[(size_t)i] + 1; T[(size_t)(S - 1 - i)] = (3 - codes[(size_t)i]) + 1; } });
	T[S] = 0;
	{ std::vector<uint8_t>().swap(codes); }
	// ---- suffix array -> BWT without '$', primary, L2, SA samples (bwt_cal_sa, bwt.c:101-123) ----
	uint64_t primary = 0, L2[5] = {0, 0, 0, 0, 0};
	std::vector<uint32_t> packed((size_t)((S + 15) / 16), 0);
	const uint64_t n_sa = (uint64_t)(S + 32) / 32;
	std::vector<uint64_t> sa(n_sa);
	// (GSA_INDEX_64BIT=1 forces the wide suffix sorter on any input: how the tests reach it with a small fixture)
	const char *f64 = getenv("GSA_INDEX_64BIT");
	if (S + 1 < (1ll << 31) - 1 && !(f64 && *f64 && *f64 != '0')) derive_bwt_sa<int32_t>(T, S, packed, primary, sa);
	else derive_bwt_sa<int64_t>(T, S, packed, primary, sa);
	{
		std::vector<std::vector<uint64_t> > part((size_t)nt, std::vector<uint64_t>(5, 0));
		parallel_ranges(S, nt, [&](int t, int64_t a, int64_t b) { uint64_t c[5] = {0, 0, 0, 0, 0}; for (int64_t i = a; i < b; i++) c[T[(size_t)i]]++; for (int k = 0; k < 5; k++) part[(size_t)t][(size_t)k] = c[k]; });
		for (int t = 0; t < nt; t++) for (int k = 0; k < 5; k++) L2[k] += part[(size_t)t][(size_t)k];      // T[i] in 1..4 -> L2[1..4] counts
	}
	for (int c = 1; c < 5; c++) L2[c] += L2[c - 1];
	{ std::vector<uint8_t>().swap(T); }
	// ---- interleave Occ every 128 (bwt_bwtupdate_core, bwtindex.c:53-75) ----
	const uint64_t n_occ = (uint64_t)(S + 127) / 128 + 1;
	std::vector<uint32_t> bwt(packed.size() + n_occ * 8, 0);
	{
		// block g (128 symbols = 8 packed words) goes to words [16 g, 16 g + 16): 8 words of running counts, then its symbols;
		// the counts in front of a thread's first block come from a pass over the per-thread totals
		const int64_t n_blk = (S + 127) / 128;
		auto count_word = [](uint32_t w, int nsym, uint64_t *c) { for (int t = 0; t < nsym; t++) c[(w >> ((~t & 15) << 1)) & 3]++; };
		std::vector<std::vector<uint64_t> > tot((size_t)nt, std::vector<uint64_t>(4, 0));
		parallel_ranges(n_blk, nt, [&](int t, int64_t a, int64_t b) {
			uint64_t c[4] = {0, 0, 0, 0};
			for (int64_t i = a * 128; i < b * 128 && i < S; i += 16) count_word(packed[(size_t)(i >> 4)], (int)(S - i < 16 ? S - i : 16), c);
			for (int k = 0; k < 4; k++) tot[(size_t)t][(size_t)k] = c[k];
		});
		const int nt_eff = (nt <= 1 || n_blk < 65536) ? 1 : nt;
		std::vector<std::vector<uint64_t> > start((size_t)nt_eff + 1, std::vector<uint64_t>(4, 0));
		for (int t = 0; t < nt_eff; t++) for (int k = 0; k < 4; k++) start[(size_t)t + 1][(size_t)k] = start[(size_t)t][(size_t)k] + tot[(size_t)t][(size_t)k];
		parallel_ranges(n_blk, nt, [&](int t, int64_t a, int64_t b) {
			uint64_t c[4]; for (int k = 0; k < 4; k++) c[k] = start[(size_t)t][(size_t)k];
			for (int64_t g = a; g < b; g++) {
				size_t k = (size_t)g * 16;
				memcpy(&bwt[k], c, 32); k += 8;
				for (int64_t i = g * 128; i < g * 128 + 128 && i < S; i += 16) { bwt[k++] = packed[(size_t)(i >> 4)]; count_word(packed[(size_t)(i >> 4)], (int)(S - i < 16 ? S - i : 16), c); }
			}
		});
		const size_t k_end = (size_t)n_blk * 8 + packed.size();
		memcpy(&bwt[k_end], start[(size_t)nt_eff].data(), 32);
		if (k_end + 8 != bwt.size()) { err = "internal: inconsistent bwt size"; return false; }
	}
	{
		FILE *fp = fopen((prefix + ".bwt").c_str(), "wb"); if (!fp) { err = "cannot write .bwt"; return false; }
		fput(fp, &primary, 8); fput(fp, &L2[1], 32); fput(fp, bwt.data(), bwt.size() * 4);
		fclose(fp);
	}
	// ---- SA sampled every 32 rows (bwt_dump_sa, bwt.c:185-196) ----
	{
		const uint64_t intv = 32, seq_len = (uint64_t)S;
		FILE *fp = fopen((prefix + ".sa").c_str(), "wb"); if (!fp) { err = "cannot write .sa"; return false; }
		fput(fp, &primary, 8); fput(fp, &L2[1], 32); fput(fp, &intv, 8); fput(fp, &seq_len, 8); fput(fp, sa.data() + 1, (n_sa - 1) * 8);
		fclose(fp);
	}
	return true;
}
