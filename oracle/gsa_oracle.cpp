// oracle/gsa_oracle.cpp -- TEST INFRASTRUCTURE ONLY (never part of the product).
//
// A plain, single-threaded CPU restatement of GSAlign's hot path:
// FM-index seed lookup -> seed grouping -> chaining -> block refinement ->
// gap filling -> gap closing (ksw2 global affine DP) -> identity filter.
// Every function cites the reference lines (relative to /root/reference) it
// follows.  It is written from the algorithm description in SURVEY.md
// Appendix A, with our own data layout (POD records, flat arrays), and is
// PINNED against the real reference: tests/test_oracle_vs_reference.py runs
// both this file and oracle/_ref/libgsref.so (the reference objects compiled
// in place) on the same inputs and demands identical seeds, groups and blocks
// after every one of the eight stages, and the committed fixtures under
// tests/golden/ hold the reference's outputs for the same comparison where
// /root/reference is absent.
//
// It is C++ rather than C for one reason: block order on score ties (and two
// more sorts with incomplete keys) is whatever libstdc++'s std::sort leaves
// (SURVEY.md App. B #10); using the same std::sort on the same permutation is
// the only way to restate that bit-exactly.
//
// Who may use this: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
// leg.  The product path (gsalign_amd/) must never link, load or call it.
#include "gsa_oracle.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

typedef uint64_t u64;
typedef int64_t i64;

// BWT_Index/bntseq.c:40-57 (ASCII -> 0..3, everything else 4)
struct Nt4 {
	unsigned char t[256];
	Nt4() { memset(t, 4, 256); t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3; }
};
const Nt4 NT4;
inline int nt4(char c) { return NT4.t[(unsigned char)c]; }

struct Frag {          // structure.h:103-113 without the two strings
	bool bSeed;
	int qPos, qLen, rLen;
	i64 rPos, PosDiff;
	std::string aln1, aln2;
};
struct Block {         // structure.h:115-122
	bool bDup = false;
	int score = 0, aln_len = 0;
	bool bDir = true; int gPos = 0, chr = 0;
	std::vector<Frag> f;
};

} // namespace

struct ora_ctx {
	// ---- index (structure.h:28-38) ----
	u64 primary, L2[5], seq_len;
	std::vector<uint32_t> bwt;
	std::vector<u64> sa;
	i64 G, G2;
	std::string ref;                         // 2G ASCII
	std::vector<int> chr_len;
	std::vector<i64> chr_fwd, chr_rev;       // bwt_index.cpp:247-248
	std::map<i64, int> ChrLocMap;            // bwt_index.cpp:251-252
	// ---- tunables ----
	int MinSeedLength = 15, MaxIndelSize = 25, MinAlnBlockScore = 200, MinAlnLength = 200, MinSeqIdy = 70;
	bool bSensitive = false, OneOnOne = false;
	// ---- per-contig state ----
	std::string q;
	int stage = 0;
	std::vector<Frag> SeedVec;
	std::vector<std::pair<int, int> > Groups;
	std::vector<Block> Blocks;
	std::vector<i64> RefChrScore;
	u64 cnt[8];
};

namespace {

// ---------------------------------------------------------------------------
// a1/a2  Occ primitives.  bwt_search.cpp:28-119.
// One 64-byte block = 16 x u32: words 0-7 hold four u64 running counts (A,C,G,T
// before the block), words 8-15 hold 128 symbols, 2 bit each, MSB first.
// We count with plain popcounts instead of the reference's byte table; the
// result is a pure function of (data, k) so the two agree.
// ---------------------------------------------------------------------------
inline void count_prefix(uint32_t w, int nsym, u64 c[4])   // symbols 0..nsym-1 of one word
{
	uint32_t m = (uint32_t)(0x55555555ull & ~((1ull << (32 - 2 * nsym)) - 1));
	uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
	c[0] += __builtin_popcount(~hi & ~lo & m);
	c[1] += __builtin_popcount(~hi & lo & m);
	c[2] += __builtin_popcount(hi & ~lo & m);
	c[3] += __builtin_popcount(hi & lo & m);
}

// bwt_occ4 (bwt_search.cpp:69-86)
void occ4(const ora_ctx *x, u64 k, u64 c[4])
{
	if (k == (u64)-1) { c[0] = c[1] = c[2] = c[3] = 0; return; }
	k -= (k >= x->primary);
	const uint32_t *p = &x->bwt[(k >> 7) << 4];
	memcpy(c, p, 32);
	p += 8;
	int n = (int)(k & 127) + 1;              // symbols of this block up to row k inclusive
	for (; n >= 16; n -= 16, ++p) count_prefix(*p, 16, c);
	if (n) count_prefix(*p, n, c);
}

// bwt_2occ4 (bwt_search.cpp:88-119): same values as two occ4 calls; the shared
// block walk only matters for the traffic counter.
void occ4x2(ora_ctx *x, u64 k, u64 l, u64 ck[4], u64 cl[4])
{
	u64 _k = k - (k >= x->primary), _l = l - (l >= x->primary);
	bool one_block = !((_l >> 7) != (_k >> 7) || k == (u64)-1 || l == (u64)-1);
	if (one_block) x->cnt[0] += 1;
	else x->cnt[0] += (k != (u64)-1) + (l != (u64)-1);
	occ4(x, k, ck); occ4(x, l, cl);
}

// bwt_occ (bwt_search.cpp:45-67)
u64 occ1(const ora_ctx *x, u64 k, int c)
{
	if (k == x->seq_len) return x->L2[c + 1] - x->L2[c];
	if (k == (u64)-1) return 0;
	u64 cc[4]; occ4(x, k, cc);
	return cc[c];
}

// bwt_invPsi (bwt_search.cpp:121-127).  Note the symbol fetch uses k-(k>primary)
// while Occ uses k-(k>=primary); they differ only at k==primary, which maps to 0.
u64 inv_psi(const ora_ctx *x, u64 k)
{
	u64 r = k - (k > x->primary);
	int sym = (x->bwt[((r >> 7) << 4) + 8 + ((r & 127) >> 4)] >> ((~r & 15) << 1)) & 3;
	u64 v = x->L2[sym] + occ1(x, k, sym);
	return k == x->primary ? 0 : v;
}

// bwt_sa (bwt_search.cpp:129-139); sa_intv = 32
u64 locate(ora_ctx *x, u64 k)
{
	u64 steps = 0;
	while (k & 31) { ++steps; k = inv_psi(x, k); x->cnt[1]++; }
	x->cnt[2]++;
	return steps + x->sa[k >> 5];
}

// BWT_Search (bwt_search.cpp:141-185).  Returns freq (0 = rejected), sets len.
int bwt_search(ora_ctx *x, int start, int stop, int *len, u64 locs[100])
{
	const std::string &s = x->q;
	int p = nt4(s[start]);
	u64 x0 = x->L2[p] + 1, x1 = x->L2[3 - p] + 1, x2 = x->L2[p + 1] - x->L2[p];
	int pos;
	for (pos = start + 1; pos < stop; pos++) {
		int nt = nt4(s[pos]);
		if (nt > 3) break;
		u64 tk[4], tl[4], o1[4], o2[4], o0[4];
		occ4x2(x, x1 - 1, x1 - 1 + x2, tk, tl);
		for (int i = 0; i < 4; i++) { o1[i] = x->L2[i] + 1 + tk[i]; o2[i] = tl[i] - tk[i]; }
		o0[3] = x0 + (x1 <= x->primary && x1 + x2 - 1 >= x->primary);
		o0[2] = o0[3] + o2[3]; o0[1] = o0[2] + o2[2]; o0[0] = o0[1] + o2[1];
		int i = 3 - nt;
		if (o2[i] == 0) break;
		x0 = o0[i]; x1 = o1[i]; x2 = o2[i];
	}
	*len = pos - start;
	if (*len < x->MinSeedLength) return 0;
	int freq = (int)x2;
	if (freq > 100) return 0;                 // MaxSeedFreq, bwt_search.cpp:3,177
	for (int i = 0; i < freq; i++) locs[i] = locate(x, x0 + i);
	return freq;
}

// CompByPosDiff / CompByQueryPos / CompByRemoval (ProcessCandidateAlignment.cpp:3-19)
bool by_posdiff(const Frag &a, const Frag &b) { return a.PosDiff == b.PosDiff ? a.qPos < b.qPos : a.PosDiff < b.PosDiff; }
bool by_qpos(const Frag &a, const Frag &b) { return a.qPos == b.qPos ? a.rPos < b.rPos : a.qPos < b.qPos; }
bool by_removal(const Frag &a, const Frag &b) { return (a.bSeed && b.bSeed) ? a.qPos < b.qPos : (a.bSeed > b.bSeed); }
bool by_score(const Block &a, const Block &b) { return a.score > b.score; }   // :21-24

// IdentifyLocalMEM (GSAlign.cpp:51-107), one thread; chunk = 10000 (GSAlign.cpp:5)
void stage1_seeds(ora_ctx *x)
{
	const int L = (int)x->q.size();
	x->SeedVec.clear();
	u64 locs[100];
	for (uint32_t cs = 0; cs < (uint32_t)L; cs += 10000) {
		uint32_t start = cs, stop = cs + 10000; if (stop > (uint32_t)L) stop = L;
		while (start < stop) {
			if (nt4(x->q[start]) > 3) { start++; continue; }
			int len, freq = bwt_search(x, (int)start, (int)stop, &len, locs);
			if (freq > 0) {
				Frag s; s.bSeed = true; s.qPos = (int)start; s.qLen = s.rLen = len;
				for (int i = 0; i < freq; i++) { s.rPos = (i64)locs[i]; s.PosDiff = s.rPos - s.qPos; x->SeedVec.push_back(s); x->cnt[3]++; }
				start += x->bSensitive ? 5 : (uint32_t)(len + 1);
			} else start++;
		}
	}
	std::sort(x->SeedVec.begin(), x->SeedVec.end(), by_posdiff);
	// SeedGrouping (GSAlign.cpp:126-143).  The reference pushes a bogus group
	// (0,1) when there are no seeds at all and then reads SeedVec[0] of an empty
	// vector; we define that case as "no groups".
	x->Groups.clear();
	int n = (int)x->SeedVec.size(), p = 0, i = 0, j = 1;
	if (n == 0) return;
	for (; j < n; i++, j++)
		if (x->SeedVec[j].PosDiff - x->SeedVec[i].PosDiff > x->MaxIndelSize) { x->Groups.push_back(std::make_pair(p, j)); p = j; }
	if (p < j) x->Groups.push_back(std::make_pair(p, j));
}

// AddAlnBlock (GSAlign.cpp:29-49)
void add_block(ora_ctx *x, int i, int j)
{
	Block b;
	b.f.assign(x->SeedVec.begin() + i, x->SeedVec.begin() + j);
	for (size_t k = 0; k < b.f.size(); k++) b.score += b.f[k].qLen;
	int region = (b.f.back().qPos + b.f.back().qLen) - b.f.front().qPos;
	if (b.score < x->MinAlnBlockScore || region < x->MinAlnLength || (b.score < 1000 && b.score < region * 0.05)) return;
	x->Blocks.push_back(b);
}

// RemoveOutlierSeeds + RefinePDFmap + Check_PD_Frequency (GSAlign.cpp:145-153,245-296)
void remove_outliers(ora_ctx *x, int Beg, int End, const std::vector<char> &uniq, int ubase)
{
	std::vector<Frag> &S = x->SeedVec;
	std::map<int, int> pdf;
	for (int i = Beg; i < End; i++) if (uniq[i - ubase]) pdf[(int)(S[i].PosDiff >> 4)]++;
	std::pair<int, int> best(0, 0);
	for (std::map<int, int>::iterator it = pdf.begin(); it != pdf.end(); ++it) if (it->second > best.second) best = *it;
	for (std::map<int, int>::iterator it = pdf.begin(); it != pdf.end(); ++it) if (std::abs(it->first - best.first) >= 3) it->second = 0;
	i64 sum = 0; int n = 0;
	for (int i = Beg; i < End; i++) if (uniq[i - ubase] && pdf[(int)(S[i].PosDiff >> 4)] > 0) { sum += S[i].PosDiff; n++; }
	i64 avg = n > 0 ? sum / n : x->G;
	for (int i = Beg; i < End; i++) if (uniq[i - ubase]) {
		int pd = (int)(S[i].PosDiff >> 4);
		if (std::llabs(avg - S[i].PosDiff) > x->MaxIndelSize && !(pdf[pd] >= 3)) S[i].bSeed = false;
	}
}

// SeedGroupAnalysis (GSAlign.cpp:305-375)
void group_analysis(ora_ctx *x, int Beg, int End)
{
	std::vector<Frag> &S = x->SeedVec;
	std::sort(S.begin() + Beg, S.begin() + End, by_qpos);
	std::vector<char> uniq(End - Beg, 0);
	int i, j, k, n, p;
	for (i = Beg, j = i + 1; i < End; i++, j++) {                       // :316-325
		if (j < End && S[i].qPos == S[j].qPos) { while (++j < End && S[i].qPos == S[j].qPos); i = j - 1; }
		else uniq[i - Beg] = 1;
	}
	for (n = uniq[0] ? 1 : 0, i = Beg, j = Beg + 1; j < End; j++) {     // :326-337
		if (uniq[j - Beg]) {
			if (S[j].PosDiff == S[j - 1].PosDiff) n++;
			else if (++n >= 30 && S[j].qPos - S[i].qPos > 3000) { remove_outliers(x, i, j, uniq, Beg); i = j; n = 0; }
		}
	}
	remove_outliers(x, i, End, uniq, Beg);                               // :338
	for (i = Beg, j = i + 1; i < End; i++, j++) {                       // :341-350
		if (j < End && S[i].qPos == S[j].qPos) {
			while (++j < End && S[i].qPos == S[j].qPos);
			// FindNeighboringPosDiffAvg (:178-206)
			i64 s1 = 0, s2 = 0; int n1 = 0, n2 = 0, p1, p2;
			for (p1 = i - 1; p1 >= Beg; p1--) if (uniq[p1 - Beg] && S[p1].bSeed) { n1++; s1 += S[p1].PosDiff; if (n1 == 5) break; }
			for (p2 = j; p2 < End && p2 > Beg; p2++) if (uniq[p2 - Beg] && S[p2].bSeed) { n2++; s2 += S[p2].PosDiff; if (n2 == 5) break; }
			i64 avg = (n1 > 0 || n2 > 0) ? (s1 + s2) / (n1 + n2) : S[i].PosDiff;
			// RemoveRedundantSeeds (:208-225)
			int idx = -1; i64 diff, md = x->G;
			for (k = i; k < j; k++) if ((diff = std::llabs(S[k].PosDiff - avg)) < x->MaxIndelSize && diff < md) { md = diff; idx = k; }
			for (k = i; k < j; k++) if (k != idx) S[k].bSeed = false;
			i = j - 1;
		}
	}
	// :353 -- the reference has no lower bound on this trim (App. B #6); a group
	// whose seeds all died makes it run off the group and then sort a negative
	// range (it crashes).  We define that as "no blocks" and count it.
	std::sort(S.begin() + Beg, S.begin() + End, by_removal);
	while (End > Beg && !S[End - 1].bSeed) End--;
	if (End == Beg) { x->cnt[7]++; return; }
	for (i = Beg, j = i + 1, k = j + 1; k < End; i++, j++, k++)          // :355-362
		if (std::llabs(S[j].PosDiff - S[i].PosDiff) > 5 && std::llabs(S[j].PosDiff - S[k].PosDiff) > 5) S[j].bSeed = false;
	std::sort(S.begin() + Beg, S.begin() + End, by_removal);
	while (End > Beg && !S[End - 1].bSeed) End--;
	if (End == Beg) { x->cnt[7]++; return; }
	for (p = i = Beg, j = i + 1; j < End; i++, j++)                      // :364-374
		if (S[j].qPos - S[i].qPos - S[i].qLen > 5000 || std::llabs(S[i].PosDiff - S[j].PosDiff) > 100) { add_block(x, p, j); p = j; }
	add_block(x, p, j);
}

// GenerateAlignmentBlocks (GSAlign.cpp:377-391)
void stage2_blocks(ora_ctx *x)
{
	x->Blocks.clear();
	for (size_t g = 0; g < x->Groups.size(); g++) {
		int b = x->Groups[g].first, e = x->Groups[g].second, score = 0;
		for (int i = b; i < e; i++) score += x->SeedVec[i].qLen;
		if (score < x->MinAlnBlockScore) continue;
		if (b < e) group_analysis(x, b, e);
	}
}

// RemoveOverlaps + RemoveBadSeeds (ProcessCandidateAlignment.cpp:63-70,189-231)
void remove_overlaps(std::vector<Frag> &v)
{
	while (true) {
		bool mod = false; int num = (int)v.size();
		for (int i = 0, j = 1; j < num; i++, j++) {
			int ov;
			if (v[j].rPos <= v[i].rPos) { mod = true; v[i].bSeed = false; continue; }
			if ((ov = (int)(v[i].rPos + v[i].rLen - v[j].rPos)) > 0) {
				v[i].qLen -= ov; v[i].rLen -= ov;
				if (v[i].qLen <= 0 || v[i].rLen <= 0) { mod = true; v[i].bSeed = false; continue; }
			}
			if ((ov = v[i].qPos + v[i].qLen - v[j].qPos) > 0) {
				v[i].qLen -= ov; v[i].rLen -= ov;
				if (v[i].qLen <= 0 || v[i].rLen <= 0) { mod = true; v[i].bSeed = false; continue; }
			}
		}
		if (!mod) break;
		std::sort(v.begin(), v.end(), by_removal);
		int n = (int)v.size(); while (n > 0 && !v[n - 1].bSeed) n--;
		v.resize(n);
	}
}

// CalAlnBlockScore (ProcessCandidateAlignment.cpp:26-36)
int block_score(const ora_ctx *x, const std::vector<Frag> &v)
{
	if (v.empty()) return 0;
	if (v.back().qPos + v.back().qLen - v.front().qPos < x->MinAlnLength) return 0;
	int s = 0; for (size_t i = 0; i < v.size(); i++) s += v[i].qLen;
	return s;
}

// RemoveBadAlnBlocks (ProcessCandidateAlignment.cpp:72-79)
void remove_bad_blocks(ora_ctx *x)
{
	std::sort(x->Blocks.begin(), x->Blocks.end(), by_score);
	size_t n = x->Blocks.size(); while (n > 0 && x->Blocks[n - 1].score == 0) n--;
	x->Blocks.resize(n);
}

// CreateKmerID / CreateKmerVecFromReadSeq (KmerAnalysis.cpp:10-17,32-76), k = 5.
// The N handling is reproduced as written (App. B #7).
std::vector<uint32_t> kmer_vec(int len, const char *seq)
{
	std::vector<uint32_t> vec;
	uint32_t wid, count = 0, head = 0, tail = 0;
	while (count < 5 && tail < (uint32_t)len) { if (seq[tail++] != 'N') count++; else count = 0; }
	if (count == 5) {
		struct L { static uint32_t id(const char *s, short pos) { uint32_t v = 0; for (uint32_t i = pos, e = pos + 5; i < e; i++) v = (v << 2) + nt4(s[i]); return v; } };
		wid = L::id(seq, (short)head); vec.push_back(wid);
		for (head += 1; tail < (uint32_t)len; head++, tail++) {
			if (seq[tail] != 'N') { wid = ((wid & 0xFF) << 2) + nt4(seq[tail]); vec.push_back(wid); }
			else {
				count = 0; tail++;
				while (count < 5 && tail < (uint32_t)len) { if (seq[tail++] != 'N') count++; else count = 0; }
				if (count == 5) { wid = L::id(seq, (short)head); vec.push_back(wid); }
				else break;
			}
		}
		std::sort(vec.begin(), vec.end());
	}
	return vec;
}

// CalGapSimilarity (KmerAnalysis.cpp:78-121)
bool gap_similar(const ora_ctx *x, int q1, int q2, i64 r1, i64 r2)
{
	bool sim = false;
	int q_len = q2 - q1, r_len = (int)(r2 - r1);
	if (r1 - q1 == r2 - q2) {
		int idy = 0; i64 r = r1;
		for (int q = q1; q < q2; q++, r++) { int a = nt4(x->ref[r]), b = nt4(x->q[q]); if (a == b || a == 4 || b == 4) idy++; }
		if (idy >= q_len * 0.5) sim = true;
	}
	if (!sim && q_len <= 5000 && r_len <= 5000) {
		std::string qf = x->q.substr(q1, q_len), rf = x->ref.substr(r1, r_len);
		std::vector<uint32_t> a = kmer_vec(q_len, qf.c_str()), b = kmer_vec(r_len, rf.c_str()), c;
		std::set_intersection(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(c));
		if ((int)c.size() > (q_len + r_len) * 0.1) sim = true;
	}
	return sim;
}

// shared tail of CheckGapsBetweenSeeds / CheckAlnBlockSpanMultipleRefChrs
// (ProcessCandidateAlignment.cpp:101-117,140-155): zero the parent, append the
// pieces that pass the strict '>' test at the END of the block list.
// NOTE: the reference keeps a reference into AlnBlockVec across push_back; if
// the vector reallocates in the middle of one split it reads freed memory and
// crashes, so a valid reference run never depends on that.
void split_block(ora_ctx *x, size_t bi, const std::vector<int> &cuts)
{
	if (cuts.empty()) return;
	std::vector<Frag> parent = x->Blocks[bi].f;
	x->Blocks[bi].score = 0;
	int i = 0;
	for (size_t c = 0; c <= cuts.size(); c++) {
		int j = c < cuts.size() ? cuts[c] : (int)parent.size();
		Block sub; sub.f.assign(parent.begin() + i, parent.begin() + j);
		if ((sub.score = block_score(x, sub.f)) > x->MinAlnBlockScore) x->Blocks.push_back(sub);
		i = j;
	}
}

// CheckGapsBetweenSeeds (ProcessCandidateAlignment.cpp:120-156)
void stage4_gaps(ora_ctx *x)
{
	size_t nb = x->Blocks.size();
	for (size_t b = 0; b < nb; b++) {
		std::vector<int> cuts;
		{
			const std::vector<Frag> &v = x->Blocks[b].f;
			int num = (int)v.size();
			for (int i = 0, j = 1; j < num; i++, j++) {
				int qGap = v[j].qPos - v[i].qPos - v[i].qLen;
				int rGap = (int)(v[j].rPos - v[i].rPos - v[i].rLen);
				if (qGap > 300 || rGap > 300)
					if (qGap > 5000 || rGap > 5000 || !gap_similar(x, v[i].qPos + v[i].qLen, v[j].qPos, v[i].rPos + v[i].rLen, v[j].rPos)) cuts.push_back(j);
			}
		}
		split_block(x, b, cuts);
	}
	remove_bad_blocks(x);
}

// CheckAlnBlockSpanMultipleRefChrs (ProcessCandidateAlignment.cpp:81-118)
void stage5_chrs(ora_ctx *x)
{
	size_t nb = x->Blocks.size();
	for (size_t b = 0; b < nb; b++) {
		std::vector<int> cuts;
		{
			const std::vector<Frag> &v = x->Blocks[b].f;
			int num = (int)v.size(); i64 last = -1;
			for (int i = 0, j = 1; j < num; j++) {
				if (last == -1) last = x->ChrLocMap.lower_bound(v[i].rPos)->first;
				if (v[j].rPos > last) { cuts.push_back(j); i = j; last = x->ChrLocMap.lower_bound(v[i].rPos)->first; }
			}
		}
		split_block(x, b, cuts);
	}
	remove_bad_blocks(x);
}

bool by_block_qpos(const Block &a, const Block &b) { return a.f.front().qPos == b.f.front().qPos ? a.score > b.score : a.f.front().qPos < b.f.front().qPos; } // GSAlign.cpp:17-21
bool by_block_rpos(const Block &a, const Block &b) { return a.f.front().rPos == b.f.front().rPos ? a.score > b.score : a.f.front().rPos < b.f.front().rPos; } // :23-27

// RemoveRedundantAlnBlocks (GSAlign.cpp:415-471), CheckDuplicatedChrScore (:409-413),
// ReverseRefCoordinate (tools.cpp:305-312)
void remove_redundant(ora_ctx *x, int type)
{
	std::vector<Block> &B = x->Blocks;
	int nb = (int)B.size();
	if (type == 1) std::sort(B.begin(), B.end(), by_block_qpos); else std::sort(B.begin(), B.end(), by_block_rpos);
	struct L {
		static bool dupchr(int s1, int s2) { return s1 > s2 && s1 >= s2 * 2; }
		static void rev(const ora_ctx *x, i64 &a, i64 &b) { i64 t = a; a = x->G2 - 1 - b; b = x->G2 - 1 - t; }
	};
	for (int i = 0; i < nb; i++) {
		if (B[i].score == 0) continue;
		i64 h1 = type == 1 ? B[i].f.front().qPos : B[i].f.front().rPos;
		i64 t1 = type == 1 ? B[i].f.back().qPos + B[i].f.back().qLen - 1 : B[i].f.back().rPos + B[i].f.back().rLen - 1;
		int c1 = x->ChrLocMap.lower_bound(B[i].f.front().rPos)->second;
		if (type == 2 && h1 >= x->G) L::rev(x, h1, t1);
		for (int j = i + 1; j < nb; j++) {
			if (B[j].score == 0) continue;
			i64 h2 = type == 1 ? B[j].f.front().qPos : B[j].f.front().rPos;
			i64 t2 = type == 1 ? B[j].f.back().qPos + B[j].f.back().qLen - 1 : B[j].f.back().rPos + B[j].f.back().rLen - 1;
			if (type == 1 && h1 == h2 && t1 == t2) { B[i].bDup = true; B[j].score = 0; continue; }
			int c2 = x->ChrLocMap.lower_bound(B[j].f.front().rPos)->second;
			if (type == 2 && h2 >= x->G) L::rev(x, h2, t2);
			if (h2 < t1) {
				i64 ov = t2 > t1 ? t1 - h2 : t2 - h2;
				float f1 = 1. * ov / (t1 - h1), f2 = 1. * ov / (t2 - h2);
				if ((f1 > f2 && f1 >= 0.9) || (x->OneOnOne && L::dupchr((int)x->RefChrScore[c2], (int)x->RefChrScore[c1]))) { B[i].score = 0; break; }
				if ((f2 > f1 && f2 >= 0.9) || (x->OneOnOne && L::dupchr((int)x->RefChrScore[c1], (int)x->RefChrScore[c2]))) B[j].score = 0;
			} else break;
		}
	}
	remove_bad_blocks(x);
}

void stage6_redundant(ora_ctx *x)
{
	for (size_t i = 0; i < x->Blocks.size(); i++) x->Blocks[i].bDup = false;            // GSAlign.cpp:510
	x->RefChrScore.assign(x->chr_len.size(), 0);                                       // EstChromosomeSimilarity :393-407
	for (size_t i = 0; i < x->Blocks.size(); i++) x->RefChrScore[x->ChrLocMap.lower_bound(x->Blocks[i].f.front().rPos)->second] += x->Blocks[i].score;
	remove_redundant(x, 1); remove_redundant(x, 2);
}

// IdentifyNormalPairs / FillAlnBlockGaps (ProcessCandidateAlignment.cpp:241-276)
void stage7_fill(ora_ctx *x)
{
	for (size_t b = 0; b < x->Blocks.size(); b++) {
		if (!(x->Blocks[b].score > 0)) continue;
		std::vector<Frag> &v = x->Blocks[b].f;
		int num = (int)v.size();
		if (num == 1) continue;
		for (int i = 0, j = 1; j < num; i++, j++) {
			int qg = v[j].qPos - (v[i].qPos + v[i].qLen); if (qg < 0) qg = 0;
			int rg = (int)(v[j].rPos - (v[i].rPos + v[i].rLen)); if (rg < 0) rg = 0;
			if (qg > 0 || rg > 0) {
				Frag g; g.bSeed = false; g.qPos = v[i].qPos + v[i].qLen; g.rPos = v[i].rPos + v[i].rLen;
				g.PosDiff = g.rPos - g.qPos; g.qLen = qg; g.rLen = rg;
				v.push_back(g);
			}
		}
		if ((int)v.size() > num) std::inplace_merge(v.begin(), v.begin() + num, v.end(), by_qpos);
	}
}

// ---------------------------------------------------------------------------
// a13  ksw_extz2_sse + ksw_backtrack (ksw2_alignment.cpp:25-68,70-249), called
// with m=5, q=2, e=1, w=-1: global, full matrix, match +1, mismatch -1, N 0.
// Cell-exact scalar evaluation of the Suzuki-Kasahara difference recurrence
// (SURVEY.md App. A.6).  s1 = reference fragment (ksw "query", index j),
// s2 = query fragment (ksw "target", index i = t); anti-diagonal r = i + j.
// Returns the op string in REVERSE order, like the reference.
// ---------------------------------------------------------------------------
std::string ksw_ops_reversed(const std::string &s1, const std::string &s2)
{
	const int qlen = (int)s1.size(), tlen = (int)s2.size();
	std::string ops;
	if (qlen <= 0 || tlen <= 0) return ops;
	std::vector<uint8_t> qs(qlen), ts(tlen);
	for (int i = 0; i < qlen; i++) qs[i] = (uint8_t)nt4(s1[i]);
	for (int i = 0; i < tlen; i++) ts[i] = (uint8_t)nt4(s2[i]);
	std::vector<int8_t> u(tlen + 1, 0), v(tlen + 1, 0), xx(tlen + 1, 0), yy(tlen + 1, 0);
	const int nr = qlen + tlen - 1;
	std::vector<size_t> rowoff(nr + 1); std::vector<int> rowst(nr);
	{ size_t o = 0; for (int r = 0; r < nr; r++) { int st = std::max(0, r - qlen + 1), en = std::min(tlen - 1, r); rowoff[r] = o; rowst[r] = st; o += (size_t)(en - st + 1); } rowoff[nr] = o; }
	std::vector<uint8_t> dir(rowoff[nr]);
	const int8_t q = 2, qe2 = 6, maxsc = 7;
	for (int r = 0; r < nr; r++) {
		int st = std::max(0, r - qlen + 1), en = std::min(tlen - 1, r);
		int8_t x1, v1;                                   // (r-1, st-1)        ksw2_alignment.cpp:157-164
		if (st > 0) { x1 = xx[st - 1]; v1 = v[st - 1]; } else { x1 = 0; v1 = r ? q : 0; }
		if (en >= r) { yy[r] = 0; u[r] = r ? q : 0; }     // (r-1, t=r) boundary :165
		uint8_t *dr = &dir[rowoff[r]];
		for (int t = st; t <= en; t++) {
			int8_t sc; uint8_t a_ = ts[t], b_ = qs[r - t];
			sc = (a_ == 4 || b_ == 4) ? 0 : (a_ == b_ ? 1 : -1);
			int8_t z = sc + qe2;
			int8_t xt1 = x1, vt1 = v1; x1 = xx[t]; v1 = v[t];
			int8_t a = xt1 + vt1, ut = u[t], b = yy[t] + ut;
			uint8_t d = a > z ? 1 : 0; z = z > a ? z : a;   // signed
			if (b > z) d = 2;
			z = (uint8_t)z > (uint8_t)b ? z : b;            // unsigned max, both non-negative
			z = (uint8_t)z < (uint8_t)maxsc ? z : maxsc;
			u[t] = z - vt1; v[t] = z - ut;
			z -= q; a -= z; b -= z;
			xx[t] = a > 0 ? a : 0; if (a > 0) d |= 0x08;
			yy[t] = b > 0 ? b : 0; if (b > 0) d |= 0x10;
			dr[t - st] = d;
		}
	}
	// ksw_backtrack (:25-68); with the full band the force_state paths never fire
	int i = tlen - 1, j = qlen - 1, state = 0;
	while (i >= 0 && j >= 0) {
		int r = i + j; uint32_t tmp = dir[rowoff[r] + (i - rowst[r])];
		if (state == 0) state = tmp & 7;
		else if (!(tmp >> (state + 2) & 1)) state = 0;
		if (state == 0) state = tmp & 7;
		if (state == 0) { ops.push_back('M'); --i; --j; }
		else if (state == 1 || state == 3) { ops.push_back('D'); --i; }
		else { ops.push_back('I'); --j; }
	}
	if (i >= 0) ops.append(i + 1, 'D');
	if (j >= 0) ops.append(j + 1, 'I');
	return ops;
}

// ksw2_alignment (ksw2_alignment.cpp:251-273): insert '-' per op, last op first
void ksw2_align(std::string &s1, std::string &s2)
{
	std::string ops = ksw_ops_reversed(s1, s2);
	std::string a, b; a.reserve(ops.size()); b.reserve(ops.size());
	size_t i1 = 0, i2 = 0;
	for (int k = (int)ops.size() - 1; k >= 0; k--) {
		if (ops[k] == 'D') { a.push_back('-'); b.push_back(s2[i2++]); }
		else if (ops[k] == 'I') { a.push_back(s1[i1++]); b.push_back('-'); }
		else { a.push_back(s1[i1++]); b.push_back(s2[i2++]); }
	}
	s1.swap(a); s2.swap(b);
}

// GenCoordinateInfo (tools.cpp:120-140)
void gen_coor(const ora_ctx *x, i64 rPos, bool *dir, int *chr, int *gpos)
{
	std::map<i64, int>::const_iterator it = x->ChrLocMap.lower_bound(rPos);
	*chr = it->second;
	if (rPos < x->G) { *dir = true; *gpos = (int)(rPos + 1 - x->chr_fwd[it->second]); }
	else { *dir = false; *gpos = (int)(it->first - rPos + 1); }
}

// GenerateFragAlignment (ProcessCandidateAlignment.cpp:290-351), CheckFragPairMismatch
// (:49-61), CountIdenticalPairs (:38-47), identity filter (GSAlign.cpp:529-540)
void stage8_align(ora_ctx *x)
{
	for (size_t b = 0; b < x->Blocks.size(); b++) {
		Block &B = x->Blocks[b];
		uint32_t aln_len = 0, score = 0;
		for (size_t k = 0; k < B.f.size(); k++) {
			Frag &f = B.f[k];
			if (f.bSeed) { aln_len += f.qLen; score += f.qLen; continue; }
			if (f.qLen == 0) { aln_len += f.rLen; f.aln1 = x->ref.substr(f.rPos, f.rLen); f.aln2.assign(f.rLen, '-'); continue; }
			if (f.rLen == 0) { aln_len += f.qLen; f.aln1.assign(f.qLen, '-'); f.aln2 = x->q.substr(f.qPos, f.qLen); continue; }
			int mism = -1;
			if (f.qLen == f.rLen) {
				mism = 0;
				for (int i = 0; i < f.qLen; i++) { int a = nt4(x->q[f.qPos + i]); if (a == 4) continue; if (a != nt4(x->ref[f.rPos + i])) mism++; }
			}
			f.aln1 = x->ref.substr(f.rPos, f.rLen); f.aln2 = x->q.substr(f.qPos, f.qLen);
			if (f.qLen == f.rLen && mism <= 5) { aln_len += f.qLen; score += f.qLen - mism; continue; }
			x->cnt[4] += (u64)f.rLen * (u64)f.qLen; x->cnt[5]++; x->cnt[6] += (u64)f.rLen + f.qLen;
			ksw2_align(f.aln1, f.aln2);
			aln_len += (uint32_t)f.aln1.size();
			int n = 0; for (size_t i = 0; i < f.aln1.size(); i++) if (nt4(f.aln1[i]) == nt4(f.aln2[i])) n++;
			score += n;
		}
		B.aln_len = (int)aln_len; B.score = (int)score;
	}
	for (size_t b = 0; b < x->Blocks.size(); b++) {
		Block &B = x->Blocks[b];
		if ((int)(100 * (1.0 * B.score / B.aln_len)) < x->MinSeqIdy) B.score = 0;
		else gen_coor(x, B.f[0].rPos, &B.bDir, &B.chr, &B.gPos);
	}
	remove_bad_blocks(x);
}

} // namespace

extern "C" {

ora_ctx *ora_create(const uint64_t hdr[5], const uint32_t *bwt, uint64_t bwt_words, const uint64_t *sa, uint64_t n_sa,
                    const char *ref, int64_t G, const int32_t *chr_len, int n_chr)
{
	ora_ctx *x = new ora_ctx();
	x->primary = hdr[0]; x->L2[0] = 0; for (int i = 1; i < 5; i++) x->L2[i] = hdr[i];
	x->seq_len = x->L2[4];
	x->bwt.assign(bwt, bwt + bwt_words);
	x->sa.assign(sa, sa + n_sa);
	x->G = G; x->G2 = 2 * G;
	x->ref.assign(ref, (size_t)(2 * G));
	i64 tot = 0;
	for (int i = 0; i < n_chr; i++) {                                   // RestoreReferenceInfo, bwt_index.cpp:240-253
		x->chr_len.push_back(chr_len[i]); x->chr_fwd.push_back(tot); tot += chr_len[i]; x->chr_rev.push_back(x->G2 - tot);
		x->ChrLocMap[x->chr_fwd[i] + chr_len[i] - 1] = i; x->ChrLocMap[x->chr_rev[i] + chr_len[i] - 1] = i;
	}
	memset(x->cnt, 0, sizeof(x->cnt));
	return x;
}
void ora_destroy(ora_ctx *x) { delete x; }

void ora_params(ora_ctx *x, int slen, int ind, int clr, int alen, int idy, int sen, int one)
{
	x->MinSeedLength = slen; x->MaxIndelSize = ind; x->MinAlnBlockScore = clr; x->MinAlnLength = alen; x->MinSeqIdy = idy;
	x->bSensitive = sen != 0; x->OneOnOne = one != 0;
	if (x->bSensitive) x->MinSeedLength = 10;                            // main.cpp:323
}
void ora_set_query(ora_ctx *x, const char *seq, int len)
{
	x->q.assign(seq, len); x->stage = 0; x->SeedVec.clear(); x->Groups.clear(); x->Blocks.clear(); memset(x->cnt, 0, sizeof(x->cnt));
}
int ora_run_to(ora_ctx *x, int stage)
{
	while (x->stage < stage) {
		switch (++x->stage) {
		case 1: stage1_seeds(x); break;
		case 2: stage2_blocks(x); break;
		case 3: for (size_t b = 0; b < x->Blocks.size(); b++) remove_overlaps(x->Blocks[b].f); break;   // CheckAlnBlockOverlaps :232-239
		case 4: stage4_gaps(x); break;
		case 5: stage5_chrs(x); break;
		case 6: stage6_redundant(x); break;
		case 7: stage7_fill(x); break;
		case 8: stage8_align(x); break;
		default: return -1;
		}
	}
	return x->stage;
}

long long ora_seed_count(ora_ctx *x) { return (long long)x->SeedVec.size(); }
void ora_seeds(ora_ctx *x, int *qpos, int *qlen, long long *rpos)
{
	for (size_t i = 0; i < x->SeedVec.size(); i++) { qpos[i] = x->SeedVec[i].qPos; qlen[i] = x->SeedVec[i].qLen; rpos[i] = x->SeedVec[i].rPos; }
}
int ora_group_count(ora_ctx *x) { return (int)x->Groups.size(); }
void ora_groups(ora_ctx *x, int *beg, int *end) { for (size_t i = 0; i < x->Groups.size(); i++) { beg[i] = x->Groups[i].first; end[i] = x->Groups[i].second; } }
int ora_block_count(ora_ctx *x) { return (int)x->Blocks.size(); }
long long ora_frag_total(ora_ctx *x) { long long n = 0; for (size_t i = 0; i < x->Blocks.size(); i++) n += (long long)x->Blocks[i].f.size(); return n; }
long long ora_aln_total(ora_ctx *x)
{
	long long n = 0;
	for (size_t i = 0; i < x->Blocks.size(); i++) for (size_t j = 0; j < x->Blocks[i].f.size(); j++) n += (long long)x->Blocks[i].f[j].aln1.size();
	return n;
}
void ora_block_meta(ora_ctx *x, int *score, int *aln_len, int *bdup, int *nfrag, int *bdir, int *gpos, int *chr)
{
	for (size_t i = 0; i < x->Blocks.size(); i++) {
		const Block &b = x->Blocks[i];
		score[i] = b.score; nfrag[i] = (int)b.f.size();
		aln_len[i] = x->stage >= 8 ? b.aln_len : 0; bdup[i] = x->stage >= 6 ? (b.bDup ? 1 : 0) : 0;
		bdir[i] = x->stage >= 8 ? (b.bDir ? 1 : 0) : 0; gpos[i] = x->stage >= 8 ? b.gPos : 0; chr[i] = x->stage >= 8 ? b.chr : 0;
	}
}
void ora_frags(ora_ctx *x, int *bseed, int *qpos, int *qlen, long long *rpos, int *rlen, int *alnlen)
{
	size_t k = 0;
	for (size_t i = 0; i < x->Blocks.size(); i++) for (size_t j = 0; j < x->Blocks[i].f.size(); j++, k++) {
		const Frag &f = x->Blocks[i].f[j];
		bseed[k] = f.bSeed ? 1 : 0; qpos[k] = f.qPos; qlen[k] = f.qLen; rpos[k] = f.rPos; rlen[k] = f.rLen; alnlen[k] = (int)f.aln1.size();
	}
}
void ora_frag_aln(ora_ctx *x, char *a1, char *a2)
{
	size_t p = 0;
	for (size_t i = 0; i < x->Blocks.size(); i++) for (size_t j = 0; j < x->Blocks[i].f.size(); j++) {
		const Frag &f = x->Blocks[i].f[j];
		memcpy(a1 + p, f.aln1.data(), f.aln1.size()); memcpy(a2 + p, f.aln2.data(), f.aln2.size()); p += f.aln1.size();
	}
}

int ora_ksw2(const char *s1, int m, const char *s2, int n, char *out1, char *out2)
{
	std::string a(s1, m), b(s2, n);
	ksw2_align(a, b);
	memcpy(out1, a.data(), a.size()); memcpy(out2, b.data(), b.size());
	return (int)a.size();
}
int ora_ksw2_ops(const char *s1, int m, const char *s2, int n, char *ops)
{
	std::string o = ksw_ops_reversed(std::string(s1, m), std::string(s2, n));
	std::reverse(o.begin(), o.end());
	memcpy(ops, o.data(), o.size());
	return (int)o.size();
}
int ora_gap_similarity(ora_ctx *x, int q1, int q2, long long r1, long long r2) { return gap_similar(x, q1, q2, r1, r2) ? 1 : 0; }
int ora_bwt_search(ora_ctx *x, int start, int stop, int *len, long long *locs)
{
	u64 l[100]; int f = bwt_search(x, start, stop, len, l);
	for (int i = 0; i < f; i++) locs[i] = (long long)l[i];
	return f;
}
long long ora_bwt_sa(ora_ctx *x, unsigned long long k) { return (long long)locate(x, k); }
void ora_counters(ora_ctx *x, uint64_t out[8]) { memcpy(out, x->cnt, sizeof(x->cnt)); }

} // extern "C"
