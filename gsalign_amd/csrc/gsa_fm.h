// gsalign_amd/csrc/gsa_fm.h -- device-side FM-index primitives (a1-a4).
//
// Reference semantics restated for a 64-lane SIMT machine:
//   bwt_occ4 / bwt_2occ4   reference src/bwt_search.cpp:69-119
//   bwt_occ / bwt_invPsi   :45-67,121-127
//   BWT_Search             :141-185
// One Occ query = one 32-byte block of 64 rows (2 x global_load_dwordx4, one sector of HBM: the layout below),
// then three 64-bit popcounts per 32 symbols instead of the reference's byte
// table; the count is a pure function of (block, row) so results are identical.
// Everything is written with selects instead of runtime-indexed arrays so that
// nothing spills to scratch.
#ifndef GSA_FM_H
#define GSA_FM_H
#include "gsa_internal.h"

__device__ __forceinline__ int gsa_nt4(uint8_t c)
{
	// nst_nt4_table (BWT_Index/bntseq.c:40-57): ACGT/acgt -> 0..3, everything else 4.  Pure ALU: as a switch the
	// compiler turns it into a lookup table in memory, i.e. a dependent load per base inside every base-wise loop.
	const u32 l = (u32)(c | 0x20) - 'a';                              // a 0, c 2, g 6, t 19
	const u32 x = (l >> 1) & 3;                                       // a 0, c 1, g 3 (t apart)
	const bool ok = l < 32 && ((0x00080045u >> l) & 1u);
	const u32 code = l == 19 ? 3u : (x ^ (x >> 1));                   // a 0, c 1, g 3 ^ 1 = 2, t 3
	return ok ? (int)code : 4;
}

// Occ blocks on the device (round 3, late): 64 rows per 32-BYTE block -- ONE 32-byte sector of HBM answers an Occ query, where
// the reference's interleaved block (128 rows: four 64-bit counts + 128 symbols = 64 bytes, bwt_search.cpp:69-119) costs two.
// Random 32-byte sectors, not bytes, are what the seed kernels are bound by (40-48 G sectors/s on an MI355X whatever the
// occupancy: tools/rand_probe.hip), and most of theirs are the two Occ blocks per step of the walks inside repeat copies.
//   c = counts of A, C, G, T in front of the block as u32, relative to the block's SUPER-block (2^occ_shift blocks; its 64-bit
//       base counts in di.occ_base, a table of a few entries; no table = no base: texts below 2^32 rows),
//   w = 64 symbols in the reference's bit order (first symbol in the top bits of w.x; w.x, w.y = the first 32).
// Built from the uploaded reference layout by k_occ_relayout (k_seed.hip); the counts are the reference's numbers regrouped, so
// every Occ value -- and the number of 128-row blocks the reference would have touched (row arithmetic) -- is unchanged.
struct FmBlock { uint4 c, w; u64 ba, bc, bg, bt; };

__device__ __forceinline__ FmBlock fm_load(const DevIndex &di, u64 blk)
{
	const uint4 *p = di.bwt + (blk << 1);
	FmBlock b; b.c = p[0]; b.w = p[1];
	b.ba = b.bc = b.bg = b.bt = 0;
	if (di.occ_base) { const ulonglong2 *sb = (const ulonglong2 *)(di.occ_base + ((blk >> di.occ_shift) << 2)); const ulonglong2 s0 = sb[0], s1 = sb[1]; b.ba = s0.x; b.bc = s0.y; b.bg = s1.x; b.bt = s1.y; }
	return b;
}

// counts of C,G,T among the first n (1..64) symbols of the block; A follows from n.  The first symbol sits in the top bits of
// a word, so "the first k symbols" are moved to the bottom by ONE shift and the symbols shifted in are code 0 = A, which is not
// counted: no mask.  Three popcounts per word (both bits, low bits, high bits): C = low - both, G = high - both, T = both.
__device__ __forceinline__ void fm_count(const FmBlock &b, int n, u32 &c1, u32 &c2, u32 &c3)
{
	const u64 M = 0x5555555555555555ull;
	const int n0 = n < 32 ? n : 32, n1 = n - 32;                      // symbols taken from word 0 (1..32) and word 1 (<= 0: none)
	const u64 w0 = (((u64)b.w.x << 32) | b.w.y) >> (64 - 2 * n0);
	const u64 w1 = n1 > 0 ? (((u64)b.w.z << 32) | b.w.w) >> (64 - 2 * n1) : 0ull;
	const u64 lo0 = w0 & M, hi0 = (w0 >> 1) & M, lo1 = w1 & M, hi1 = (w1 >> 1) & M;
	c3 = (u32)(__popcll(hi0 & lo0) + __popcll(hi1 & lo1));
	c1 = (u32)(__popcll(lo0) + __popcll(lo1)) - c3;
	c2 = (u32)(__popcll(hi0) + __popcll(hi1)) - c3;
}

struct Occ4 { u64 a, c, g, t; };

__device__ __forceinline__ u64 occ_sel(const Occ4 &o, int i) { return i == 0 ? o.a : (i == 1 ? o.c : (i == 2 ? o.g : o.t)); }

// Occ(c, k) for all four c at a row inside one loaded block (n = symbols up to the row, inclusive: 1..64)
__device__ __forceinline__ Occ4 fm_occ4_in(const FmBlock &b, int n)
{
	u32 c1, c2, c3; fm_count(b, n, c1, c2, c3);
	Occ4 o;
	o.a = b.ba + b.c.x + (u32)(n - c1 - c2 - c3);
	o.c = b.bc + b.c.y + c1;
	o.g = b.bg + b.c.z + c2;
	o.t = b.bt + b.c.w + c3;
	return o;
}

// bwt_2occ4(k, l): returns the number of 64-byte blocks THE REFERENCE touches (its blocks hold 128 rows).
// Branch-free on the memory side: both blocks are requested back to back (the second
// request is an L1 hit when the rows share a block) and waited for once.
__device__ __forceinline__ int fm_2occ4(const DevIndex &di, u64 k, u64 l, Occ4 &ck, Occ4 &cl)
{
	const bool kn = (k == (u64)-1), ln = (l == (u64)-1);
	const u64 kk = kn ? 0 : k - (k >= di.primary), ll = ln ? 0 : l - (l >= di.primary);
	const FmBlock bk = fm_load(di, kk >> 6);
	const FmBlock bl = fm_load(di, ll >> 6);
	const Occ4 z = {0, 0, 0, 0};
	ck = fm_occ4_in(bk, (int)(kk & 63) + 1);
	cl = fm_occ4_in(bl, (int)(ll & 63) + 1);
	if (kn) ck = z;
	if (ln) cl = z;
	if (!kn && !ln && (kk >> 7) == (ll >> 7)) return 1;
	return (kn ? 0 : 1) + (ln ? 0 : 1);
}

struct FmIntv { u64 x0, x1, x2; };

__device__ __forceinline__ u64 l2_sel(const DevIndex &di, int i) { return i == 0 ? di.L2[0] : (i == 1 ? di.L2[1] : (i == 2 ? di.L2[2] : (i == 3 ? di.L2[3] : di.L2[4]))); }

__device__ __forceinline__ FmIntv fm_init(const DevIndex &di, int p)
{
	FmIntv v; v.x0 = l2_sel(di, p) + 1; v.x1 = l2_sel(di, 3 - p) + 1; v.x2 = l2_sel(di, p + 1) - l2_sel(di, p);
	return v;
}

// one forward extension by base nt (0..3); returns false if the interval dies
__device__ __forceinline__ bool fm_extend(const DevIndex &di, FmIntv &ik, int nt, u32 &blocks)
{
	Occ4 tk, tl;
	blocks += fm_2occ4(di, ik.x1 - 1, ik.x1 - 1 + ik.x2, tk, tl);
	const u64 o2a = tl.a - tk.a, o2c = tl.c - tk.c, o2g = tl.g - tk.g, o2t = tl.t - tk.t;
	const int i = 3 - nt;
	const u64 o2i = i == 0 ? o2a : (i == 1 ? o2c : (i == 2 ? o2g : o2t));
	if (o2i == 0) return false;
	u64 o0 = ik.x0 + ((ik.x1 <= di.primary && ik.x1 + ik.x2 - 1 >= di.primary) ? 1 : 0);   // ok[3].x0
	// ok[2].x0 = ok[3].x0 + ok[3].x2, ok[1].x0 = ..., ok[0].x0 = ...
	if (i < 3) o0 += o2t;
	if (i < 2) o0 += o2g;
	if (i < 1) o0 += o2c;
	ik.x0 = o0; ik.x1 = l2_sel(di, i) + 1 + occ_sel(tk, i); ik.x2 = o2i;
	return true;
}

// the same step on two blocks that are already in registers (kk, ll = primary-corrected rows)
__device__ __forceinline__ bool fm_extend_loaded(const DevIndex &di, FmIntv &ik, int nt, const FmBlock &bk, const FmBlock &bl, u64 kk, u64 ll, bool kn, bool ln, u32 &blocks)
{
	const Occ4 z = {0, 0, 0, 0};
	Occ4 tk = fm_occ4_in(bk, (int)(kk & 63) + 1), tl = fm_occ4_in(bl, (int)(ll & 63) + 1);
	if (kn) tk = z;
	if (ln) tl = z;
	blocks += (!kn && !ln && (kk >> 7) == (ll >> 7)) ? 1 : ((kn ? 0 : 1) + (ln ? 0 : 1));
	const u64 o2a = tl.a - tk.a, o2c = tl.c - tk.c, o2g = tl.g - tk.g, o2t = tl.t - tk.t;
	const int i = 3 - nt;
	const u64 o2i = i == 0 ? o2a : (i == 1 ? o2c : (i == 2 ? o2g : o2t));
	if (o2i == 0) return false;
	u64 o0 = ik.x0 + ((ik.x1 <= di.primary && ik.x1 + ik.x2 - 1 >= di.primary) ? 1 : 0);
	if (i < 3) o0 += o2t;
	if (i < 2) o0 += o2g;
	if (i < 1) o0 += o2c;
	ik.x0 = o0; ik.x1 = l2_sel(di, i) + 1 + occ_sel(tk, i); ik.x2 = o2i;
	return true;
}

// BWT_Search without the locate step: maximal forward match from `start`, capped at
// `stop`.  codes[] holds nt4 codes (0..4), e.g. a query window staged in LDS.
__device__ __forceinline__ int fm_search_codes(const DevIndex &di, const uint8_t *codes, int start, int stop, FmIntv &ik, u32 &blocks)
{
	ik = fm_init(di, codes[start]);
	int pos;
	for (pos = start + 1; pos < stop; pos++) {
		const int nt = codes[pos];
		if (nt > 3) break;
		if (!fm_extend(di, ik, nt, blocks)) break;
	}
	return pos - start;
}

// same on raw ASCII
__device__ __forceinline__ int fm_search(const DevIndex &di, const uint8_t *q, int start, int stop, FmIntv &ik, u32 &blocks)
{
	ik = fm_init(di, gsa_nt4(q[start]));
	int pos;
	for (pos = start + 1; pos < stop; pos++) {
		const int nt = gsa_nt4(q[pos]);
		if (nt > 3) break;
		if (!fm_extend(di, ik, nt, blocks)) break;
	}
	return pos - start;
}

// Direct text comparison for a UNIQUE interval (x2 == 1).  Returns how many of the next
// (at most 16) positions satisfy pos+t < clen, tp+t < tend, codes[pos+t] <= 3 and
// ref[tp+t] == "ACGT"[codes[pos+t]], counted from t = 0 up to the first failure.
// Equivalent to that many successful bwt_2occ4 extension steps: with one occurrence
// left, a forward extension succeeds iff the text continues with the query base, and it
// leaves x0 (the row of the only suffix) and x2 = 1 unchanged.  ref is RefSequence
// (upper-case ACGT, bwt_index.cpp:199-209), allocated with 64 bytes of slack.
// The three aligned 8-byte words covering [tp, tp+16) are passed in (a = word at tp & ~7).
__device__ __forceinline__ int text_match16w(u64 a, u64 b, u64 c, i64 tp, i64 tend, const uint8_t *codes, int pos, int clen)
{
	int avail = clen - pos;
	if (tend - tp < (i64)avail) avail = (int)(tend - tp);
	if (avail > 16) avail = 16;
	if (avail <= 0) return 0;
	const int sh = (int)(tp & 7) * 8;
	const u64 lo = sh ? (a >> sh) | (b << (64 - sh)) : a;
	const u64 hi = sh ? (b >> sh) | (c << (64 - sh)) : b;
	u64 elo = 0, ehi = 0;
#pragma unroll
	for (int t = 0; t < 8; t++) {
		const int p0 = pos + t < clen ? pos + t : clen - 1, p1 = pos + 8 + t < clen ? pos + 8 + t : clen - 1;
		const u32 c0 = codes[p0], c1 = codes[p1];
		elo |= (u64)(c0 <= 3 ? (0x54474341u >> (8 * c0)) & 0xFFu : 0xFFu) << (8 * t);
		ehi |= (u64)(c1 <= 3 ? (0x54474341u >> (8 * c1)) & 0xFFu : 0xFFu) << (8 * t);
	}
	u64 x = lo ^ elo;
	int n = x ? (__ffsll((unsigned long long)x) - 1) >> 3 : 8;
	if (n == 8) { x = hi ^ ehi; n = 8 + (x ? (__ffsll((unsigned long long)x) - 1) >> 3 : 8); }
	return n < avail ? n : avail;
}

// bwt_invPsi: one LF step.  Symbol fetch uses k-(k>primary), Occ uses k-(k>=primary);
// the two differ only at k == primary, which maps to row 0.
__device__ __forceinline__ u64 fm_lf(const DevIndex &di, u64 k)
{
	if (k == di.primary) return 0;
	const u64 x = k - (k > di.primary);
	const FmBlock b = fm_load(di, x >> 6);
	const int s = (int)(x & 63);
	const u32 w = s < 32 ? (s < 16 ? b.w.x : b.w.y) : (s < 48 ? b.w.z : b.w.w);
	const int sym = (w >> ((~s & 15) << 1)) & 3;
	const Occ4 o = fm_occ4_in(b, s + 1);
	return l2_sel(di, sym) + occ_sel(o, sym);
}

// bwt_sa by walking (sa_intv = 32, sampled by row) -- used to densify the SA at
// index upload and by the leaf operator
__device__ __forceinline__ u64 fm_locate_walk(const DevIndex &di, u64 k, u32 &steps)
{
	u64 s = 0;
	while (k & 31) { ++s; k = fm_lf(di, k); }
	steps += (u32)s;
	return s + di.sa[k >> 5];
}

// bwt_sa through the dense SA built at gsa_create (one 4- or 8-byte read)
__device__ __forceinline__ u64 fm_locate(const DevIndex &di, u64 k)
{
	if (k == 0) return (u64)-1;                      // sa[0] = -1 sentinel (bwt_index.cpp:40)
	return di.sa32 ? (u64)di.sa32[k] : di.sa64[k];
}

#endif
