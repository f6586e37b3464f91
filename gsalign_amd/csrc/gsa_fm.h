// gsalign_amd/csrc/gsa_fm.h -- device-side FM-index primitives (a1-a4).
//
// Reference semantics restated for a 64-lane SIMT machine:
//   bwt_occ4 / bwt_2occ4   reference src/bwt_search.cpp:69-119
//   bwt_occ / bwt_invPsi   :45-67,121-127
//   BWT_Search             :141-185
// One Occ query = one 64-byte block (4 x global_load_dwordx4 issued together),
// then three 64-bit popcounts per 32 symbols instead of the reference's byte
// table; the count is a pure function of (block, row) so results are identical.
#ifndef GSA_FM_H
#define GSA_FM_H
#include "gsa_internal.h"

__device__ __forceinline__ int gsa_nt4(uint8_t c)
{
	// nst_nt4_table (BWT_Index/bntseq.c:40-57): ACGT/acgt -> 0..3, everything else 4
	switch (c | 0x20) { case 'a': return 0; case 'c': return 1; case 'g': return 2; case 't': return 3; default: return 4; }
}

struct FmBlock { uint4 c0, c1, w0, w1; };

__device__ __forceinline__ FmBlock fm_load(const uint4 *bwt, u64 blk)
{
	const uint4 *p = bwt + (blk << 2);
	FmBlock b; b.c0 = p[0]; b.c1 = p[1]; b.w0 = p[2]; b.w1 = p[3];
	return b;
}

// counts of C,G,T among the first n (1..128) symbols of the block; A follows from n
__device__ __forceinline__ void fm_count(const FmBlock &b, int n, u32 &c1, u32 &c2, u32 &c3)
{
	const u64 M = 0x5555555555555555ull;
	u64 W[4] = { ((u64)b.w0.x << 32) | b.w0.y, ((u64)b.w0.z << 32) | b.w0.w, ((u64)b.w1.x << 32) | b.w1.y, ((u64)b.w1.z << 32) | b.w1.w };
	c1 = c2 = c3 = 0;
#pragma unroll
	for (int j = 0; j < 4; j++) {
		int nj = n - 32 * j; nj = nj < 0 ? 0 : (nj > 32 ? 32 : nj);
		u64 m = nj == 0 ? 0ull : (M & ~((nj == 32) ? 0ull : ((1ull << (64 - 2 * nj)) - 1)));
		u64 lo = W[j] & M, hi = (W[j] >> 1) & M;
		c3 += __popcll(hi & lo & m);
		c2 += __popcll(hi & ~lo & m);
		c1 += __popcll(~hi & lo & m);
	}
}

// Occ(c, k) for all four c at the rows inside one loaded block
__device__ __forceinline__ void fm_occ4_in(const FmBlock &b, int n, u64 cnt[4])
{
	u32 c1, c2, c3; fm_count(b, n, c1, c2, c3);
	cnt[0] = (((u64)b.c0.y << 32) | b.c0.x) + (u32)(n - c1 - c2 - c3);
	cnt[1] = (((u64)b.c0.w << 32) | b.c0.z) + c1;
	cnt[2] = (((u64)b.c1.y << 32) | b.c1.x) + c2;
	cnt[3] = (((u64)b.c1.w << 32) | b.c1.z) + c3;
}

// bwt_2occ4(k, l): returns the number of 64-byte blocks touched
__device__ __forceinline__ int fm_2occ4(const DevIndex &di, u64 k, u64 l, u64 ck[4], u64 cl[4])
{
	const bool kn = (k == (u64)-1), ln = (l == (u64)-1);
	u64 kk = k - (k >= di.primary), ll = l - (l >= di.primary);
	int touched = 0;
	if (!kn && !ln && (kk >> 7) == (ll >> 7)) {
		FmBlock b = fm_load(di.bwt, kk >> 7);
		fm_occ4_in(b, (int)(kk & 127) + 1, ck);
		fm_occ4_in(b, (int)(ll & 127) + 1, cl);
		return 1;
	}
	FmBlock bk, bl;
	if (!kn) bk = fm_load(di.bwt, kk >> 7);
	if (!ln) bl = fm_load(di.bwt, ll >> 7);
	if (kn) { ck[0] = ck[1] = ck[2] = ck[3] = 0; } else { fm_occ4_in(bk, (int)(kk & 127) + 1, ck); touched++; }
	if (ln) { cl[0] = cl[1] = cl[2] = cl[3] = 0; } else { fm_occ4_in(bl, (int)(ll & 127) + 1, cl); touched++; }
	return touched;
}

struct FmIntv { u64 x0, x1, x2; };

__device__ __forceinline__ FmIntv fm_init(const DevIndex &di, int p)
{
	FmIntv v; v.x0 = di.L2[p] + 1; v.x1 = di.L2[3 - p] + 1; v.x2 = di.L2[p + 1] - di.L2[p];
	return v;
}

// one forward extension by base nt (0..3); returns false if the interval dies
__device__ __forceinline__ bool fm_extend(const DevIndex &di, FmIntv &ik, int nt, u32 &blocks)
{
	u64 tk[4], tl[4];
	blocks += fm_2occ4(di, ik.x1 - 1, ik.x1 - 1 + ik.x2, tk, tl);
	u64 o2[4];
#pragma unroll
	for (int i = 0; i < 4; i++) o2[i] = tl[i] - tk[i];
	const int i = 3 - nt;
	if (o2[i] == 0) return false;
	u64 o0 = ik.x0 + ((ik.x1 <= di.primary && ik.x1 + ik.x2 - 1 >= di.primary) ? 1 : 0);   // ok[3].x0
	// ok[2].x0 = ok[3].x0 + ok[3].x2, ok[1].x0 = ..., ok[0].x0 = ...
	if (i < 3) o0 += o2[3];
	if (i < 2) o0 += o2[2];
	if (i < 1) o0 += o2[1];
	ik.x0 = o0; ik.x1 = di.L2[i] + 1 + tk[i]; ik.x2 = o2[i];
	return true;
}

// BWT_Search without the locate step: maximal forward match from `start`, capped at `stop`
__device__ __forceinline__ int fm_search(const DevIndex &di, const uint8_t *q, int start, int stop, FmIntv &ik, u32 &blocks)
{
	ik = fm_init(di, gsa_nt4(q[start]));
	int pos;
	for (pos = start + 1; pos < stop; pos++) {
		int nt = gsa_nt4(q[pos]);
		if (nt > 3) break;
		if (!fm_extend(di, ik, nt, blocks)) break;
	}
	return pos - start;
}

// bwt_invPsi: one LF step.  Symbol fetch uses k-(k>primary), Occ uses k-(k>=primary);
// the two differ only at k == primary, which maps to row 0.
__device__ __forceinline__ u64 fm_lf(const DevIndex &di, u64 k)
{
	if (k == di.primary) return 0;
	u64 x = k - (k > di.primary);
	FmBlock b = fm_load(di.bwt, x >> 7);
	int s = (int)(x & 127);
	u32 w = s < 64 ? (s < 32 ? (s < 16 ? b.w0.x : b.w0.y) : (s < 48 ? b.w0.z : b.w0.w))
	               : (s < 96 ? (s < 80 ? b.w1.x : b.w1.y) : (s < 112 ? b.w1.z : b.w1.w));
	int sym = (w >> ((~s & 15) << 1)) & 3;
	u64 cnt[4]; fm_occ4_in(b, s + 1, cnt);
	return di.L2[sym] + cnt[sym];
}

// bwt_sa: locate row k (sa_intv = 32, sampled by row)
__device__ __forceinline__ u64 fm_locate(const DevIndex &di, u64 k, u32 &steps)
{
	u64 s = 0;
	while (k & 31) { ++s; k = fm_lf(di, k); }
	steps += (u32)s;
	return s + di.sa[k >> 5];
}

#endif
