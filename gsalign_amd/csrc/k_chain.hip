// gsalign_amd/csrc/k_chain.hip -- stage 2: seed-group analysis / chaining (a7).
//
// Replaces GenerateAlignmentBlocks -> SeedGroupAnalysis -> {RemoveOutlierSeeds,
// RefinePDFmap, FindNeighboringPosDiffAvg, RemoveRedundantSeeds, AddAlnBlock}
// (reference src/GSAlign.cpp:29-49,145-153,178-225,245-391).
//
// The reference walks each seed group sequentially.  Here every step is a
// data-parallel pass over ALL seeds of ALL groups at once (one lane per seed,
// group boundaries carried per seed), because one group -- the main diagonal --
// usually holds most of a contig's seeds:
//   sort (group,qPos,rPos) -> unique flags -> outlier windows (the greedy window
//   segmentation becomes a "next window start" function + chain walk) -> per-
//   window PosDiff histogram via one radix sort + run lengths -> multi-hit
//   resolution from ranked alive-unique neighbours -> compaction -> noise stencil
//   -> compaction -> block cuts -> AddAlnBlock filter.
// Exactness notes: integer means truncate toward zero,
// PosDiff>>4 is an arithmetic shift, the modal bucket is the first maximum in
// ascending key order, bucket counts are read after zeroing.
#include "gsa_ctx.h"
#include "gsa_scan.h"
#include "gsa_gap.h"

#define TPB 256
#define GID(n) i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; if (i >= (n)) return

__device__ __forceinline__ i64 d_llabs(i64 v) { return v < 0 ? -v : v; }

// ---- A. (group,qPos,rPos) order ---------------------------------------------------
// The reference skips groups whose total seed length is below MinAlnBlockScore
// (GSAlign.cpp:387).  Any block cut from such a group scores below the same threshold
// and AddAlnBlock drops it (:145-153), so analysing ALL groups gives the same block
// list; that keeps the seed count a host-known launch bound (no read-back here).
__global__ void k_group_keys(i64 n, const i32 *__restrict__ s_q, const i32 *__restrict__ s_gid, int qbits, u64 *key, u32 *val)
{
	GID(n);
	key[i] = ((u64)(u32)s_gid[i] << qbits) | (u32)s_q[i];
	val[i] = (u32)i;
}

__global__ void k_gather_active(i64 na, const u32 *__restrict__ perm, const i32 *__restrict__ s_q, const i32 *__restrict__ s_len, const i64 *__restrict__ s_r,
                                const i32 *__restrict__ s_gid, const i32 *__restrict__ g_beg, i32 *a_q, i32 *a_len, i64 *a_r, i32 *a_gb, i32 *a_ge)
{
	GID(na);
	const u32 s = perm[i];
	a_q[i] = s_q[s]; a_len[i] = s_len[s]; a_r[i] = s_r[s];
	const i32 g = s_gid[s];
	a_gb[i] = g_beg[g]; a_ge[i] = g_beg[g + 1];
}

// ---- A'. the same order without the PosDiff sort ---------------------------------------
// Group starts from the bitmap of occupied PosDiff values (k_seed_select sets it): PosDiff b starts a group iff it is
// occupied and none of b-1 .. b-MaxIndelSize is (GSAlign.cpp SeedGrouping: a jump of more than MaxIndelSize between
// consecutive sorted values).  MaxIndelSize <= 31 here, so a word and its predecessor decide.
__device__ __forceinline__ u32 pd_starts(const u32 *__restrict__ bm, i64 w, int max_indel)
{
	const u64 m = ((u64)bm[w] << 32) | (w > 0 ? bm[w - 1] : 0u);
	if (max_indel <= 0) return (u32)(m >> 32);
	u64 sm = m << 1; int cov = 1;                 // OR of m shifted up by 1 .. cov
	while (2 * cov <= max_indel) { sm |= sm << cov; cov *= 2; }
	if (cov < max_indel) sm |= sm << (max_indel - cov);
	return (u32)((m & ~sm) >> 32);
}
// Round 4: the scan runs over the COARSE bitmap (one bit per block of 32 bitmap words = 1024 PosDiff values, set together with the bitmap
// by k_seed_select) and only looks into the blocks that hold a hit: against a human reference the bitmap is 776 MB per contig of which a few
// hundred thousand cache lines are occupied -- the word-by-word scan read all of it and wrote as much again (group starts below each WORD),
// which is why such contigs used to fall back to the PosDiff sort.  gpre[blk] = group starts below block blk, written for occupied blocks.
// group starts in the 32 words of block blk as 32 counts: the block is ONE 128-byte line, fetched with eight 16-byte loads at once (a loop
// of pd_starts() over the words is 64 dependent-looking loads: 4.2 instead of 1.9 ms of chaining on a 250 Mb contig when first tried)
__device__ __forceinline__ u32 pd_starts_pair(u32 cur, u32 prev, int max_indel)
{
	const u64 m = ((u64)cur << 32) | prev;
	if (max_indel <= 0) return cur;
	u64 sm = m << 1; int cov = 1;
	while (2 * cov <= max_indel) { sm |= sm << cov; cov *= 2; }
	if (cov < max_indel) sm |= sm << (max_indel - cov);
	return (u32)((m & ~sm) >> 32);
}
__device__ __forceinline__ void pd_block_load(const u32 *__restrict__ bm, i64 blk, u32 (&w)[33])
{
	const uint4 *p = (const uint4 *)(bm + (blk << 5));
	w[0] = blk > 0 ? bm[(blk << 5) - 1] : 0u;
#pragma unroll
	for (int k = 0; k < 8; k++) { const uint4 q = p[k]; w[1 + 4 * k] = q.x; w[2 + 4 * k] = q.y; w[3 + 4 * k] = q.z; w[4 + 4 * k] = q.w; }
}
struct OpPdScan {
#ifdef LB_BISECT
	static constexpr int lb_id = 0;
#endif      // over the BLOCKS (32 bitmap words each); a block without a hit costs a look at its coarse bit
	const u32 *bm, *cb; int max_indel; i32 *gpre, *mail; i64 nw;
	__device__ i32 value(i64 blk, int) const
	{
		if (!((cb[blk >> 5] >> (blk & 31)) & 1u)) return 0;
		u32 w[33]; pd_block_load(bm, blk, w);
		i32 n = 0;
#pragma unroll
		for (int k = 0; k < 32; k++) n += __popc(pd_starts_pair(w[k + 1], w[k], max_indel));
		return n;
	}
	__device__ void emit(i64 blk, const i32 *v, const i32 *ex) const { if (v[0] || ((cb[blk >> 5] >> (blk & 31)) & 1u)) gpre[blk] = ex[0]; }
	__device__ void done(const i32 *t) const { mail[M_NG] = t[0]; }
};
// Round 5: the same numbers in two passes that only look at what is occupied.  Against a human reference the bitmap has 6 M blocks per contig of which
// a few hundred thousand hold a hit; the pass over ALL blocks was 6 000 tiles of look-back machinery for coarse bits that are zero.  (1) a pass over the
// coarse WORDS (32 blocks each: 190 000 elements) lists the touched blocks in block order (value = popcount, emit = the set bits' block numbers); (2) a pass
// over that list counts the group starts of each touched block (one 128-byte line each, all independent) and leaves gpre[block].
struct OpPdTouched {
#ifdef LB_BISECT
	static constexpr int lb_id = 1;
#endif
	const u32 *cb; i64 nblk; i32 *tlist, *mail;
	__device__ i32 value(i64 w, int) const { u32 x = cb[w]; const i64 left = nblk - (w << 5); if (left < 32) x &= left <= 0 ? 0u : ((1u << left) - 1u); return __popc(x); }
	__device__ void emit(i64 w, const i32 *v, const i32 *ex) const
	{
		if (!v[0]) return;
		u32 x = cb[w]; const i64 left = nblk - (w << 5); if (left < 32) x &= left <= 0 ? 0u : ((1u << left) - 1u);
		i32 at = ex[0];
		while (x) { const int b = __ffs((int)x) - 1; x &= x - 1; tlist[at++] = (i32)((w << 5) + b); }
	}
	__device__ void done(const i32 *t) const { mail[M_NTOUCH] = t[0]; }
};
struct OpPdScanList {
#ifdef LB_BISECT
	static constexpr int lb_id = 2;
#endif
	const u32 *bm; const i32 *tlist; int max_indel; i32 *gpre, *mail;
	struct Item { i32 blk, n; };
	__device__ Item load(i64 j) const
	{
		Item it; it.blk = -1; it.n = 0;
		if (j >= mail[M_NTOUCH]) return it;
		it.blk = tlist[j];
		u32 w[33]; pd_block_load(bm, it.blk, w);
#pragma unroll
		for (int k = 0; k < 32; k++) it.n += __popc(pd_starts_pair(w[k + 1], w[k], max_indel));
		return it;
	}
	__device__ i32 value(const Item &it, i64, int) const { return it.n; }
	__device__ void emit(const Item &it, i64, const i32 *, const i32 *ex) const { if (it.blk >= 0) gpre[it.blk] = ex[0]; }
	__device__ void done(const i32 *t) const { mail[M_NG] = t[0]; }
};
// key = (group, qPos, rank among the hits of the same start); val = index of the hit
__global__ void k_pd_keys(i64 n, const u64 *__restrict__ hkey, const u32 *__restrict__ hval, const u32 *__restrict__ bm, const i32 *__restrict__ gpre, int max_indel, int qbits,
                          u64 *key, u32 *val)
{
	GID(n);
	const u64 k = hkey[i];
	const i64 pd = (i64)(k >> qbits); const u32 q = (u32)(k & ((1ull << qbits) - 1));
	const i64 w = pd >> 5; const int b = (int)(pd & 31);
	// group id = starts below my block (OpPdScan) + in the words of my block below mine + in my word up to my bit: the block is one 128-byte line
	u32 wd[33]; pd_block_load(bm, w >> 5, wd);
	const int wi = (int)(w & 31);
	i32 below = 0; u32 st = 0;
#pragma unroll
	for (int k = 0; k < 32; k++) { const u32 sk = pd_starts_pair(wd[k + 1], wd[k], max_indel); below += k < wi ? __popc(sk) : 0; st = k == wi ? sk : st; }
	const i32 gid = gpre[w >> 5] + below + __popc(st & (b == 31 ? ~0u : ((2u << b) - 1))) - 1;
	key[i] = ((u64)(u32)gid << (qbits + 7)) | ((u64)q << 7) | (hval[i] >> 16);
	val[i] = (u32)i;
}
__global__ void k_pd_heads(i64 n, const u64 *__restrict__ key, int gshift, i32 *g_beg)
{
	GID(n);
	const i32 g = (i32)(key[i] >> gshift);
	if (i == 0 || (i32)(key[i - 1] >> gshift) != g) g_beg[g] = (i32)i;
	if (i == n - 1) g_beg[g + 1] = (i32)n;
}
__global__ void k_pd_gather(i64 n, const u64 *__restrict__ key, const u32 *__restrict__ perm, const u64 *__restrict__ hkey, const u32 *__restrict__ hval, int gshift, int qbits, Bundle bnd,
                            const i32 *__restrict__ g_beg, i32 *a_q, i32 *a_len, i64 *a_r, i32 *a_gb, i32 *a_ge, u32 *bm, u32 *cb)
{
	GID(n);
	const u32 src = perm[i];
	const u64 k = hkey[src];
	const i32 q = (i32)(k & ((1ull << qbits) - 1)); const i64 pdk = (i64)(k >> qbits);
	i64 pd = pdk - bnd.lmax;                        // rPos - q  (a bundle: q is the position in the concatenation, see Bundle)
	if (bnd.n) { const i32 ci = bnd.chunk_contig[q / GSA_CHUNK]; pd -= (i64)bnd.off[ci] + (i64)ci * bnd.pds; }
	a_q[i] = q; a_len[i] = (i32)(hval[src] & 0xffffu); a_r[i] = pd + q;
	bm[pdk >> 5] = 0; cb[pdk >> 15] = 0;            // (the bitmaps are done with: wiped for the next contig by the hits that set them)
	const i32 g = (i32)(key[i] >> gshift);
	a_gb[i] = g_beg[g]; a_ge[i] = g_beg[g + 1];
}

// ---- B. unique flags, break flags, window-start candidates -------------------------
// (each struct below is one fused pass: value -> exclusive scan -> emit, see gsa_scan.h)
// uniq: no other seed of the group shares qPos (GSAlign.cpp:316-325)
// brk : unique and PosDiff differs from the previous seed (the only places where
//       an outlier window may close, :328-331)
struct OpUniqBrk {
#ifdef LB_BISECT
	static constexpr int lb_id = 3;
#endif
	static constexpr bool clamped = true;
	i64 na; const i32 *a_q; const i64 *a_r; const i32 *a_gb, *a_ge;
	i32 *uniq, *brk, *alive, *cuEx, *brkEx, *blist;
	struct Item { i32 gb, ge, q, qm, qp, u, b; i64 r, rm; };
	__device__ Item load(i64 i) const      // (loads only, neighbours at clamped indices: see lb_is_clamped; the flags are prep()'s)
	{
		const i64 im = i > 0 ? i - 1 : 0, ip = i + 1 < na ? i + 1 : i;
		Item it; it.gb = a_gb[i]; it.ge = a_ge[i]; it.q = a_q[i]; it.qm = a_q[im]; it.qp = a_q[ip]; it.r = a_r[i]; it.rm = a_r[im]; it.u = it.b = 0;
		return it;
	}
	__device__ void prep(Item &it, i64 i) const
	{
		const bool u = !(i > it.gb && it.qm == it.q) && !(i + 1 < it.ge && it.qp == it.q);
		it.u = u ? 1 : 0; it.b = (u && i > it.gb && (it.r - it.q) != (it.rm - it.qm)) ? 1 : 0;
	}
	__device__ i32 value(const Item &it, i64, int c) const { return c == 0 ? it.u : it.b; }
	__device__ void emit(const Item &, i64 i, const i32 *v, const i32 *ex) const
	{
		uniq[i] = v[0]; brk[i] = v[1]; alive[i] = 1; cuEx[i] = ex[0]; brkEx[i] = ex[1];
		if (v[1]) blist[ex[1]] = (i32)i;
	}
	__device__ void done(const i32 *t) const { cuEx[na] = t[0]; brkEx[na] = t[1]; uniq[na] = 0; brk[na] = 0; }
};

// Window starts.  Every group head is one; further starts only exist in groups with at least
// GSA_WIN_SEEDS unique seeds (a window needs that many to close, GSAlign.cpp:326-338).  Candidates =
// heads and break positions of those "big" groups; ws[] starts out as the head flags.
struct OpCand {
#ifdef LB_BISECT
	static constexpr int lb_id = 4;
#endif
	static constexpr bool clamped = true;
	i64 na; const i32 *a_gb, *a_ge, *cuEx, *brk;
	i32 *candf, *candEx, *clist, *ws;
	struct Item { i32 cand, head, gb, b, ce, cb; };
	__device__ Item load(i64 i) const
	{
		Item it; it.gb = a_gb[i]; it.b = brk[i]; it.ce = cuEx[a_ge[i]]; it.cb = cuEx[it.gb]; it.cand = it.head = 0;
		return it;
	}
	__device__ void prep(Item &it, i64 i) const { it.head = i == it.gb ? 1 : 0; it.cand = (it.ce - it.cb >= GSA_WIN_SEEDS && (i == it.gb || it.b)) ? 1 : 0; }
	__device__ i32 value(const Item &it, i64, int) const { return it.cand; }
	__device__ void emit(const Item &it, i64 i, const i32 *v, const i32 *ex) const
	{
		candf[i] = v[0]; candEx[i] = ex[0]; ws[i] = it.head;
		if (v[0]) clist[ex[0]] = (i32)i;
	}
	__device__ void done(const i32 *t) const { candEx[na] = t[0]; candf[na] = 0; ws[na] = 0; }
};

// next window start for every candidate; a group without a further window hands over to the first
// candidate behind it (the head of the next big group) or to na
__global__ void k_next_window(i64 na, const i32 *__restrict__ a_q, const i32 *__restrict__ a_gb, const i32 *__restrict__ a_ge,
                              const i32 *__restrict__ uniq, const i32 *__restrict__ cuEx, const i32 *__restrict__ brk, const i32 *__restrict__ brkEx,
                              const i32 *__restrict__ blist, const i32 *__restrict__ candf, const i32 *__restrict__ candEx, const i32 *__restrict__ clist, i32 *next, i32 *nextk,
                              uint4 *tab_clear, i64 tab_cap)
{
	// (the outlier filter's hash table -- empty = all bits set -- is cleared here, two stages ahead of its use, instead of by a fill operation of its own)
	for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < tab_cap; j += (i64)gridDim.x * blockDim.x) tab_clear[j] = make_uint4(~0u, ~0u, ~0u, ~0u);
	GID(na);
	if (!candf[i]) { next[i] = -1; return; }
	const i32 gb = a_gb[i], ge = a_ge[i];
	// n counts unique seeds: from the group head inclusive for the first window, after a restart exclusive
	const i32 base = (i == gb) ? cuEx[gb] : cuEx[i] + uniq[i];
	// first j in (i, ge) with cuIncl[j] - base >= 30   (cuIncl[j] = cuEx[j+1])
	i32 lo = (i32)i + 1, hi = ge;
	while (lo < hi) { i32 mid = (lo + hi) >> 1; if (cuEx[mid + 1] - base >= GSA_WIN_SEEDS) hi = mid; else lo = mid + 1; }
	const i32 j1 = lo;
	lo = (i32)i + 1; hi = ge;
	const i32 qi = a_q[i];
	while (lo < hi) { i32 mid = (lo + hi) >> 1; if (a_q[mid] - qi > GSA_WIN_SPAN) hi = mid; else lo = mid + 1; }
	const i32 j0 = j1 > lo ? j1 : lo;
	i32 nx = ge;
	if (j0 < ge) { const i32 nB = brkEx[na]; const i32 k = brkEx[j0]; if (k < nB) { const i32 cand = blist[k]; if (cand < ge) nx = cand; } }
	i32 nk = candEx[nx];                                   // the same hop in candidate ranks (the walk's LDS form)
	if (nx == ge) { nk = candEx[ge]; nx = nk < candEx[na] ? clist[nk] : (i32)na; }
	next[i] = nx; nextk[candEx[i]] = nk;
}

// Greedy window segmentation.  chain(p) = next[p] runs through every candidate that is a window
// start, across all big groups: ONE chain over the whole array -- sequential, but like the seed
// chunks it is a functional graph whose paths merge (window ends snap to the sparse "break"
// positions).  One 1024-lane workgroup cuts the array into tiles, walks every tile speculatively
// from its first candidate, then re-enters each tile at the previous tile's exit until no exit
// moves, and only then marks the starts.  Exactly the reference's segmentation (GSAlign.cpp:326-338).
#define WALK_T 1024
#define WALK_MAXTILES 8192
#define WALK_LDS_CAND 20480      // up to this many candidates the whole problem sits in LDS (16-bit candidate ranks)
__global__ void __launch_bounds__(WALK_T) k_walk_windows(i64 na, const i32 *__restrict__ candEx, const i32 *__restrict__ clist, const i32 *__restrict__ next, const i32 *__restrict__ nextk, i32 *ws)
{
	// one LDS arena, carved differently by the two paths: 16 KB tile bounds + 3 x 40 KB ranks + 2 x 2.5 KB bits here,
	// 2 x 32 KB tile entries / exits in the global-memory path
	__shared__ __attribute__((aligned(16))) i32 buf[2 * 2048 + 3 * WALK_LDS_CAND / 2 + 2 * WALK_LDS_CAND / 32];
	static_assert(sizeof(buf) >= 2 * WALK_MAXTILES * sizeof(i32), "arena too small for the global-memory path");
	uint16_t *s_nk = (uint16_t *)(buf + 2 * 2048);
	uint16_t (*s_J)[WALK_LDS_CAND] = (uint16_t (*)[WALK_LDS_CAND])(s_nk + WALK_LDS_CAND);
	u32 *s_on = (u32 *)(s_nk + 3 * WALK_LDS_CAND), *s_start = s_on + WALK_LDS_CAND / 32;
	__shared__ int changed;
	const int tid = threadIdx.x;
	const i32 nC = candEx[na];
	if (nC <= WALK_LDS_CAND) {
		// Candidate space, everything in LDS.  nk[k] = hop to the next start after candidate k.  The array is cut into
		// tiles; (1) one lane per tile computes, for EVERY candidate of its tile, where a walk from it leaves the tile
		// (backwards: leave(k) = next(k) if that is outside, else leave(next(k))); (2) the tile entries of the true chain
		// are the orbit of candidate 0 under leave(), chased by one lane; (3) one lane per tile walks
		// from its entry and sets a bit per start, all lanes scatter the marks.  (The earlier version re-walked tiles
		// until no exit moved: 28 passes on the bench.)
		uint16_t *nk = s_nk;
		i32 *tile_kb = buf, *tile_ke = buf + 2048;
		i64 ts = 256; while ((na + ts - 1) / ts > 2048) ts <<= 1;
		const int nt = (int)((na + ts - 1) / ts);
		for (int k0 = tid; k0 < nC; k0 += 8 * WALK_T) {
			i32 d[8];
#pragma unroll
			for (int u = 0; u < 8; u++) { const int k = k0 + u * WALK_T; d[u] = k < nC ? nextk[k] - k : 0; }
#pragma unroll
			for (int u = 0; u < 8; u++) { const int k = k0 + u * WALK_T; if (k < nC) nk[k] = (uint16_t)(d[u] < 0xffff ? d[u] : 0xffff); }
		}
		for (int w = tid; w < (nC + 31) / 32; w += WALK_T) { s_on[w] = 0; s_start[w] = 0; }
		for (int t = tid; t < nt; t += WALK_T) {
			const i64 pe = (i64)(t + 1) * ts < na ? (i64)(t + 1) * ts : na;
			tile_kb[t] = candEx[(i64)t * ts]; tile_ke[t] = candEx[pe];
		}
		__syncthreads();
#define WALK_NEXT(k) (nk[k] != 0xffff ? (k) + (i32)nk[k] : nextk[k])
		// (1) where does a walk from k leave k's tile (0xffff: past the last candidate)
		for (int t = tid; t < nt; t += WALK_T) {
			const i32 kb = tile_kb[t], ke = tile_ke[t];
			for (i32 k = ke - 1; k >= kb; k--) {
				const i32 nx = WALK_NEXT(k);
				s_J[0][k] = nx >= ke ? (uint16_t)(nx < nC ? nx : 0xffff) : s_J[0][nx];
			}
		}
		__syncthreads();
		// (2) orbit of candidate 0: one lane chases it (one dependent LDS read per tile, ~30 ns each: 9 us for 300 tiles;
		//     pointer doubling over all candidates took 23)
		if (tid == 0 && nC > 0) {
			for (u32 k = 0; k != 0xffff; k = s_J[0][k]) s_on[k >> 5] |= 1u << (k & 31);
		}
		__syncthreads();
		// (3) marks
		for (int t = tid; t < nt; t += WALK_T) {
			const i32 kb = tile_kb[t], ke = tile_ke[t];
			i32 e = -1;
			for (i32 k = kb; k < ke; k++) if ((s_on[k >> 5] >> (k & 31)) & 1u) { e = k; break; }
			if (e < 0) continue;
			for (i32 k = e; k < ke; k = WALK_NEXT(k)) atomicOr(&s_start[k >> 5], 1u << (k & 31));
		}
		__syncthreads();
		for (int k0 = tid; k0 < nC; k0 += 8 * WALK_T) {
			i32 p[8];
#pragma unroll
			for (int u = 0; u < 8; u++) { const int k = k0 + u * WALK_T; p[u] = (k < nC && ((s_start[k >> 5] >> (k & 31)) & 1u)) ? clist[k] : -1; }
#pragma unroll
			for (int u = 0; u < 8; u++) if (p[u] >= 0) ws[p[u]] = 1;
		}
#undef WALK_NEXT
		return;
	}
	// position space, the chain stays in global memory
	i32 *entry = buf, *exit_ = buf + WALK_MAXTILES;
	i64 ts = 256; while ((na + ts - 1) / ts > WALK_MAXTILES) ts <<= 1;
	const int nt = (int)((na + ts - 1) / ts);
	for (int t = tid; t < nt; t += WALK_T) {
		const i32 k = candEx[(i64)t * ts];                               // first candidate at or behind the tile start
		entry[t] = k < nC ? clist[k] : (i32)na; exit_[t] = -1;           // exit -1 = "must be (re)walked"
	}
	__syncthreads();
	for (;;) {
		for (int t = tid; t < nt; t += WALK_T) {
			if (exit_[t] >= 0) continue;                                   // entry unchanged since the last walk
			const i64 e = (i64)(t + 1) * ts < na ? (i64)(t + 1) * ts : na;
			i64 p = entry[t];
			while (p < e) { const i32 nx = next[p]; p = nx >= 0 ? nx : p + 1; }      // (entries and chain positions are candidates: nx >= 0)
			exit_[t] = (i32)p;
		}
		if (tid == 0) changed = 0;
		__syncthreads();
		// true entry of a tile = exit of the tile before it (entry[] is private to the owning lane)
		bool moved[WALK_MAXTILES / WALK_T];
		for (int t = tid, k = 0; t < nt; t += WALK_T, k++) {
			moved[k] = false;
			if (t == 0) continue;
			const i32 ne = exit_[t - 1];
			if (ne != entry[t]) { entry[t] = ne; moved[k] = true; changed = 1; }
		}
		__syncthreads();
		const int again = changed;
		for (int t = tid, k = 0; t < nt; t += WALK_T, k++) if (moved[k]) exit_[t] = -1;
		__syncthreads();
		if (!again) break;
	}
	for (int t = tid; t < nt; t += WALK_T) {
		const i64 e = (i64)(t + 1) * ts < na ? (i64)(t + 1) * ts : na;
		for (i64 p = entry[t]; p < e; p = next[p]) ws[p] = 1;
	}
}

// ---- C. outliers: per-window histogram of PosDiff>>4 over unique seeds -----------
// RefinePDFmap counts, per window, the unique seeds per PosDiff>>4 bucket (a std::map).  Here the
// (window, bucket) counts live in an open-addressing hash table in HBM (one CAS + one add per unique
// seed, no sort): the same pass that numbers the windows inserts the seeds.
struct Bucket { unsigned long long key; u32 cnt; u32 pad; };      // cleared to 0xff: key = empty, count = cnt + 1
#define BKT_EMPTY (~0ull)
// (round 5: any capacity, not a power of two -- a table of 2^k >= 2 na entries was 3 na on average, cleared and scanned once per contig: 4.6 GB per human genome;
//  the slot is the high half of hash x capacity)
__device__ __forceinline__ u32 bkt_hash(unsigned long long k, u32 cap) { return (u32)((((k * 0x9E3779B97F4A7C15ull) >> 32) * (unsigned long long)cap) >> 32); }

// window id per seed = (number of starts up to and including it) - 1
struct OpWindowBuckets {
#ifdef LB_BISECT
	static constexpr int lb_id = 5;
#endif
	static constexpr bool clamped = true;
	i64 na; const i32 *a_q; const i64 *a_r; const i32 *uniq, *ws; i64 bmin; u32 cap;
	i32 *wsEx, *slot_of; Bucket *tab; unsigned long long *wbest, *wsum; i32 *wn;
	struct Item { i32 ws, uniq; i64 pd; };
	__device__ Item load(i64 i) const { Item it; it.ws = ws[i]; it.uniq = uniq[i]; it.pd = a_r[i] - a_q[i]; return it; }
	__device__ i32 value(const Item &it, i64, int) const { return it.ws; }
	__device__ void emit(const Item &it, i64 i, const i32 *v, const i32 *ex) const
	{
		wsEx[i] = ex[0];
		if (it.ws) { wbest[ex[0]] = 0; wsum[ex[0]] = 0; wn[ex[0]] = 0; }      // (the sums of window ex[0], which this seed opens; round 5: not 20 bytes per SEED -- windows are few)
		// One table operation per DISTINCT (window, bucket) of the wavefront: its 64 lanes hold 64 consecutive seeds, which mostly share both (the main
		// diagonal: two or three keys per wavefront), so the lanes with the first pending lane's key step aside together -- that lane inserts the key and adds
		// their number; twice, then the lanes that are left (a repeat's scattered hits) go for themselves, all at once.  Measured per 250 Mb contig: every seed for
		// itself 296 us, key after key until none is left 320, one key + the rest at once 263, two keys + the rest 232.  (One compare-and-swap and one add per SEED were 5.4 M dependent atomics per 250 Mb contig: 0.30 ms, the longest of the passes.)
		i32 slot = -1;
		const u32 w = (u32)(ex[0] + v[0] - 1);
		const u32 b = (u32)((it.pd >> 4) - bmin);              // bucket, shifted to be non-negative
		const unsigned long long key = ((unsigned long long)w << 32) | b;
		bool pend = it.uniq != 0;
		for (int round = 0; round < 2 && __any(pend); round++) {
			const unsigned long long live = __ballot(pend);
			const int lead = __ffsll((long long)live) - 1;
			const unsigned long long k0 = __shfl(key, lead);
			const bool mine = pend && key == k0;
			const unsigned long long grp = __ballot(mine);
			u32 h = 0;
			if ((int)(threadIdx.x & 63) == lead) {
				h = bkt_hash(key, cap);
				for (;;) {
					const unsigned long long old = atomicCAS(&tab[h].key, BKT_EMPTY, key);
					if (old == BKT_EMPTY || old == key) break;
					h = h + 1 == cap ? 0u : h + 1;
				}
				atomicAdd(&tab[h].cnt, (u32)__popcll(grp));
			}
			h = __shfl(h, lead);
			if (mine) { slot = (i32)h; pend = false; }
		}
		if (pend) {      // (a wavefront of scattered seeds -- a repeat's hits: every lane for itself, all at once)
			u32 h = bkt_hash(key, cap);
			for (;;) {
				const unsigned long long old = atomicCAS(&tab[h].key, BKT_EMPTY, key);
				if (old == BKT_EMPTY || old == key) break;
				h = h + 1 == cap ? 0u : h + 1;
			}
			atomicAdd(&tab[h].cnt, 1u);
			slot = (i32)h;
		}
		slot_of[i] = slot;
	}
	// Round 5, late: the rows of a thread go through the table TOGETHER.  emit() above costs a row up to three round trips to the memory-side atomic unit (the
	// compare-and-swap of its first key's leader, of its second key's leader, of the lanes that are left), and the four rows of a thread took them one after the
	// other: 12 round trips per tile, the longest pass of the chain (223 us per 250 Mb contig).  Here every round issues the compare-and-swaps of all rows, then
	// looks at the answers.  Same table operations, same counts; which slot a key lands in does not matter to anybody (k_window_mode takes maxima).
	static constexpr bool has_emit_all = true;
	template <int ITEMS>
	__device__ void emit_all(const Item (&it)[ITEMS], i64 i0, const bool (&in)[ITEMS], const i32 (&v)[ITEMS], const i32 (&ex)[ITEMS]) const
	{
		const int lane = (int)(threadIdx.x & 63);
		unsigned long long key[ITEMS]; bool pend[ITEMS]; i32 slot[ITEMS];
#pragma unroll
		for (int k = 0; k < ITEMS; k++) {
			const i64 i = i0 + (i64)k * LB_TPB;
			slot[k] = -1; pend[k] = false; key[k] = 0;
			if (in[k]) {
				wsEx[i] = ex[k];
				if (it[k].ws) { wbest[ex[k]] = 0; wsum[ex[k]] = 0; wn[ex[k]] = 0; }
				const u32 w = (u32)(ex[k] + v[k] - 1);
				const u32 b = (u32)((it[k].pd >> 4) - bmin);
				key[k] = ((unsigned long long)w << 32) | b;
				pend[k] = it[k].uniq != 0;
			}
		}
		for (int round = 0; round < 2; round++) {
			int lead[ITEMS]; unsigned long long k0[ITEMS], old[ITEMS]; u32 h[ITEMS]; unsigned long long grp[ITEMS]; bool act[ITEMS], mine[ITEMS];
#pragma unroll
			for (int k = 0; k < ITEMS; k++) {
				const unsigned long long live = __ballot(pend[k]);
				act[k] = live != 0;
				lead[k] = act[k] ? __ffsll((long long)live) - 1 : 0;
				k0[k] = __shfl(key[k], lead[k]);
				mine[k] = pend[k] && key[k] == k0[k];
				grp[k] = __ballot(mine[k]);
				h[k] = bkt_hash(k0[k], cap);
				old[k] = k0[k];
			}
#pragma unroll
			for (int k = 0; k < ITEMS; k++) if (act[k] && lane == lead[k]) old[k] = atomicCAS(&tab[h[k]].key, BKT_EMPTY, k0[k]);      // (all rows' first probes in flight together)
#pragma unroll
			for (int k = 0; k < ITEMS; k++) {
				if (act[k] && lane == lead[k]) {
					while (old[k] != BKT_EMPTY && old[k] != k0[k]) { h[k] = h[k] + 1 == cap ? 0u : h[k] + 1; old[k] = atomicCAS(&tab[h[k]].key, BKT_EMPTY, k0[k]); }
					atomicAdd(&tab[h[k]].cnt, (u32)__popcll(grp[k]));
				}
				const u32 hs = __shfl(h[k], lead[k]);
				if (mine[k]) { slot[k] = (i32)hs; pend[k] = false; }
			}
		}
		{	// (what is left: scattered seeds -- a repeat's hits --, every lane for itself, all rows at once)
			u32 h[ITEMS]; unsigned long long old[ITEMS];
#pragma unroll
			for (int k = 0; k < ITEMS; k++) { h[k] = bkt_hash(key[k], cap); old[k] = key[k]; if (pend[k]) old[k] = atomicCAS(&tab[h[k]].key, BKT_EMPTY, key[k]); }
#pragma unroll
			for (int k = 0; k < ITEMS; k++) if (pend[k]) {
				while (old[k] != BKT_EMPTY && old[k] != key[k]) { h[k] = h[k] + 1 == cap ? 0u : h[k] + 1; old[k] = atomicCAS(&tab[h[k]].key, BKT_EMPTY, key[k]); }
				atomicAdd(&tab[h[k]].cnt, 1u);
				slot[k] = (i32)h[k];
			}
		}
#pragma unroll
		for (int k = 0; k < ITEMS; k++) if (in[k]) slot_of[i0 + (i64)k * LB_TPB] = slot[k];
	}
	__device__ void done(const i32 *t) const { wsEx[na] = t[0]; wbest[na] = 0; wsum[na] = 0; wn[na] = 0; }
};

// modal bucket per window: first maximum in ascending bucket order (RefinePDFmap, GSAlign.cpp:251)
__global__ void k_window_mode(i64 cap, const Bucket *__restrict__ tab, unsigned long long *wbest)
{
	GID(cap);
	const unsigned long long k = tab[i].key;
	if (k == BKT_EMPTY) return;
	atomicMax(&wbest[(u32)(k >> 32)], ((unsigned long long)(tab[i].cnt + 1u) << 32) | (0xFFFFFFFFu - (u32)k));
}

// (the seeds of a window are neighbours: their contributions are summed along the wavefront first, one pair of atomics
//  per window and wavefront instead of one per seed)
__global__ void k_window_avg(i64 na, const i32 *__restrict__ slot_of, const Bucket *__restrict__ tab, const i32 *__restrict__ a_q, const i64 *__restrict__ a_r,
                             const unsigned long long *__restrict__ wbest, const i32 *__restrict__ ws, const i32 *__restrict__ wsEx, unsigned long long *wsum, i32 *wn, Bundle bnd)
{
	const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	const int lane = threadIdx.x & 63;
	u32 w = 0xffffffffu; unsigned long long vs = 0; i32 vn = 0;
	if (i < na) {
		w = (u32)(wsEx[i] + ws[i] - 1);
		const i32 sl = slot_of[i];
		if (sl >= 0) {
			const i64 kk = (i64)(u32)tab[sl].key, mode = (i64)(0xFFFFFFFFu - (u32)wbest[w]);      // (shifted buckets: only their difference is used)
			if (d_llabs(kk - mode) < 3) { vs = (unsigned long long)(a_r[i] - a_q[i] + bundle_off(bnd, a_q[i])); vn = 1; }     // surviving bucket (:256); the TRUE PosDiff: the mean truncates toward zero
		}
	}
	for (int d = 1; d < 64; d <<= 1) {
		const u32 ow = __shfl_up(w, d); const unsigned long long os = __shfl_up(vs, d); const i32 on = __shfl_up(vn, d);
		if (lane >= d && ow == w) { vs += os; vn += on; }
	}
	const u32 nw = __shfl_down(w, 1);
	if (i < na && vn > 0 && (lane == 63 || nw != w)) { atomicAdd(&wsum[w], vs); atomicAdd(&wn[w], vn); }
}

// ---- D. multi-hit query positions (GSAlign.cpp:178-225,341-350) -------------------
// Round 5: RemoveOutlierSeeds' verdict (GSAlign.cpp:260-296; a kernel of its own until now, k_outlier_kill) is taken where the alive unique seeds are ranked:
// load() decides "outlier" from the window's modal bucket, mean and this seed's bucket count, emit() writes alive[] and the rank.
struct OpAliveUnique {
#ifdef LB_BISECT
	static constexpr int lb_id = 6;
#endif      // outlier verdict + ranks of the alive unique seeds
	static constexpr bool clamped = true;
	i64 na; const i32 *uniq; i32 *alive, *auEx, *aulist;
	const i32 *slot_of; const Bucket *tab; const i32 *a_q; const i64 *a_r; const unsigned long long *wbest, *wsum; const i32 *wn; i64 G; i32 max_indel; Bundle bnd;
	struct Item { i32 alive, uniq, sl, q, wn, boff; u32 cnt; unsigned long long key, wbest, wsum; i64 r; };
	// load(): the gathers, three levels deep (slot -> table entry -> window sums), at indices that are valid whatever the slot says; prep(): the verdict
	__device__ Item load(i64 i) const
	{
		Item it; it.uniq = uniq[i]; it.alive = 1;                                    // (every seed is alive in front of this pass: OpUniqBrk)
		it.sl = slot_of[i]; it.q = a_q[i]; it.r = a_r[i];
		const uint4 e = *(const uint4 *)&tab[it.sl < 0 ? 0 : it.sl];                  // {key lo, key hi, cnt, pad}
		it.key = ((unsigned long long)e.y << 32) | e.x; it.cnt = e.z;
		const u32 w = it.sl < 0 ? 0u : e.y;                                         // (slot < 0: no entry -- window 0's sums are read and not used)
		it.wbest = wbest[w]; it.wsum = wsum[w]; it.wn = wn[w];
		it.boff = bundle_off_flat(bnd, it.q);
		return it;
	}
	__device__ void prep(Item &it, i64) const
	{
		if (it.sl < 0) return;
		const i64 kk = (i64)(u32)it.key, mode = (i64)(0xFFFFFFFFu - (u32)it.wbest);
		const i32 cnt = d_llabs(kk - mode) < 3 ? (i32)(it.cnt + 1u) : 0;            // counts are read after zeroing (App. B #23)
		const i64 avg = it.wn > 0 ? (i64)it.wsum / it.wn : G;                        // C division: truncation toward zero
		const i64 pd = it.r - it.q + it.boff;
		if (d_llabs(avg - pd) > max_indel && cnt < 3) it.alive = 0;                 // GSAlign.cpp:290, Min_PD_Freq = 3
	}
	__device__ i32 value(const Item &it, i64, int) const { return (it.uniq && it.alive) ? 1 : 0; }
	__device__ void emit(const Item &it, i64 i, const i32 *v, const i32 *ex) const { alive[i] = it.alive; auEx[i] = ex[0]; if (v[0]) aulist[ex[0]] = (i32)i; }
	__device__ void done(const i32 *t) const { auEx[na] = t[0]; }
};

// FindNeighboringPosDiffAvg / RemoveRedundantSeeds for the multi-hit run that holds seed i (GSAlign.cpp:178-225,341-350): the member of the run that stays
// (-1: none).  Every member of a run computes the same answer from the ranked alive unique seeds around the run (a kernel of its own per run HEAD until round
// 5, k_multihit; now part of the compaction pass: a run is a handful of seeds, a hundred at most).
__device__ __forceinline__ i32 multihit_keep(i64 i, const i32 *__restrict__ a_q, const i64 *__restrict__ a_r, i32 gb, i32 ge, const i32 *__restrict__ auEx, const i32 *__restrict__ aulist,
                                             i64 G, i32 max_indel, const Bundle &bnd)
{
	const i32 q = a_q[i];
	i32 h = (i32)i; while (h > gb && a_q[h - 1] == q) h--;
	i32 j = (i32)i + 1; while (j < ge && a_q[j] == q) j++;
	i64 s1 = 0, s2 = 0; i32 n1 = 0, n2 = 0;
	for (i32 k = auEx[h] - 1; k >= auEx[gb] && n1 < 5; k--) { const i32 s = aulist[k]; s1 += a_r[s] - a_q[s]; n1++; }
	for (i32 k = auEx[j]; k < auEx[ge] && n2 < 5; k++) { const i32 s = aulist[k]; s2 += a_r[s] - a_q[s]; n2++; }
	// (a bundle: a group lies in one contig; the mean is taken over TRUE PosDiff values -- C division truncates toward zero)
	const i64 o = bundle_off(bnd, q);
	const i64 avg = (n1 > 0 || n2 > 0) ? (s1 + s2 + o * (n1 + n2)) / (n1 + n2) : (a_r[h] - q + o);
	i32 idx = -1; i64 md = G;
	for (i32 k = h; k < j; k++) { const i64 d = d_llabs((a_r[k] - q + o) - avg); if (d < max_indel && d < md) { md = d; idx = k; } }
	return idx;
}

// ---- E. compaction, noise stencil, block cuts -----------------------------------
// After the first compaction only "is my neighbour in my group" is ever asked, so the seeds
// carry a group id (the group's old begin index) instead of group bounds.
struct OpCompactAlive {
#ifdef LB_BISECT
	static constexpr int lb_id = 7;
#endif
	static constexpr bool clamped = true;
	const i32 *alive, *a_q, *a_len; const i64 *a_r; const i32 *a_gb, *a_ge, *auEx, *aulist; i64 G; i32 max_indel; Bundle bnd;
	i32 *b_q, *b_len; i64 *b_r; i32 *b_g, *mail; i64 na;
	struct Item { i32 alive, q, len, g, ge, qm, qp; i64 r; };
	__device__ Item load(i64 i) const
	{
		const i64 im = i > 0 ? i - 1 : 0, ip = i + 1 < na ? i + 1 : i;
		Item it; it.alive = alive[i]; it.q = a_q[i]; it.g = a_gb[i]; it.ge = a_ge[i]; it.len = a_len[i]; it.r = a_r[i]; it.qm = a_q[im]; it.qp = a_q[ip];
		return it;
	}
	__device__ void prep(Item &it, i64 i) const      // (the multi-hit case is rare and slow: behind everybody's loads)
	{
		it.alive = it.alive ? 1 : 0;
		const bool multi = it.alive && ((i + 1 < it.ge && it.qp == it.q) || (i > it.g && it.qm == it.q));      // a multi-hit query position: one member of the run stays at most
		if (multi && multihit_keep(i, a_q, a_r, it.g, it.ge, auEx, aulist, G, max_indel, bnd) != (i32)i) it.alive = 0;
	}
	__device__ i32 value(const Item &it, i64, int) const { return it.alive; }
	__device__ void emit(const Item &it, i64, const i32 *v, const i32 *ex) const
	{
		if (!v[0]) return;
		const i32 p = ex[0];
		b_q[p] = it.q; b_len[p] = it.len; b_r[p] = it.r; b_g[p] = it.g;
	}
	__device__ void done(const i32 *t) const { mail[M_NB] = t[0]; }
};

// 3-point noise filter (GSAlign.cpp:355-362): pure stencil on PosDiff, then compaction #2
struct OpNoise {
#ifdef LB_BISECT
	static constexpr int lb_id = 8;
#endif
	static constexpr bool clamped = true;
	const i32 *b_q, *b_len; const i64 *b_r; const i32 *b_g;
	i32 *c_q, *c_len; i64 *c_r; i32 *c_g, *mail; i64 na;
	struct Item { i32 keep, q, len, g, gm, gp, qm, qp, nb; i64 r, rm, rp; };
	__device__ Item load(i64 i) const
	{
		// (what lies beyond the nb compacted seeds is stale but allocated: read, not used)
		const i64 im = i > 0 ? i - 1 : 0, ip = i + 1 < na ? i + 1 : i;
		Item it; it.nb = mail[M_NB]; it.q = b_q[i]; it.len = b_len[i]; it.r = b_r[i]; it.g = b_g[i];
		it.gm = b_g[im]; it.gp = b_g[ip]; it.qm = b_q[im]; it.qp = b_q[ip]; it.rm = b_r[im]; it.rp = b_r[ip]; it.keep = 0;
		return it;
	}
	__device__ void prep(Item &it, i64 i) const
	{
		it.keep = i < it.nb ? 1 : 0;
		if (i > 0 && i + 1 < it.nb && it.gm == it.g && it.gp == it.g) {
			const i64 pd = it.r - it.q, p0 = it.rm - it.qm, p1 = it.rp - it.qp;
			if (d_llabs(pd - p0) > 5 && d_llabs(pd - p1) > 5) it.keep = 0;
		}
	}
	__device__ i32 value(const Item &it, i64, int) const { return it.keep; }
	__device__ void emit(const Item &it, i64, const i32 *v, const i32 *ex) const
	{
		if (!v[0]) return;
		const i32 p = ex[0];
		c_q[p] = it.q; c_len[p] = it.len; c_r[p] = it.r; c_g[p] = it.g;
	}
	__device__ void done(const i32 *t) const { mail[M_NC] = t[0]; mail[M_MAXBLK] = 0; mail[M_MAXBLK + 1] = 0; }      // (+ the slot OpBlockFilter's atomicMax fills)
};

// block heads (GSAlign.cpp:364-374): group head, query gap > MaxSeedGap, or diagonal jump > 100
// (second component: prefix sums of the seed lengths, 32-bit wrapping -- only differences over a block are used)
struct OpBlockHeads {
#ifdef LB_BISECT
	static constexpr int lb_id = 9;
#endif
	static constexpr bool clamped = true;
	i64 na; const i32 *c_q, *c_len; const i64 *c_r; const i32 *c_g;
	i32 *bhead, *bheadEx, *bstart; u32 *ps; i32 *mail;
	struct Item { i32 h, len, nc, q, qm, lenm, g, gm; i64 r, rm; };
	__device__ Item load(i64 i) const
	{
		const i64 im = i > 0 ? i - 1 : 0;
		Item it; it.nc = mail[M_NC]; it.q = c_q[i]; it.qm = c_q[im]; it.len = c_len[i]; it.lenm = c_len[im]; it.g = c_g[i]; it.gm = c_g[im]; it.r = c_r[i]; it.rm = c_r[im]; it.h = 0;
		return it;
	}
	__device__ void prep(Item &it, i64 i) const
	{
		const i64 pd = it.r - it.q, p0 = it.rm - it.qm;
		const bool head = i == 0 || it.gm != it.g || it.q - it.qm - it.lenm > GSA_MAX_SEED_GAP || d_llabs(p0 - pd) > 100;
		it.h = (i < it.nc && head) ? 1 : 0; if (i >= it.nc) it.len = 0;
	}
	__device__ i32 value(const Item &it, i64, int c) const { return c == 1 ? it.len : it.h; }
	__device__ void emit(const Item &, i64 i, const i32 *v, const i32 *ex) const { bhead[i] = v[0]; bheadEx[i] = ex[0]; ps[i] = (u32)ex[1]; if (v[0]) bstart[ex[0]] = (i32)i; }
	__device__ void done(const i32 *t) const { bheadEx[na] = t[0]; bhead[na] = 0; ps[na] = (u32)t[1]; mail[M_NBRAW] = t[0]; }
};

// AddAlnBlock filter (GSAlign.cpp:29-49) over the raw blocks + the table of the kept ones
struct OpBlockFilter {
#ifdef LB_BISECT
	static constexpr int lb_id = 10;
#endif
	static constexpr bool clamped = true;
	i64 na; const i32 *bstart, *c_q, *c_len; const u32 *ps; Params prm;
	i32 *bkeep, *bkeepEx, *blk_beg, *blk_end, *blk_score, *mail;
	__device__ void span(i64 b, i32 &s, i32 &e) const { const i32 nAll = mail[M_NBRAW]; s = bstart[b]; e = (b + 1 < nAll) ? bstart[b + 1] : mail[M_NC]; }
	struct Item { i32 keep, s, e, score; };
	__device__ Item load(i64 b) const
	{
		// (a raw block that does not exist reads stale starts: clamped into the arrays, its result not used)
		const i32 nAll = mail[M_NBRAW], nc = mail[M_NC];
		const i32 s0 = bstart[b], sn = bstart[b + 1];
		Item it; it.s = s0; it.e = (b + 1 < nAll) ? sn : nc;
		const i64 sc = it.s < 0 ? 0 : (it.s < na ? it.s : na - 1), ec = it.e < 1 ? 1 : (it.e <= na ? it.e : na);
		it.score = (i32)(ps[ec] - ps[sc]);
		const i32 region = c_q[ec - 1] + c_len[ec - 1] - c_q[sc];
		const bool drop = it.score < prm.MinAlnBlockScore || region < prm.MinAlnLength || (it.score < 1000 && (double)it.score < region * 0.05);
		it.keep = (b < nAll && !drop) ? 1 : 0;
		return it;
	}
	__device__ i32 value(const Item &it, i64, int) const { return it.keep; }
	__device__ void emit(const Item &it, i64 b, const i32 *v, const i32 *ex) const
	{
		bkeep[b] = v[0]; bkeepEx[b] = ex[0];
		if (!v[0]) return;
		blk_beg[ex[0]] = it.s; blk_end[ex[0]] = it.e; blk_score[ex[0]] = it.score;
		// the best-scoring kept block {score, its number}: OpEarlyGaps launches no early DP for small blocks inside its query span (see there)
		atomicMax((unsigned long long *)(mail + M_MAXBLK), ((unsigned long long)(u32)it.score << 32) | (u32)ex[0]);
	}
	__device__ void done(const i32 *t) const { mail[M_NBLK] = t[0]; }
};

// ---- F. the large DP gaps, two stages early ------------------------------------------
// The striped DP of the largest gap is the contig's latency floor, and almost every gap between two
// consecutive seeds of an S2 block survives S3-S6 unchanged.  So the large ones (the ones that will need
// the striped kernel) are listed HERE and launched while S3-S6 still run; stage 6 takes a result only if
// the gap it finds is exactly the one listed (same coordinates), anything else -- a gap S3 changed, a
// gap S4 keeps although it looked hopeless -- goes the normal way later.  Gaps beyond MaxSeedGap are
// always cut by S4 and are not listed; what S4's similarity test or the list logic drops was computed
// in vain.
struct OpEarlyGaps {
#ifdef LB_BISECT
	static constexpr int lb_id = 11;
#endif
	const i32 *q, *len; const i64 *r; const i32 *head, *headEx, *bkeep, *bkeepEx; i32 *bid; const uint8_t *query, *ref;
	// the kept block a seed belongs to (-1: its raw block was dropped by AddAlnBlock); written to bid[] by emit() for the
	// stages behind (was a kernel of its own in front of this pass)
	__device__ i32 bid_of(i64 i) const { const i32 b = headEx[i] + head[i] - 1; return bkeep[b] ? bkeepEx[b] : -1; }
	i32 *e_id, *e_list; i64 *off1, *off2, *opsoff; i32 *mail;
	// Round 5: WHICH blocks get their large gaps aligned early.  A stage-2 block still has to pass the redundancy filter (RemoveRedundantAlnBlocks,
	// GSAlign.cpp:415-471: a block covered to 90 % by an overlapping better one goes), and the reference only aligns the gaps of the blocks that do
	// (FillAlnBlockGaps comes behind it, :510-513).  On repeat-rich input most stage-2 blocks are copy-against-copy alignments of interspersed repeats
	// inside the span of the contig's main block -- sparse seeds, kilobase gaps -- and every one of them is removed there: the human-like workload
	// launched 12 900 workgroups of the upper size class early for 304 jobs of that class in the final result (12.5 of a contig's 30 ms).  So a block
	// whose query span lies inside the best-scoring block's and that scores under a quarter of it -- or under a sixteenth of it wherever it lies (a
	// contig cut into several main blocks by N runs: the copy-against-copy blocks inside the second and third are as redundant) -- is left to the late
	// launch (which aligns whatever survives and was not aligned early: nothing is assumed).  Only WHEN a gap is aligned changes, never what is aligned.
	const i32 *blk_beg, *blk_end, *blk_score;
	__device__ bool early_block(i32 b) const
	{
		const unsigned long long mx = *(const unsigned long long *)(mail + M_MAXBLK);
		const i32 bm = (i32)(u32)mx; const i64 sm = (i64)(mx >> 32);
		if (b == bm || (i64)blk_score[b] * 4 >= sm) return true;
		if ((i64)blk_score[b] * 16 < sm) return false;
		const i32 qs = q[blk_beg[b]], qe = q[blk_end[b] - 1] + len[blk_end[b] - 1];
		const i32 ms = q[blk_beg[bm]], me = q[blk_end[bm] - 1] + len[blk_end[bm] - 1];
		return !(qs >= ms && qe <= me);
	}
	i32 *e_rec;      // record of an early job, -1 until stage 7 finds it (k_gap_class)
	__device__ bool gap(i64 s, i32 &qp, i64 &rp, i32 &qg, i32 &rg) const
	{
		// same block, and one that AddAlnBlock keeps: the raw blocks it drops are pairs of stray seeds kilobases apart -- listing
		// their gaps (measured: right behind the block heads, 12 us earlier) costs 0.45 ms of wasted striped DP on a 50 Mb contig
		if (s + 1 >= mail[M_NC]) return false;
		{ const i32 b0 = bid_of(s); if (b0 < 0 || bid_of(s + 1) != b0 || !early_block(b0)) return false; }
		qp = q[s] + len[s]; rp = r[s] + len[s];
		qg = q[s + 1] - qp; if (qg < 0) qg = 0;
		const i64 rg64 = r[s + 1] - rp; rg = rg64 < 0 ? 0 : (i32)rg64;
		if (qg > GSA_MAX_SEED_GAP || rg64 > GSA_MAX_SEED_GAP || !dp_is_large(rg, qg)) return false;      // (cheap tests first)
		i32 mism;
		return classify_gap(query, ref, qp, rp, qg, rg, mism) == FT_DP;
	}
	// (everything a seed's verdict needs is read once, in front of the scan: Item.  Round 5, late: load() only loads -- this seed and the next, their raw blocks'
	//  keep flags and numbers, the block's score, all at clamped indices -- and prep() decides; the sequence comparison of a large gap (rare) and the span test of a
	//  middling block are prep()'s too, behind everybody's loads.  Same decisions as gap() / bid_of() / early_block() above, which the tests of stage 2 still call.)
#ifndef EG_CLAMPED
#define EG_CLAMPED false      // (measured: 236 us per 250 Mb contig with the elements one after the other, 276 with their loads together -- the verdict needs 14 values per seed, and 154 - 176 VGPRs)
#endif
	static constexpr bool clamped = EG_CLAMPED;
	struct Item { i32 in, bid, g, qp, qg, rg; i64 rp;      // in: s < the live seed count; g: a large DP gap follows s; qg / rg: its two lengths as the job list takes them
	              i32 q0, len0, q1, k0, x0, k1, x1, sc0; i64 r0, r1; };      // raw: this seed, the next one's start, keep flag / kept number of their raw blocks, score of this one's block
	__device__ Item load(i64 s) const
	{
		const i64 sp = s + 1 < na ? s + 1 : s;
		Item it; it.in = 0; it.bid = -1; it.g = 0; it.qp = it.qg = it.rg = 0; it.rp = 0;
		it.q0 = q[s]; it.len0 = len[s]; it.r0 = r[s]; it.q1 = q[sp]; it.r1 = r[sp];
		i64 b0 = (i64)headEx[s] + head[s] - 1, b1 = (i64)headEx[sp] + head[sp] - 1;
		b0 = b0 < 0 ? 0 : (b0 < na ? b0 : na - 1); b1 = b1 < 0 ? 0 : (b1 < na ? b1 : na - 1);
		it.k0 = bkeep[b0]; it.x0 = bkeepEx[b0]; it.k1 = bkeep[b1]; it.x1 = bkeepEx[b1];
		const i64 kb = it.x0 < 0 ? 0 : (it.x0 <= na ? it.x0 : na);
		it.sc0 = blk_score[kb];
		return it;
	}
	__device__ void prep(Item &it, i64 s) const
	{
		const i32 nc = mail[M_NC];
		if (s >= nc) return;
		it.in = 1; it.bid = it.k0 ? it.x0 : -1;
		if (s + 1 >= nc) return;
		const i32 b0 = it.bid, b1 = it.k1 ? it.x1 : -1;
		if (b0 < 0 || b1 != b0) return;
		{	// early_block(b0) with the score at hand
			const unsigned long long mx = *(const unsigned long long *)(mail + M_MAXBLK);
			const i32 bm = (i32)(u32)mx; const i64 sm = (i64)(mx >> 32);
			if (!(b0 == bm || (i64)it.sc0 * 4 >= sm)) {
				if ((i64)it.sc0 * 16 < sm) return;
				const i32 qs = q[blk_beg[b0]], qe = q[blk_end[b0] - 1] + len[blk_end[b0] - 1];
				const i32 ms = q[blk_beg[bm]], me = q[blk_end[bm] - 1] + len[blk_end[bm] - 1];
				if (qs >= ms && qe <= me) return;
			}
		}
		const i32 qp = it.q0 + it.len0; const i64 rp = it.r0 + it.len0;
		i32 qg = it.q1 - qp; if (qg < 0) qg = 0;
		const i64 rg64 = it.r1 - rp; const i32 rg = rg64 < 0 ? 0 : (i32)rg64;
		if (qg > GSA_MAX_SEED_GAP || rg64 > GSA_MAX_SEED_GAP || !dp_is_large(rg, qg)) return;      // (cheap tests first)
		i32 mism;
		if (classify_gap(query, ref, qp, rp, qg, rg, mism) != FT_DP) return;
		it.g = 1; it.qp = qp; it.rp = rp; it.qg = it.q1 - qp; it.rg = (i32)(it.r1 - rp);
	}
	__device__ i32 value(const Item &it, i64, int c) const { return c == 1 ? (it.g ? it.qg + it.rg : 0) : it.g; }
	__device__ void emit(const Item &it, i64 s, const i32 *v, const i32 *ex) const
	{
		if (it.in) bid[s] = it.bid;
		e_id[s] = v[0] ? ex[0] : -1;
		if (!v[0]) return;
		const i32 e = ex[0];
		lb_pub(&e_list[3 * e], e); lb_pub(&e_list[3 * e + 1], it.rg); lb_pub(&e_list[3 * e + 2], it.qg);      // (finish() reads the list)
		off1[e] = it.rp; off2[e] = it.qp; opsoff[e] = ex[1]; e_rec[e] = -1;
	}
	__device__ void done(const i32 *t) const { lb_pub(&mail[M_NEARLY], t[0]); lb_pub(&mail[M_EOPS], t[1]); lb_pub(&mail[M_DPERR3], 0); }
	// the last tile puts the two counts and the head of the list into pinned memory: the host launches from there
	i32 *h_early; i32 h_cap;
	i64 na;      // (elements of the launch: the arrays' length; set behind the aggregate)
	__device__ void finish(int tid) const
	{
		const i32 ne = __hip_atomic_load(&mail[M_NEARLY], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (tid == 0) { h_early[0] = ne; h_early[1] = __hip_atomic_load(&mail[M_EOPS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
		const i32 w = 3 * (ne < h_cap ? ne : h_cap);
		lb_copy_out(h_early + 4, e_list, w, tid);
	}
};
#define EARLY_CHUNK 4096      // large gaps copied with the first look (more -> a second copy)

// host side of F: called from stage 3 once its passes are enqueued (the list has arrived long before)
int launch_early_dp(gsa_ctx *c)
{
	if (!c->early_listed) return GSA_OK;      // (nothing listed, or already launched)
	c->early_listed = false;
	GSA_CHECK(c, hipEventSynchronize(c->ev[16]));
	const i32 *h = c->p_early.as<i32>();
	const i32 ne = h[0]; const i64 eops = h[1];
	if (ne <= 0) return GSA_OK;
	const size_t first_e = (size_t)std::min<i64>(c->n_a, EARLY_CHUNK);
	hipStream_t sa = c->stream_aux[0];
	if ((size_t)ne > first_e) {
		if (!pin_ensure<i32>(c, c->p_early, 4 + 3 * (size_t)ne)) return GSA_ERR_NOMEM;
		h = c->p_early.as<i32>();
		GSA_CHECK(c, hipMemcpyAsync(c->p_early.as<i32>() + 4, c->e_list.p, (size_t)ne * 12, hipMemcpyDeviceToHost, sa));
		GSA_CHECK(c, hipStreamSynchronize(sa));
	}
	c->h_early.assign(h + 4, h + 4 + 3 * (size_t)ne);
	std::vector<LgJob> large((const LgJob *)(h + 4), (const LgJob *)(h + 4) + ne);
	if (!dev_ensure<uint8_t>(c, c->e_ops, (size_t)eops + 64) || !dev_ensure<uint8_t>(c, c->e_rev, (size_t)eops + 64) || !dev_ensure<i32>(c, c->e_nops, (size_t)ne + 1) || !dev_ensure<i32>(c, c->e_rec, (size_t)ne + 1)) return GSA_ERR_NOMEM;
	// stream_aux[0] already waits for the list (ev[16] was recorded behind it on the main stream)
	GSA_CHECK(c, hipStreamWaitEvent(sa, c->ev[16], 0));
	int rc = launch_stripes(c, sa, large, c->di.ref, c->e_off1.as<i64>(), c->q_dev, c->e_off2.as<i64>(),
	                        c->e_ops.as<uint8_t>(), c->e_opsoff.as<i64>(), c->e_nops.as<i32>(), c->e_rev.as<uint8_t>(), M_DPERR3);
	if (rc) return rc;
	GSA_CHECK(c, hipEventRecord(c->ev[14], sa));
	c->n_early = ne; c->early_in_flight = true; c->dbg[6] = (u64)ne;      // (gsa_get_seed_stats[6]: large gaps launched early, [7]: large jobs of the late launch)
	return GSA_OK;
}

#define LAUNCH(k, n, ...) hipLaunchKernelGGL(k, dim3(grid_for((size_t)(n), TPB)), dim3(TPB), 0, st, __VA_ARGS__)
#define ENS(T, buf, n) do { if (!dev_ensure<T>(c, c->buf, (size_t)(n))) return GSA_ERR_NOMEM; } while (0)
#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// The same segmentation for contigs whose chain does not fit one workgroup's LDS (round 5: ONE launch; until then leave() per tile, 13-14
// pointer-doubling launches for the orbit of candidate 0 and a marking launch -- each of the doubling launches a few microseconds of work that,
// with four contexts in flight, waited ~80 us for its turn: 3.8 % of all kernel time, profiles/archive/r04_kernels_human_full.txt).
// All in candidate space: nextk[k] = rank of the next window start after candidate k (nC: none).  The candidates are cut into SLICES of 16 384; a
// workgroup takes a slice (by ticket: the slices in front of it belong to workgroups that already run), keeps its hops in LDS as 16-bit offsets and
// computes, for EVERY candidate k of the slice, the last element of the walk from k that still lies in k's sub-tile (64 candidates: E1, one lane
// per sub-tile, back to front), in k's mid-tile (1 024: E2, a wavefront per mid-tile, sub-tile by sub-tile from the back -- a sub-tile's 64 values
// only need values of later sub-tiles) and in the slice (E3, mid-tile by mid-tile from the back, 1 024 lanes at once).  All of that needs no entry
// point and runs in every slice at the same time.  What is sequential is one look-up per slice: the slice's entry e arrives in a self-validating
// word {launch epoch, rank}, its exit is nextk[E3[e]], which is published as the entry of the slice it lands in (~2 us per slice: 60 slices for
// the 1 M candidates of a 250 Mb contig).  The marks then go level by level: one lane walks the mid-tile entries (h(E2)), 16 lanes the sub-tile
// entries (h(E1)), 256 lanes the starts inside their sub-tile, all lanes scatter ws[].  Exactly the orbit of candidate 0 under next().
#define WC_SL 16384
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_walk_chain keeps 4 x 32 KB of slice tables in LDS: gfx950 (160 KB of LDS per CU) only"
#endif
#define WC_T 1024
__global__ void __launch_bounds__(WC_T) k_walk_chain(i64 na, const i32 *__restrict__ candEx, const i32 *__restrict__ clist, const i32 *__restrict__ nextk, i32 *ws,
                                                     u32 *ticket, u32 ticket0, unsigned long long *entry_w, u32 epoch, i32 *err)
{
	__shared__ uint16_t h[WC_SL], E1[WC_SL], E2[WC_SL], E3[WC_SL];
	__shared__ u32 on[WC_SL / 32];
	__shared__ uint16_t mid_e[WC_SL / 1024], sub_e[WC_SL / 64];
	__shared__ u32 s_slice; __shared__ i32 s_entry;
	const int tid = threadIdx.x;
	const i32 nC = candEx[na];
	if (tid == 0) s_slice = atomicAdd(ticket, 1u) - ticket0;
	__syncthreads();
	const i32 sl = (i32)s_slice, base = sl * WC_SL;
	const i32 nsl = (nC + WC_SL - 1) / WC_SL;
	if (sl >= nsl) return;
	const i32 cnt = nC - base < WC_SL ? nC - base : WC_SL;
	for (int k = tid; k < WC_SL; k += WC_T) {
		u32 v = 0xffffu;
		if (k < cnt) { const i32 nx = nextk[base + k] - base; if (nx < cnt) v = (u32)nx; }      // (next() only moves forward: nx > k)
		h[k] = (uint16_t)v;
	}
	for (int w = tid; w < WC_SL / 32; w += WC_T) on[w] = 0;
	if (tid < WC_SL / 1024) mid_e[tid] = 0xffff;
	if (tid < WC_SL / 64) sub_e[tid] = 0xffff;
	__syncthreads();
	if (tid < WC_SL / 64) {
		const int b = tid * 64;
		for (int k = b + 63; k >= b; k--) { const u32 nx = h[k]; E1[k] = (uint16_t)((nx != 0xffffu && (nx >> 6) == (u32)tid) ? E1[nx] : (u32)k); }
	}
	__syncthreads();
	{
		const u32 w = (u32)tid >> 6, l = (u32)tid & 63u;      // (16 wavefronts, 16 mid-tiles)
		for (int j = 15; j >= 0; j--) {
			const u32 k = w * 1024u + (u32)j * 64u + l;
			const u32 e1 = E1[k], nx = h[e1];
			E2[k] = (uint16_t)((nx != 0xffffu && (nx >> 10) == w) ? E2[nx] : e1);
			__syncthreads();
		}
	}
	for (int j = WC_SL / 1024 - 1; j >= 0; j--) {
		const u32 k = (u32)j * 1024u + (u32)tid;
		const u32 e2 = E2[k], nx = h[e2];
		E3[k] = (uint16_t)(nx != 0xffffu ? E3[nx] : e2);
		__syncthreads();
	}
	if (tid == 0) {
		i32 e = -2;
		if (sl == 0) e = 0;
		else {
			u32 spins = 0; unsigned long long t_spin0 = 0;
			for (;;) {
				const unsigned long long wv = __hip_atomic_load(&entry_w[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if ((u32)(wv >> 32) == epoch) { e = (i32)(u32)wv - 2; break; }
				if ((++spins & 1023u) == 0 && (t_spin0 == 0 ? (t_spin0 = wall_clock64(), false) : wall_clock64() - t_spin0 > LB_WAIT_TICKS)) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }      // (5 s of wall clock: gsa_scan.h)
				__builtin_amdgcn_s_sleep(1);
			}
		}
		s_entry = e;
		if (e >= 0) {
			const i32 x = nextk[base + (i32)E3[e - base]];      // where the walk leaves the slice: a later slice's entry, or nC
			const i32 t = x < nC ? x / WC_SL : nsl;
			if (t < nsl) __hip_atomic_store(&entry_w[t], ((unsigned long long)epoch << 32) | (u32)(x + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			for (i32 s2 = sl + 1; s2 < t; s2++) __hip_atomic_store(&entry_w[s2], (unsigned long long)epoch << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (slices the walk jumps over: no entry)
		}
	}
	__syncthreads();
	const i32 e = s_entry;
	if (e < 0) return;
	if (tid == 0) { u32 x = (u32)(e - base); for (;;) { mid_e[x >> 10] = (uint16_t)x; const u32 nx = h[E2[x]]; if (nx == 0xffffu) break; x = nx; } }
	__syncthreads();
	if (tid < WC_SL / 1024 && mid_e[tid] != 0xffff) { u32 x = mid_e[tid]; for (;;) { sub_e[x >> 6] = (uint16_t)x; const u32 nx = h[E1[x]]; if (nx == 0xffffu || (nx >> 10) != (u32)tid) break; x = nx; } }
	__syncthreads();
	if (tid < WC_SL / 64 && sub_e[tid] != 0xffff) { u32 x = sub_e[tid]; for (;;) { atomicOr(&on[x >> 5], 1u << (x & 31)); const u32 nx = h[x]; if (nx == 0xffffu || (nx >> 6) != (u32)tid) break; x = nx; } }
	__syncthreads();
	for (int k0 = tid; k0 < cnt; k0 += 8 * WC_T) {
		i32 p[8];
#pragma unroll
		for (int u = 0; u < 8; u++) { const int k = k0 + u * WC_T; p[u] = (k < cnt && ((on[k >> 5] >> (k & 31)) & 1u)) ? clist[base + k] : -1; }
#pragma unroll
		for (int u = 0; u < 8; u++) if (p[u] >= 0) ws[p[u]] = 1;
	}
}

int stage2_chain(gsa_ctx *c)
{
	hipStream_t st = c->stream;
	c->n_blocks2 = 0; c->n_c = 0; c->n_b = 0; c->n_a = 0; c->blocks.clear(); c->s2_host = true;
	c->n_early = 0; c->early_listed = false;
	c->h_blk_beg.clear(); c->h_blk_end.clear(); c->h_blk_score.clear();
	const i64 n = c->n_seeds;
	if (n == 0) return GSA_OK;
	c->s2_host = false;
	if (c->profiling) hipEventRecord(c->ev[4], st);
	// A. all seeds in (group, qPos, rPos) order
	const i64 na = n; c->n_a = na;
	ENS(i64, d_i64a, n + 2); ENS(i32, d_flag2, n + 2); ENS(i32, d_scan2, n + 2); ENS(i32, d_flag, n + 1); ENS(i32, d_scan, n + 1);
	const int gbits = ceil_log2_u64((u64)n + 1);
	ENS(i32, a_q, na); ENS(i32, a_len, na); ENS(i64, a_r, na); ENS(i32, a_gb, na); ENS(i32, a_ge, na);
	if (c->pd_path && c->qbits + 7 + gbits <= 64) {
		// group ids from the PosDiff bitmap, then ONE sort by (group, qPos, rank) straight from the located hits
		const i64 nw = c->pd_words;
		const i64 nblk = (nw + 31) >> 5;      // blocks of 32 bitmap words (one coarse bit each)
		ENS(i32, d_gpre, ((nw + 31) >> 5) + 2); ENS(u64, d_key_c, n); ENS(u32, d_val_c, n); ENS(u64, d_key_b, n); ENS(u32, d_val_b, n); ENS(i32, g_beg, n + 2);
		i32 *mail_ = c->d_mail.as<i32>();
		if (nblk > c->opt.pd_two_level_min) {
			// (a large bitmap -- a human reference: 6 M blocks per contig --: the touched blocks are listed first, then only those are counted)
			const i64 nw2 = (nblk + 31) >> 5;
			{ OpPdTouched op = { c->d_pdcb.as<u32>(), nblk, c->d_flag.as<i32>(), mail_ }; RC((lb_launch<1, 4>(c, nw2, op))); }
			{ OpPdScanList op = { c->d_pdbm.as<u32>(), c->d_flag.as<i32>(), c->prm.MaxIndelSize, c->d_gpre.as<i32>(), mail_ }; RC((lb_launch<1, 4>(c, n < nblk ? n : nblk, op, nullptr, 8))); }
		} else
		{ OpPdScan op = { c->d_pdbm.as<u32>(), c->d_pdcb.as<u32>(), c->prm.MaxIndelSize, c->d_gpre.as<i32>(), mail_, nw };
		  if (nblk <= (1 << 20)) RC((lb_launch<1, 1>(c, nblk, op, nullptr, 8))); else RC((lb_launch<1, 4>(c, nblk, op, nullptr, 8))); }      // (a small bitmap: a block per thread -- four in a row are four times two dependent looks)
		{}      // (eight workgroups per CU: a block costs two dependent looks and nothing else)      // (16 bitmap words per thread: a popcount each -- the pass is the 94 MB read of a 250 Mb contig's bitmap, not 23 000 tiles of look-back)
		LAUNCH(k_pd_keys, n, n, c->d_key_a.as<u64>(), c->d_val_a.as<u32>(), c->d_pdbm.as<u32>(), c->d_gpre.as<i32>(), c->prm.MaxIndelSize, c->qbits, c->d_key_c.as<u64>(), c->d_val_c.as<u32>());
		// (hits straight from k_seed_select are in (qPos, rank) order already: the stable sort only has to order the group ids -- three passes for 22 bits.
		//  Hits that arrived from other GPUs' chunk ranges sit behind each other in arrival order: the whole key)
		RC(gsa_sort_pairs_u64_u32(c, c->d_key_c.as<u64>(), c->d_key_b.as<u64>(), c->d_val_c.as<u32>(), c->d_val_b.as<u32>(), (size_t)na, c->hits_sorted ? c->qbits + 7 : 0, c->qbits + 7 + gbits));
		LAUNCH(k_pd_heads, n, n, c->d_key_b.as<u64>(), c->qbits + 7, c->g_beg.as<i32>());
		LAUNCH(k_pd_gather, n, n, c->d_key_b.as<u64>(), c->d_val_b.as<u32>(), c->d_key_a.as<u64>(), c->d_val_a.as<u32>(), c->qbits + 7, c->qbits, c->bnd, c->g_beg.as<i32>(),
		       c->a_q.as<i32>(), c->a_len.as<i32>(), c->a_r.as<i64>(), c->a_gb.as<i32>(), c->a_ge.as<i32>(), c->d_pdbm.as<u32>(), c->d_pdcb.as<u32>());
		c->pdbm_dirty = false;
	} else {
		RC(seed_view_sort(c));
		ENS(u64, d_key_a, n); ENS(u64, d_key_b, n); ENS(u32, d_val_a, n); ENS(u32, d_val_b, n);
		LAUNCH(k_group_keys, n, n, c->s_q.as<i32>(), c->s_gid.as<i32>(), c->qbits, c->d_key_a.as<u64>(), c->d_val_a.as<u32>());
		RC(gsa_sort_pairs_u64_u32(c, c->d_key_a.as<u64>(), c->d_key_b.as<u64>(), c->d_val_a.as<u32>(), c->d_val_b.as<u32>(), (size_t)na, 0, c->qbits + gbits));
		LAUNCH(k_gather_active, na, na, c->d_val_b.as<u32>(), c->s_q.as<i32>(), c->s_len.as<i32>(), c->s_r.as<i64>(), c->s_gid.as<i32>(), c->g_beg.as<i32>(),
		       c->a_q.as<i32>(), c->a_len.as<i32>(), c->a_r.as<i64>(), c->a_gb.as<i32>(), c->a_ge.as<i32>());
	}
	// B. unique / break flags, window chain
	ENS(i32, a_uniq, na + 1); ENS(i32, a_cu, na + 1); ENS(i32, a_alive, na + 1); ENS(i32, a_brk, na + 1); ENS(i32, a_aurank, na + 1);
	ENS(i32, a_aulist, na + 1); ENS(i32, a_next, na + 1); ENS(i32, a_ws, na + 1); ENS(i32, a_wid, na + 1); ENS(i32, a_runinfo, na + 1);
	i32 *uniq = c->a_uniq.as<i32>(), *cuEx = c->a_cu.as<i32>(), *alive = c->a_alive.as<i32>(), *brk = c->a_brk.as<i32>();
	i32 *brkEx = c->a_aurank.as<i32>(), *blist = c->a_aulist.as<i32>(), *next = c->a_next.as<i32>(), *ws = c->a_ws.as<i32>(), *wsEx = c->a_wid.as<i32>();
	i32 *mail = c->d_mail.as<i32>();
	{ OpUniqBrk op = { na, c->a_q.as<i32>(), c->a_r.as<i64>(), c->a_gb.as<i32>(), c->a_ge.as<i32>(), uniq, brk, alive, cuEx, brkEx, blist }; RC((lb_launch<2>(c, na, op))); }
	i32 *candf = c->d_flag.as<i32>(), *candEx = c->d_scan.as<i32>(), *clist = c->a_runinfo.as<i32>();
	{ OpCand op = { na, c->a_gb.as<i32>(), c->a_ge.as<i32>(), cuEx, brk, candf, candEx, clist, ws }; RC((lb_launch<1>(c, na, op))); }
	const i64 cap = na + na / 4 + 1024;      // the outlier filter's (window, bucket) table: one key per unique seed at most -- load factor <= 0.8 if every seed is unique and alone in its bucket, a few per cent on real input
	ENS(Bucket, d_btab, cap);
	static_assert(sizeof(Bucket) == 16, "the table is cleared as uint4s");
	LAUNCH(k_next_window, na, na, c->a_q.as<i32>(), c->a_gb.as<i32>(), c->a_ge.as<i32>(), uniq, cuEx, brk, brkEx, blist, candf, candEx, clist, next, c->d_flag2.as<i32>(), c->d_btab.as<uint4>(), cap);
	if (na <= c->opt.walk_chain_min) hipLaunchKernelGGL(k_walk_windows, dim3(1), dim3(WALK_T), 0, st, na, candEx, clist, next, c->d_flag2.as<i32>(), ws);
	else {
		// (large contigs: slices of the candidate list, one launch -- k_walk_chain; candidates <= seeds bounds the grid, the ticket counter and the
		//  entry words are never reset: the host passes the counter's value and the launch's epoch)
		const u32 grid = (u32)((na + WC_SL - 1) / WC_SL);
		const size_t cap0 = c->w_j0.cap;
		ENS(unsigned long long, w_j0, (size_t)grid + 8);
		if (c->w_j0.cap != cap0) { GSA_CHECK(c, hipMemsetAsync(c->w_j0.p, 0, c->w_j0.cap, st)); c->walk_ticket = 0; c->walk_epoch = 0; }
		unsigned long long *words = c->w_j0.as<unsigned long long>();
		hipLaunchKernelGGL(k_walk_chain, dim3(grid), dim3(WC_T), 0, st, na, candEx, clist, (const i32 *)c->d_flag2.as<i32>(), ws, (u32 *)words, c->walk_ticket, words + 4, c->walk_epoch + 1, mail + M_LBERR);
		// (the slice numbering of every later launch depends on these two: they advance only for a launch that was accepted)
		GSA_CHECK(c, hipGetLastError());
		c->walk_epoch++; c->walk_ticket += grid;
	}
	// C. outliers
	ENS(unsigned long long, w_best, na + 1); ENS(unsigned long long, w_sum, na + 1); ENS(i32, w_n, na + 1);
	const i64 bmin = ((-(i64)c->qlen) >> 4) - 1;
	i32 *slot_of = c->a_runinfo.as<i32>();
	{ OpWindowBuckets op = { na, c->a_q.as<i32>(), c->a_r.as<i64>(), uniq, ws, bmin, (u32)cap, wsEx, slot_of, c->d_btab.as<Bucket>(),
	                         c->w_best.as<unsigned long long>(), c->w_sum.as<unsigned long long>(), c->w_n.as<i32>() }; RC((lb_launch<1, 4>(c, na, op))); }
	LAUNCH(k_window_mode, cap, cap, c->d_btab.as<Bucket>(), c->w_best.as<unsigned long long>());
	LAUNCH(k_window_avg, na, na, slot_of, c->d_btab.as<Bucket>(), c->a_q.as<i32>(), c->a_r.as<i64>(), c->w_best.as<unsigned long long>(), ws, wsEx,
	       c->w_sum.as<unsigned long long>(), c->w_n.as<i32>(), c->bnd);
	// D. outlier verdict + ranks of the alive unique seeds (one pass); the multi-hit positions are settled inside the compaction pass
	i32 *auEx = c->a_aurank.as<i32>(), *aulist = c->a_aulist.as<i32>();
	{ OpAliveUnique op = { na, uniq, alive, auEx, aulist, slot_of, c->d_btab.as<Bucket>(), c->a_q.as<i32>(), c->a_r.as<i64>(),
	                       c->w_best.as<unsigned long long>(), c->w_sum.as<unsigned long long>(), c->w_n.as<i32>(), c->G, c->prm.MaxIndelSize, c->bnd }; RC((lb_launch<1>(c, na, op))); }
	// E. compaction #1, noise stencil + compaction #2 (counts stay in the mailbox)
	ENS(i32, b_q, na); ENS(i32, b_len, na); ENS(i64, b_r, na); ENS(i32, b_gb, na);
	{ OpCompactAlive op = { alive, c->a_q.as<i32>(), c->a_len.as<i32>(), c->a_r.as<i64>(), c->a_gb.as<i32>(), c->a_ge.as<i32>(), auEx, aulist, c->G, c->prm.MaxIndelSize, c->bnd,
	                        c->b_q.as<i32>(), c->b_len.as<i32>(), c->b_r.as<i64>(), c->b_gb.as<i32>(), mail, na }; RC((lb_launch<1>(c, na, op))); }
	ENS(i32, c_q, na); ENS(i32, c_len, na + 1); ENS(i64, c_r, na); ENS(i32, c_gb, na); ENS(i32, c_bid, na);
	{ OpNoise op = { c->b_q.as<i32>(), c->b_len.as<i32>(), c->b_r.as<i64>(), c->b_gb.as<i32>(),
	                 c->c_q.as<i32>(), c->c_len.as<i32>(), c->c_r.as<i64>(), c->c_gb.as<i32>(), mail, na }; RC((lb_launch<1>(c, na, op))); }
	// block cuts + AddAlnBlock
	i32 *bhead = c->a_uniq.as<i32>(), *bheadEx = c->a_cu.as<i32>(), *bstart = c->a_brk.as<i32>();
	{ OpBlockHeads op = { na, c->c_q.as<i32>(), c->c_len.as<i32>(), c->c_r.as<i64>(), c->c_gb.as<i32>(), bhead, bheadEx, bstart, c->d_flag2.as<u32>(), mail }; RC((lb_launch<2>(c, na, op))); }
	i32 *bkeep = c->a_ws.as<i32>(), *bkeepEx = c->a_wid.as<i32>();
	ENS(i32, blk_beg, na + 1); ENS(i32, blk_end, na + 1); ENS(i32, blk_score, na + 1);
	{ OpBlockFilter op = { na, bstart, c->c_q.as<i32>(), c->c_len.as<i32>(), c->d_flag2.as<u32>(), c->prm, bkeep, bkeepEx,
	                       c->blk_beg.as<i32>(), c->blk_end.as<i32>(), c->blk_score.as<i32>(), mail }; RC((lb_launch<1>(c, na, op))); }
	if (c->profiling) { hipEventRecord(c->ev[5], st); c->ev_pending |= 2; }
	// F. list the large DP gaps; the list travels to the host while stage 3 is being enqueued
	ENS(i32, e_id, na + 2); ENS(i32, e_rec, na + 2); ENS(i32, e_list, 3 * (na + 1)); ENS(i64, e_off1, na + 1); ENS(i64, e_off2, na + 1); ENS(i64, e_opsoff, na + 2);
	if (!pin_ensure<i32>(c, c->p_early, 4 + 3 * (size_t)EARLY_CHUNK)) return GSA_ERR_NOMEM;
	{ OpEarlyGaps op = { c->c_q.as<i32>(), c->c_len.as<i32>(), c->c_r.as<i64>(), bhead, bheadEx, bkeep, bkeepEx, c->c_bid.as<i32>(), c->q_dev, c->di.ref,
	                     c->e_id.as<i32>(), c->e_list.as<i32>(), c->e_off1.as<i64>(), c->e_off2.as<i64>(), c->e_opsoff.as<i64>(), mail,
	                     c->blk_beg.as<i32>(), c->blk_end.as<i32>(), c->blk_score.as<i32>(), c->e_rec.as<i32>(),
	                     c->p_early.as<i32>(), (i32)std::min<i64>(na, EARLY_CHUNK) }; op.na = na; RC((lb_launch<2>(c, na, op))); }
	GSA_CHECK(c, hipEventRecord(c->ev[16], st));
	c->early_listed = true;
	return GSA_OK;      // counts stay in the mailbox; stage 3 reads them with its own first read-back
}

// Host copies of the stage-2 counts and block table (stage views, tests).
int stage2_fetch_host(gsa_ctx *c)
{
	if (c->s2_host) return GSA_OK;
	GSA_CHECK(c, hipStreamSynchronize(c->stream));
	GSA_CHECK(c, hipMemcpy(c->h_mail, c->d_mail.p, MAIL_N * sizeof(i32), hipMemcpyDeviceToHost));
	c->n_b = c->h_mail[M_NB]; c->n_c = c->h_mail[M_NC]; c->n_blocks2 = c->h_mail[M_NBLK];
	const i32 nblk = c->n_blocks2;
	c->h_blk_beg.resize(nblk); c->h_blk_end.resize(nblk); c->h_blk_score.resize(nblk);
	if (nblk > 0) {
		GSA_CHECK(c, hipMemcpy(c->h_blk_beg.data(), c->blk_beg.p, (size_t)nblk * 4, hipMemcpyDeviceToHost));
		GSA_CHECK(c, hipMemcpy(c->h_blk_end.data(), c->blk_end.p, (size_t)nblk * 4, hipMemcpyDeviceToHost));
		GSA_CHECK(c, hipMemcpy(c->h_blk_score.data(), c->blk_score.p, (size_t)nblk * 4, hipMemcpyDeviceToHost));
	}
	c->s2_host = true;
	return GSA_OK;
}

#ifdef LB_TIMING
// experiment build: tick sums of the fused passes of THIS translation unit (chaining): [0] ticket, [1] loads + scan, [2] look-back, [3] emit, [4] tiles; reset = 1 clears them
extern "C" int gsa_debug_lb_prof(unsigned long long *out, int reset)
{
	if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lb_prof), sizeof(g_lb_prof)) != hipSuccess) return -1;
	if (reset) { unsigned long long z[8] = { 0 }; if (hipMemcpyToSymbol(HIP_SYMBOL(g_lb_prof), z, sizeof(z)) != hipSuccess) return -1; }
	return 0;
}
#endif
