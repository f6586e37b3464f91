// gsalign_amd/csrc/gsa_sort.hip -- the pipeline's own LSD radix sort of (u64 key, u32 value) pairs (round 4: rocPRIM's radix sort was the
// one library kernel on the hot path -- eleven digit passes with two fill kernels each on 57-bit keys).  Stable, 8 bits per pass, three
// launches per pass and no fill kernel:
//   k_rs_hist     a tile of 4096 keys -> its 256 digit counts, written digit-major (hist[d * n_tiles + tile])
//   OpRsScan      exclusive prefix over hist in that order (a fused look-back pass, gsa_scan.h): where the keys of (digit, tile) go
//   k_rs_scatter  the tile again: every key's rank among the keys of its digit IN INPUT ORDER (stability: the callers sort by the group id alone
//                 on top of the (qPos, rank) order the seeds leave k_seed_select in), then the scatter
// A tile is four waves x 16 rows x 64 lanes: wave w owns 1024 consecutive keys, row k of it 64 consecutive ones (coalesced loads), so
// "input order" is (wave, row, lane).  The rank inside a wave comes from eight ballots per row (the lanes that hold my digit) and a per-wave
// digit counter in LDS that only that wave touches; the waves' counts are chained digit by digit after a barrier.
#include "gsa_scan.h"

#define RS_ITEMS 16
#define RS_TILE (256 * RS_ITEMS)

__global__ void __launch_bounds__(256) k_rs_hist(i64 n, const u64 *__restrict__ key, int shift, u32 dmask, u32 *hist, u32 n_tiles)
{
	__shared__ u32 h[256];
	const u32 tile = blockIdx.x, tid = threadIdx.x;
	h[tid] = 0;
	__syncthreads();
	const i64 base = (i64)tile * RS_TILE + (tid >> 6) * (RS_TILE / 4) + (tid & 63);
	// (all sixteen keys of the thread first -- at clamped indices, so that no load sits behind a branch and they go out together; round 5: one waited load per key before)
	u64 kk[RS_ITEMS];
#pragma unroll
	for (int k = 0; k < RS_ITEMS; k++) { const i64 i = base + (i64)k * 64; kk[k] = key[i < n ? i : n - 1]; }
#pragma unroll
	for (int k = 0; k < RS_ITEMS; k++) { const i64 i = base + (i64)k * 64; if (i < n) atomicAdd(&h[(u32)(kk[k] >> shift) & dmask], 1u); }
	__syncthreads();
	hist[(size_t)tid * n_tiles + tile] = h[tid];
}

struct OpRsScan {
	const u32 *h; u32 *o;
	__device__ i32 value(i64 i, int) const { return (i32)h[i]; }
	__device__ void emit(i64 i, const i32 *, const i32 *ex) const { o[i] = (u32)ex[0]; }
	__device__ void done(const i32 *) const {}
};

__global__ void __launch_bounds__(256) k_rs_scatter(i64 n, const u64 *__restrict__ kin, const u32 *__restrict__ vin, u64 *kout, u32 *vout, int shift, u32 dmask,
                                                    const u32 *__restrict__ scan, u32 n_tiles)
{
	__shared__ u32 cnt[4][256];
	const u32 tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#pragma unroll
	for (int w = 0; w < 4; w++) cnt[w][tid] = 0;
	__syncthreads();
	const i64 base = (i64)tile * RS_TILE + wv * (RS_TILE / 4) + lane;
	u64 kk[RS_ITEMS]; u32 rk[RS_ITEMS], vv[RS_ITEMS];
	const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
	// (keys and values of the thread's sixteen rows first, at clamped indices: all loads in flight together)
#pragma unroll
	for (int k = 0; k < RS_ITEMS; k++) { const i64 i = base + (i64)k * 64, ic = i < n ? i : n - 1; kk[k] = kin[ic]; vv[k] = vin[ic]; }
#pragma unroll
	for (int k = 0; k < RS_ITEMS; k++) {
		const i64 i = base + (i64)k * 64;
		const bool valid = i < n;
		if (!valid) kk[k] = 0;
		const u32 d = (u32)(kk[k] >> shift) & dmask;
		// the lanes of this row that hold my digit
		unsigned long long m = __ballot(valid);
#pragma unroll
		for (int b = 0; b < 8; b++) { const bool bit = (d >> b) & 1u; const unsigned long long bal = __ballot(bit); m &= bit ? bal : ~bal; }
		u32 r = 0;
		if (valid) {
			const u32 prior = cnt[wv][d];                          // (only this wave touches cnt[wv]: the rows before this one)
			r = prior + (u32)__popcll(m & lt);
			if ((m & lt) == 0) cnt[wv][d] = prior + (u32)__popcll(m);      // the first lane of the group moves the counter on
		}
		rk[k] = r | (d << 16);
	}
	__syncthreads();
	{	// digit tid: where this tile's keys of that digit start, wave after wave
		u32 run = scan[(size_t)tid * n_tiles + tile];
#pragma unroll
		for (int w = 0; w < 4; w++) { const u32 c = cnt[w][tid]; cnt[w][tid] = run; run += c; }
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < RS_ITEMS; k++) {
		const i64 i = base + (i64)k * 64;
		if (i < n) { const u32 pos = cnt[wv][rk[k] >> 16] + (rk[k] & 0xffffu); kout[pos] = kk[k]; vout[pos] = vv[k]; }
	}
}

// kin / vin (n pairs, left as they are) -> kout / vout, sorted by bits [begin_bit, end_bit) of the key, stable.  Passes ping-pong between
// kout / vout and a scratch pair in c->tmp, the first one reads kin / vin, the last one writes kout / vout.
int gsa_sort_pairs_u64_u32(gsa_ctx *c, const u64 *kin, u64 *kout, const u32 *vin, u32 *vout, size_t n, int begin_bit, int end_bit)
{
	if (n == 0) return GSA_OK;
	if (end_bit <= begin_bit) end_bit = begin_bit + 1;
	const int bits = end_bit - begin_bit, passes = (bits + 7) / 8;
	const u32 n_tiles = (u32)((n + RS_TILE - 1) / RS_TILE);
	const size_t hwords = (size_t)256 * n_tiles;
	const size_t scratch = passes > 1 ? ((n * 8 + 255) & ~(size_t)255) + n * 4 : 0;
	const size_t need = ((scratch + 255) & ~(size_t)255) + 2 * hwords * 4 + 1024;
	if (need > c->tmp.cap) {
		if (c->tmp.p) { ctx_quiesce(c); hipFree(c->tmp.p); c->tmp.p = nullptr; c->tmp.cap = 0; }
		const size_t want = need + need / 4 + 4096;
		if (hipMalloc(&c->tmp.p, want) != hipSuccess) return gsa_fail(c, GSA_ERR_NOMEM, "hipMalloc (sort scratch)");
		c->tmp.cap = want;
	}
	uint8_t *t = (uint8_t *)c->tmp.p;
	u64 *ks = (u64 *)t; u32 *vs = (u32 *)(t + ((n * 8 + 255) & ~(size_t)255));
	u32 *hist = (u32 *)(t + ((scratch + 255) & ~(size_t)255)), *scan = hist + hwords;
	hipStream_t st = c->stream;
	const u64 *ki = kin; const u32 *vi = vin;
	for (int p = 0; p < passes; p++) {
		const int shift = begin_bit + 8 * p, w = bits - 8 * p < 8 ? bits - 8 * p : 8;
		const u32 dmask = (1u << w) - 1u;
		// the last pass writes kout: with an odd number of passes the first one does too
		const bool to_out = ((passes - 1 - p) & 1) == 0;
		u64 *ko = to_out ? kout : ks; u32 *vo = to_out ? vout : vs;
		hipLaunchKernelGGL(k_rs_hist, dim3(n_tiles), dim3(256), 0, st, (i64)n, ki, shift, dmask, hist, n_tiles);
		{ OpRsScan op = { hist, scan }; int rc = lb_launch<1, 4>(c, (i64)hwords, op, st); if (rc) return rc; }
		hipLaunchKernelGGL(k_rs_scatter, dim3(n_tiles), dim3(256), 0, st, (i64)n, ki, vi, ko, vo, shift, dmask, (const u32 *)scan, n_tiles);
		GSA_CHECK(c, hipGetLastError());
		ki = ko; vi = vo;
	}
	return GSA_OK;
}
