// gsalign_amd/csrc/k_seed.hip -- stage 1: seed exploration (a4, a5), locate (a3),
// ordering by (PosDiff, qPos) and SeedGrouping (a6).
//
// Replaces IdentifyLocalMEM + BWT_Search + bwt_sa + SeedGrouping
// (reference src/GSAlign.cpp:51-107,126-143; src/bwt_search.cpp:121-185).
#include "gsa_ctx.h"
#include "gsa_fm.h"

enum { CNT_OCCBLK = 0, CNT_LF = 1, CNT_HITS = 2, CNT_SEEDS = 3, CNT_DPCELLS = 4, CNT_DPJOBS = 5, CNT_DPMN = 6, CNT_MEMS = 8, CNT_OVERFLOW = 9 };

// ---------------------------------------------------------------------------
// Seed exploration.  Work unit = one 10 000-bp chunk (absolute position, App. B
// #1).  The walk inside a chunk is a chain: next start = start+len+1 after a hit
// (start+5 with -sen), start+1 after a miss, so one lane owns one chunk.
// Every accepted match appends its freq BWT rows to the pending-hit list.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_seed_chunks(DevIndex di, const uint8_t *__restrict__ q, i32 qlen, Params prm,
                                                     u64 *cnt, u64 *hit_row, i32 *hit_qpos, i32 *hit_len, u64 hit_cap)
{
	const i64 chunk = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	i64 s0 = chunk * GSA_CHUNK;
	if (s0 >= qlen) return;
	int start = (int)s0, stop = (int)(s0 + GSA_CHUNK < qlen ? s0 + GSA_CHUNK : qlen);
	u32 blocks = 0;
	while (start < stop) {
		if (gsa_nt4(q[start]) > 3) { start++; continue; }
		FmIntv ik;
		int len = fm_search(di, q, start, stop, ik, blocks);
		if (len >= prm.MinSeedLength && ik.x2 <= GSA_MAX_SEED_FREQ) {
			u32 f = (u32)ik.x2;
			u64 off = atomicAdd((unsigned long long *)&cnt[CNT_HITS], (unsigned long long)f);
			if (off + f <= hit_cap) {
				for (u32 i = 0; i < f; i++) { hit_row[off + i] = ik.x0 + i; hit_qpos[off + i] = start; hit_len[off + i] = len; }
			} else cnt[CNT_OVERFLOW] = 1;
			start += prm.bSensitive ? 5 : len + 1;
		} else start++;
	}
	atomicAdd((unsigned long long *)&cnt[CNT_OCCBLK], (unsigned long long)blocks);
}

// ---------------------------------------------------------------------------
// Locate: one lane per pending hit, ~31 dependent LF steps each (a3).  Emits the
// 64-bit sort key ((PosDiff + qlen) << qbits) | qPos and the seed length.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_locate(DevIndex di, i64 n, const u64 *__restrict__ hit_row, const i32 *__restrict__ hit_qpos,
                                                 const i32 *__restrict__ hit_len, i32 qlen, int qbits, u64 *key, u32 *val, u64 *cnt)
{
	i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	u32 steps = 0;
	if (i < n) {
		u64 r = fm_locate(di, hit_row[i], steps);
		i64 pd = (i64)r - hit_qpos[i] + qlen;
		key[i] = ((u64)pd << qbits) | (u32)hit_qpos[i];
		val[i] = (u32)hit_len[i];
	}
	// one atomic per wave
	for (int o = 32; o; o >>= 1) steps += __shfl_down(steps, o);
	if ((threadIdx.x & 63) == 0 && steps) atomicAdd((unsigned long long *)&cnt[CNT_LF], (unsigned long long)steps);
}

// sorted keys -> SoA seeds + "new group starts here" flag (SeedGrouping, a6)
__global__ void k_decode_seeds(i64 n, const u64 *__restrict__ key, const u32 *__restrict__ val, i32 qlen, int qbits, i32 max_indel,
                               i32 *s_q, i32 *s_len, i64 *s_r, i32 *flag)
{
	i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > n) return;
	if (i == n) { flag[n] = 0; return; }
	const u64 qmask = (1ull << qbits) - 1;
	u64 k = key[i];
	i32 qp = (i32)(k & qmask); i64 pd = (i64)(k >> qbits) - qlen;
	s_q[i] = qp; s_len[i] = (i32)val[i]; s_r[i] = pd + qp;
	i32 f = 1;
	if (i > 0) { i64 pd0 = (i64)(key[i - 1] >> qbits) - qlen; f = (pd - pd0 > max_indel) ? 1 : 0; }
	flag[i] = f;
}

__global__ void k_group_ids(i64 n, const i32 *__restrict__ flag, const i32 *__restrict__ ex, i32 *gid, i32 *g_beg)
{
	i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > n) return;
	if (i == n) { g_beg[ex[n]] = (i32)n; return; }
	i32 g = ex[i] + flag[i] - 1;
	gid[i] = g;
	if (flag[i]) g_beg[g] = (i32)i;
}

int stage1_seed(gsa_ctx *c)
{
	const i32 qlen = c->qlen;
	hipStream_t st = c->stream;
	GSA_CHECK(c, hipMemsetAsync(c->d_cnt.p, 0, 16 * sizeof(u64), st));
	c->n_seeds = 0; c->n_groups = 0;
	if (qlen <= 0) return GSA_OK;
	const i64 n_chunks = ((i64)qlen + GSA_CHUNK - 1) / GSA_CHUNK;
	// pending-hit capacity: grows and retries on overflow
	size_t cap = c->d_hit_row.cap / sizeof(u64);
	if (cap < (size_t)qlen / 16 + 4096) cap = (size_t)qlen / 16 + 4096;
	i64 n_hits = 0;
	for (int attempt = 0; attempt < 8; attempt++) {
		if (!dev_ensure<u64>(c, c->d_hit_row, cap) || !dev_ensure<i32>(c, c->d_hit_qpos, cap) || !dev_ensure<i32>(c, c->d_hit_len, cap)) return GSA_ERR_NOMEM;
		GSA_CHECK(c, hipMemsetAsync(c->d_cnt.p, 0, 16 * sizeof(u64), st));
		if (c->profiling) hipEventRecord(c->ev[0], st);
		hipLaunchKernelGGL(k_seed_chunks, dim3(grid_for(n_chunks, 64)), dim3(64), 0, st, c->di, c->d_query.as<uint8_t>(), qlen, c->prm,
		                   c->d_cnt.as<u64>(), c->d_hit_row.as<u64>(), c->d_hit_qpos.as<i32>(), c->d_hit_len.as<i32>(), (u64)cap);
		if (c->profiling) hipEventRecord(c->ev[1], st);
		GSA_CHECK(c, hipMemcpyAsync(c->h_cnt, c->d_cnt.p, 16 * sizeof(u64), hipMemcpyDeviceToHost, st));
		GSA_CHECK(c, hipStreamSynchronize(st));
		n_hits = (i64)c->h_cnt[CNT_HITS];
		if (!c->h_cnt[CNT_OVERFLOW]) break;
		cap = (size_t)n_hits + (size_t)n_hits / 8 + 4096;
		if (attempt == 7) return gsa_fail(c, GSA_ERR_LIMIT, "pending-hit buffer overflow");
	}
	c->n_seeds = n_hits;
	if (n_hits == 0) return GSA_OK;
	if (n_hits >= (1ll << 31) - 2) return gsa_fail(c, GSA_ERR_LIMIT, "more than 2^31 seeds in one contig");
	const size_t n = (size_t)n_hits;
	if (!dev_ensure<u64>(c, c->d_key_a, n) || !dev_ensure<u64>(c, c->d_key_b, n) || !dev_ensure<u32>(c, c->d_val_a, n) || !dev_ensure<u32>(c, c->d_val_b, n)) return GSA_ERR_NOMEM;
	hipLaunchKernelGGL(k_locate, dim3(grid_for(n, 256)), dim3(256), 0, st, c->di, (i64)n, c->d_hit_row.as<u64>(), c->d_hit_qpos.as<i32>(), c->d_hit_len.as<i32>(),
	                   qlen, c->qbits, c->d_key_a.as<u64>(), c->d_val_a.as<u32>(), c->d_cnt.as<u64>());
	if (c->profiling) hipEventRecord(c->ev[2], st);
	int rc = prim_sort_pairs_u64_u32(c, c->d_key_a.as<u64>(), c->d_key_b.as<u64>(), c->d_val_a.as<u32>(), c->d_val_b.as<u32>(), n, 0, c->qbits + c->pdbits);
	if (rc) return rc;
	if (!dev_ensure<i32>(c, c->s_q, n) || !dev_ensure<i32>(c, c->s_len, n) || !dev_ensure<i64>(c, c->s_r, n) || !dev_ensure<i32>(c, c->s_gid, n) ||
	    !dev_ensure<i32>(c, c->d_flag, n + 1) || !dev_ensure<i32>(c, c->d_scan, n + 1) || !dev_ensure<i32>(c, c->g_beg, n + 1)) return GSA_ERR_NOMEM;
	hipLaunchKernelGGL(k_decode_seeds, dim3(grid_for(n + 1, 256)), dim3(256), 0, st, (i64)n, c->d_key_b.as<u64>(), c->d_val_b.as<u32>(), qlen, c->qbits, c->prm.MaxIndelSize,
	                   c->s_q.as<i32>(), c->s_len.as<i32>(), c->s_r.as<i64>(), c->d_flag.as<i32>());
	rc = prim_exscan_i32(c, c->d_flag.as<i32>(), c->d_scan.as<i32>(), n + 1);
	if (rc) return rc;
	hipLaunchKernelGGL(k_group_ids, dim3(grid_for(n + 1, 256)), dim3(256), 0, st, (i64)n, c->d_flag.as<i32>(), c->d_scan.as<i32>(), c->s_gid.as<i32>(), c->g_beg.as<i32>());
	if (c->profiling) hipEventRecord(c->ev[3], st);
	i32 ng = 0;
	GSA_CHECK(c, hipMemcpyAsync(&ng, c->d_scan.as<i32>() + n, sizeof(i32), hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipMemcpyAsync(c->h_cnt, c->d_cnt.p, 16 * sizeof(u64), hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipStreamSynchronize(st));
	c->n_groups = ng;
	c->counters[0] = c->h_cnt[CNT_OCCBLK]; c->counters[1] = c->h_cnt[CNT_LF]; c->counters[2] = (u64)n_hits; c->counters[3] = (u64)n_hits;
	if (c->profiling) {
		float ms;
		hipEventElapsedTime(&ms, c->ev[0], c->ev[1]); c->kernel_ms[0] = ms;
		hipEventElapsedTime(&ms, c->ev[1], c->ev[2]); c->kernel_ms[1] = ms;
		hipEventElapsedTime(&ms, c->ev[2], c->ev[3]); c->kernel_ms[2] = ms;
	}
	return GSA_OK;
}

// ---------------------------------------------------------------------------
// leaf operator: BWT_Search for explicit windows (gsa_bwt_search_batch)
// ---------------------------------------------------------------------------
__global__ void k_search_batch(DevIndex di, const uint8_t *__restrict__ q, Params prm, i32 n, const i32 *start, const i32 *stop,
                               i32 *out_len, i32 *out_freq, i64 *out_loc)
{
	i32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	FmIntv ik; u32 blocks = 0, steps = 0;
	int len = fm_search(di, q, start[i], stop[i], ik, blocks);
	out_len[i] = len;
	int f = 0;
	if (len >= prm.MinSeedLength && ik.x2 <= GSA_MAX_SEED_FREQ) {
		f = (int)ik.x2;
		for (int h = 0; h < f; h++) out_loc[(i64)i * GSA_MAX_SEED_FREQ + h] = (i64)fm_locate(di, ik.x0 + h, steps);
	}
	out_freq[i] = f;
}

extern "C" int gsa_bwt_search_batch(gsa_ctx *c, int32_t n, const int32_t *start, const int32_t *stop, int32_t *out_len, int32_t *out_freq, int64_t *out_loc)
{
	if (!c || n < 0) return GSA_ERR_ARG;
	if (c->qlen <= 0) return gsa_fail(c, GSA_ERR_STATE, "gsa_set_query first");
	if (n == 0) return GSA_OK;
	for (int i = 0; i < n; i++) if (start[i] < 0 || start[i] >= c->qlen || stop[i] > c->qlen || stop[i] <= start[i]) return gsa_fail(c, GSA_ERR_ARG, "window out of range");
	hipStream_t st = c->stream;
	i32 *d_start = nullptr, *d_stop = nullptr, *d_len = nullptr, *d_freq = nullptr; i64 *d_loc = nullptr;
	GSA_CHECK(c, hipMalloc(&d_start, n * 4)); GSA_CHECK(c, hipMalloc(&d_stop, n * 4)); GSA_CHECK(c, hipMalloc(&d_len, n * 4)); GSA_CHECK(c, hipMalloc(&d_freq, n * 4));
	GSA_CHECK(c, hipMalloc(&d_loc, (size_t)n * GSA_MAX_SEED_FREQ * 8));
	GSA_CHECK(c, hipMemcpyAsync(d_start, start, n * 4, hipMemcpyHostToDevice, st));
	GSA_CHECK(c, hipMemcpyAsync(d_stop, stop, n * 4, hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(k_search_batch, dim3(grid_for(n, 64)), dim3(64), 0, st, c->di, c->d_query.as<uint8_t>(), c->prm, n, d_start, d_stop, d_len, d_freq, d_loc);
	GSA_CHECK(c, hipMemcpyAsync(out_len, d_len, n * 4, hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipMemcpyAsync(out_freq, d_freq, n * 4, hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipMemcpyAsync(out_loc, d_loc, (size_t)n * GSA_MAX_SEED_FREQ * 8, hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipStreamSynchronize(st));
	hipFree(d_start); hipFree(d_stop); hipFree(d_len); hipFree(d_freq); hipFree(d_loc);
	return GSA_OK;
}
