// gsalign_amd/csrc/k_dp.hip -- batched gap-closing DP (a13) and the public leaf
// operator gsa_ksw2_batch.  Device code is in gsa_dp.h.
#include "gsa_ctx.h"
#include "gsa_dp.h"

__global__ void k_dp_cells(i32 n, const i32 *__restrict__ len1, const i32 *__restrict__ len2, i32 *cells)
{
	i32 i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i <= n) cells[i] = i < n ? len1[i] * len2[i] : 0;
}

// one wavefront (= one 64-thread workgroup) per alignment
__global__ void __launch_bounds__(64) k_dp_wave(i32 first, i32 n_jobs, const uint8_t *__restrict__ pool1, const i64 *__restrict__ off1, const i32 *__restrict__ len1,
                                                 const uint8_t *__restrict__ pool2, const i64 *__restrict__ off2, const i32 *__restrict__ len2,
                                                 uint8_t *dir, const i64 *__restrict__ diroff, i64 dirbase, uint8_t *rev, uint8_t *ops, const i64 *__restrict__ ops_off,
                                                 i32 *ops_len, int npad)
{
	extern __shared__ __attribute__((aligned(16))) int8_t lds[];
	const i32 job = first + blockIdx.x;
	if (job >= first + n_jobs) return;
	const int m = len1[job], n = len2[job];
	const uint8_t *s1 = pool1 + off1[job], *s2 = pool2 + off2[job];
	uint8_t *d = dir + (diroff[job] - dirbase);
	uint8_t *rv = rev + ops_off[job], *op = ops + ops_off[job];
	const int lane = threadIdx.x;
	if (m <= 0 || n <= 0) { if (lane == 0) ops_len[job] = 0; return; }
	dp_fill(s1, m, s2, n, lds, npad, d);
	__syncthreads();
	__shared__ int s_nops;
	if (lane == 0) s_nops = dp_backtrack(d, m, n, rv);
	__syncthreads();
	const int nops = s_nops;
	for (int p = lane; p < nops; p += 64) op[p] = rv[nops - 1 - p];
	if (lane == 0) ops_len[job] = nops;
}

// All pointers are device pointers.  Jobs are processed in batches so that the
// direction bytes of one batch fit the budget.
int run_ksw2_jobs(gsa_ctx *c, i32 n, const uint8_t *pool1, const i64 *off1, const i32 *len1,
                  const uint8_t *pool2, const i64 *off2, const i32 *len2, uint8_t *ops, const i64 *ops_off, i32 *ops_len)
{
	if (n <= 0) return GSA_OK;
	hipStream_t st = c->stream;
	i32 *cells = dev_ensure<i32>(c, c->d_flag2, (size_t)n + 1);
	i64 *coff = dev_ensure<i64>(c, c->j_cells, (size_t)n + 1);
	if (!cells || !coff) return GSA_ERR_NOMEM;
	hipLaunchKernelGGL(k_dp_cells, dim3(grid_for(n + 1, 256)), dim3(256), 0, st, n, len1, len2, cells);
	int rc = prim_exscan_i32_i64(c, cells, coff, (size_t)n + 1); if (rc) return rc;
	std::vector<i64> h_coff((size_t)n + 1); std::vector<i32> h_len2((size_t)n); std::vector<i64> h_ooff((size_t)n); std::vector<i32> h_len1((size_t)n);
	GSA_CHECK(c, hipMemcpyAsync(h_coff.data(), coff, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipMemcpyAsync(h_len2.data(), len2, (size_t)n * 4, hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipMemcpyAsync(h_len1.data(), len1, (size_t)n * 4, hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipMemcpyAsync(h_ooff.data(), ops_off, (size_t)n * 8, hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipStreamSynchronize(st));
	i64 ops_total = 0;
	for (i32 i = 0; i < n; i++) { i64 e = h_ooff[i] + h_len1[i] + h_len2[i]; if (e > ops_total) ops_total = e; }
	uint8_t *rev = dev_ensure<uint8_t>(c, c->d_i64a, (size_t)ops_total + 64);
	if (!rev) return GSA_ERR_NOMEM;
	const i64 budget = 6ll << 30;      // direction bytes per batch
	c->counters[4] += (u64)h_coff[n]; c->counters[5] += (u64)n;
	for (i32 i = 0; i < n; i++) c->counters[6] += (u64)h_len1[i] + (u64)h_len2[i];
	i32 first = 0;
	while (first < n) {
		i32 last = first; int nmax = 0;
		while (last < n && (last == first || h_coff[last + 1] - h_coff[first] <= budget)) { if (h_len2[last] > nmax) nmax = h_len2[last]; last++; }
		const i64 bytes = h_coff[last] - h_coff[first];
		uint8_t *dir = dev_ensure<uint8_t>(c, c->d_scan2, (size_t)bytes + 64);
		if (!dir) return GSA_ERR_NOMEM;
		const int npad = (nmax + 63) & ~63;
		if ((size_t)npad * 4 > 150 * 1024) return gsa_fail(c, GSA_ERR_LIMIT, "DP fragment longer than 38400 bases");
		hipLaunchKernelGGL(k_dp_wave, dim3(last - first), dim3(64), (size_t)npad * 4, st, first, last - first, pool1, off1, len1, pool2, off2, len2,
		                   dir, coff, h_coff[first], rev, ops, ops_off, ops_len, npad);
		first = last;
	}
	GSA_CHECK(c, hipGetLastError());
	return GSA_OK;
}

extern "C" int gsa_ksw2_batch(gsa_ctx *c, int32_t n_pairs, const char *pool1, const int64_t *off1, const int32_t *len1,
                              const char *pool2, const int64_t *off2, const int32_t *len2, char *ops, const int64_t *ops_off, int32_t *ops_len)
{
	if (!c || n_pairs < 0) return GSA_ERR_ARG;
	if (n_pairs == 0) return GSA_OK;
	hipStream_t st = c->stream;
	const size_t n = (size_t)n_pairs;
	i64 p1 = 0, p2 = 0, po = 0;
	for (size_t i = 0; i < n; i++) {
		if (len1[i] < 0 || len2[i] < 0) return gsa_fail(c, GSA_ERR_ARG, "negative fragment length");
		if (off1[i] + len1[i] > p1) p1 = off1[i] + len1[i];
		if (off2[i] + len2[i] > p2) p2 = off2[i] + len2[i];
		if (ops_off[i] + len1[i] + len2[i] > po) po = ops_off[i] + len1[i] + len2[i];
	}
	uint8_t *d_p1, *d_p2, *d_ops; i64 *d_o1, *d_o2, *d_oo; i32 *d_l1, *d_l2, *d_ol;
	GSA_CHECK(c, hipMalloc(&d_p1, p1 + 1)); GSA_CHECK(c, hipMalloc(&d_p2, p2 + 1)); GSA_CHECK(c, hipMalloc(&d_ops, po + 1));
	GSA_CHECK(c, hipMalloc(&d_o1, n * 8)); GSA_CHECK(c, hipMalloc(&d_o2, n * 8)); GSA_CHECK(c, hipMalloc(&d_oo, n * 8));
	GSA_CHECK(c, hipMalloc(&d_l1, n * 4)); GSA_CHECK(c, hipMalloc(&d_l2, n * 4)); GSA_CHECK(c, hipMalloc(&d_ol, n * 4));
	GSA_CHECK(c, hipMemcpyAsync(d_p1, pool1, p1, hipMemcpyHostToDevice, st)); GSA_CHECK(c, hipMemcpyAsync(d_p2, pool2, p2, hipMemcpyHostToDevice, st));
	GSA_CHECK(c, hipMemcpyAsync(d_o1, off1, n * 8, hipMemcpyHostToDevice, st)); GSA_CHECK(c, hipMemcpyAsync(d_o2, off2, n * 8, hipMemcpyHostToDevice, st));
	GSA_CHECK(c, hipMemcpyAsync(d_oo, ops_off, n * 8, hipMemcpyHostToDevice, st));
	GSA_CHECK(c, hipMemcpyAsync(d_l1, len1, n * 4, hipMemcpyHostToDevice, st)); GSA_CHECK(c, hipMemcpyAsync(d_l2, len2, n * 4, hipMemcpyHostToDevice, st));
	int rc = run_ksw2_jobs(c, n_pairs, d_p1, d_o1, d_l1, d_p2, d_o2, d_l2, d_ops, d_oo, d_ol);
	if (rc == GSA_OK) {
		GSA_CHECK(c, hipMemcpyAsync(ops, d_ops, po, hipMemcpyDeviceToHost, st));
		GSA_CHECK(c, hipMemcpyAsync(ops_len, d_ol, n * 4, hipMemcpyDeviceToHost, st));
		GSA_CHECK(c, hipStreamSynchronize(st));
	}
	hipFree(d_p1); hipFree(d_p2); hipFree(d_ops); hipFree(d_o1); hipFree(d_o2); hipFree(d_oo); hipFree(d_l1); hipFree(d_l2); hipFree(d_ol);
	return rc;
}
