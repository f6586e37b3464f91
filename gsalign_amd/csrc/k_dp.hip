// gsalign_amd/csrc/k_dp.hip -- batched gap-closing DP (a13) and the public leaf
// operator gsa_ksw2_batch.  Cell recurrence and traceback automaton: gsa_dp.h.
//
// Two kernels, chosen per job by size (jobs are launched largest first):
//  * k_dp_small  n <= 64 and m+n-1 <= 128 (the bulk of the jobs: median 11 x 11).
//    One wavefront per alignment, lane t owns target column t.  The (u,v,x,y) state
//    lives in REGISTERS; the left neighbour and the reference base travel one lane
//    up per anti-diagonal with DPP wave shifts (systolic array); direction bytes
//    and the traceback stay in LDS.  No barrier, no global traffic but the result.
//  * k_dp_wg<T>  everything else (up to 5000 x 5000): one T-thread workgroup per
//    alignment, state in LDS (x and v ping-pong so one barrier per anti-diagonal
//    suffices), direction bytes to HBM diagonal-major (coalesced), traceback by
//    one lane on 64x64 tiles staged through LDS.
#include <algorithm>
#include "gsa_ctx.h"
#include "gsa_dp.h"

#define SMALL_ROWS 128
#define SMALL_WAVES 4

__global__ void __launch_bounds__(64 * SMALL_WAVES) k_dp_small(i32 n_jobs, const i32 *__restrict__ order, const uint8_t *__restrict__ pool1, const i64 *__restrict__ off1,
                                                                const i32 *__restrict__ len1, const uint8_t *__restrict__ pool2, const i64 *__restrict__ off2,
                                                                const i32 *__restrict__ len2, uint8_t *ops, const i64 *__restrict__ ops_off, i32 *ops_len)
{
	__shared__ uint8_t s_dir[SMALL_WAVES][SMALL_ROWS * 64];
	__shared__ uint8_t s_rev[SMALL_WAVES][SMALL_ROWS + 64];
	__shared__ int s_n[SMALL_WAVES];
	const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const i32 slot = blockIdx.x * SMALL_WAVES + w;
	if (slot >= n_jobs) return;
	const i32 job = order[slot];
	const int m = len1[job], n = len2[job];
	const uint8_t *s1 = pool1 + off1[job], *s2 = pool2 + off2[job];
	uint8_t *dir = s_dir[w], *rev = s_rev[w];
	const int cq = lane < n ? gsa_nt4(s2[lane]) : 4;
	// reference base for lane t at diagonal r is s1[r - t]: it enters at lane 0 and moves one lane up per diagonal
	const int c1a = lane < m ? gsa_nt4(s1[lane]) : 4, c1b = lane + 64 < m ? gsa_nt4(s1[lane + 64]) : 4;
	int u = lane ? 2 : 0, v = 0, x = 0, y = 0, wref = 4;
	const int nr = m + n - 1;
	for (int r = 0; r < nr; r++) {
		const int inb = r < m ? (r < 64 ? __shfl(c1a, r) : __shfl(c1b, r - 64)) : 4;      // s1[r] broadcast
		wref = wave_shr1(wref, inb);
		const int xt1 = wave_shr1(x, 0), vt1 = wave_shr1(v, r ? 2 : 0);                     // (r-1,t-1); boundary for t = 0 (:157-164)
		const int jj = r - lane;
		if (lane < n && jj >= 0 && jj < m) {
			int un, vn, xn, yn;
			const int d = dp_cell(xt1, vt1, u, y, cq, wref, un, vn, xn, yn);
			u = un; v = vn; x = xn; y = yn;
			dir[r * 64 + lane] = (uint8_t)d;
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	if (lane == 0) {
		int i = n - 1, j = m - 1, state = 0, k = 0;
		while (i >= 0 && j >= 0) rev[k++] = (uint8_t)dp_bt_step(dir[(i + j) * 64 + i], state, i, j);
		for (; i >= 0; --i) rev[k++] = 'D';
		for (; j >= 0; --j) rev[k++] = 'I';
		s_n[w] = k;
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	const int nops = s_n[w];
	uint8_t *op = ops + ops_off[job];
	for (int p = lane; p < nops; p += 64) op[p] = rev[nops - 1 - p];
	if (lane == 0) ops_len[job] = nops;
}

template <int T, int KMAX>
__global__ void __launch_bounds__(T) k_dp_wg(i32 n_jobs, const i32 *__restrict__ order, const i64 *__restrict__ diroff, const uint8_t *__restrict__ pool1,
                                              const i64 *__restrict__ off1, const i32 *__restrict__ len1, const uint8_t *__restrict__ pool2, const i64 *__restrict__ off2,
                                              const i32 *__restrict__ len2, uint8_t *dirbase, uint8_t *revbase, uint8_t *ops, const i64 *__restrict__ ops_off, i32 *ops_len, int mpad)
{
	extern __shared__ __attribute__((aligned(16))) int8_t lds[];        // the reference fragment as nt4 codes
	__shared__ uint8_t tile[64][64];
	__shared__ int xchg[2][T / 64][KMAX];                               // (x | v << 8) of each wave's last lane, per column set, ping-pong by diagonal parity
	__shared__ int s_i, s_j, s_state, s_k;
	if ((i32)blockIdx.x >= n_jobs) return;
	const i32 job = order[blockIdx.x];
	const int m = len1[job], n = len2[job], tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const uint8_t *s1 = pool1 + off1[job], *s2 = pool2 + off2[job];
	uint8_t *dir = dirbase + diroff[blockIdx.x];
	uint8_t *rev = revbase + ops_off[job], *op = ops + ops_off[job];
	if (m <= 0 || n <= 0) { if (tid == 0) ops_len[job] = 0; return; }
	// Thread tid owns target columns t = tid + k*T: their (u,v,x,y) state stays in registers for the
	// whole fill; the left neighbour's (x,v) of the previous diagonal arrives by a DPP wave shift, and
	// only the lane at a wave boundary goes through LDS.  One barrier per anti-diagonal.
	int8_t *C1 = lds;
	for (int t = tid; t < m; t += T) C1[t] = (int8_t)gsa_nt4(s1[t]);
	int u[KMAX], y[KMAX], x[KMAX], v[KMAX], cq[KMAX];
#pragma unroll
	for (int k = 0; k < KMAX; k++) { const int t = tid + k * T; u[k] = t ? 2 : 0; y[k] = 0; x[k] = 0; v[k] = 0; cq[k] = t < n ? gsa_nt4(s2[t]) : 4; }
	if (lane == 63) {
#pragma unroll
		for (int k = 0; k < KMAX; k++) { xchg[0][w][k] = 0; xchg[1][w][k] = 0; }
	}
	__syncthreads();
	const int nr = m + n - 1;
	i64 off = 0;
	for (int r = 0; r < nr; r++) {
		const int st = r - m + 1 > 0 ? r - m + 1 : 0, en = r < n - 1 ? r : n - 1;
		const int par = r & 1;
#pragma unroll
		for (int k = 0; k < KMAX; k++) {
			const int t = tid + k * T;
			if (k * T > en) break;                                       // uniform: no column of this set is on the diagonal yet / any more
			// (x,v) of column t-1 on diagonal r-1
			int fill;
			if (w > 0) fill = xchg[par][w - 1][k];
			else if (k > 0) fill = xchg[par][T / 64 - 1][k - 1];
			else fill = (r ? 2 : 0) << 8;                                // t = 0 boundary: x1 = 0, v1 = q (:157-164)
			const int packed = wave_shr1(x[k] | (v[k] << 8), fill);
			const int xt1 = packed & 0xff, vt1 = packed >> 8;
			const int jj = r - t;
			if (t < n && jj >= 0 && jj < m) {
				int un, vn, xn, yn;
				const int d = dp_cell(xt1, vt1, u[k], y[k], cq[k], C1[jj], un, vn, xn, yn);
				u[k] = un; v[k] = vn; x[k] = xn; y[k] = yn;
				dir[off + (t - st)] = (uint8_t)d;
			}
			if (lane == 63) xchg[par ^ 1][w][k] = x[k] | (v[k] << 8);
		}
		off += en - st + 1;
		__syncthreads();
	}
	// ---- traceback on 64 x 64 tiles: rows = diagonals R..R-63, columns = targets T0..T0-63 ----
	if (tid == 0) { s_i = n - 1; s_j = m - 1; s_state = 0; s_k = 0; }
	__threadfence_block();
	__syncthreads();
	while (s_i >= 0 && s_j >= 0) {
		const int R = s_i + s_j, T0 = s_i;
		for (int e = tid; e < 64 * 64; e += T) {
			const int rr = e >> 6, cc = e & 63;
			const int r = R - rr, t = T0 - cc;
			uint8_t val = 0;
			if (r >= 0 && t >= 0) {
				const int st = r - m + 1 > 0 ? r - m + 1 : 0, en = r < n - 1 ? r : n - 1;
				if (t >= st && t <= en) val = dir[dp_rowoff(r, m, n) + (t - st)];
			}
			tile[rr][cc] = val;
		}
		__syncthreads();
		if (tid == 0) {
			int i = s_i, j = s_j, state = s_state, k = s_k;
			while (i >= 0 && j >= 0) {
				const int rr = R - (i + j), cc = T0 - i;
				if (rr > 63 || cc > 63) break;
				rev[k++] = (uint8_t)dp_bt_step(tile[rr][cc], state, i, j);
			}
			s_i = i; s_j = j; s_state = state; s_k = k;
		}
		__syncthreads();
	}
	if (tid == 0) {
		int i = s_i, j = s_j, k = s_k;
		for (; i >= 0; --i) rev[k++] = 'D';
		for (; j >= 0; --j) rev[k++] = 'I';
		s_k = k; ops_len[job] = k;
	}
	__threadfence_block();
	__syncthreads();
	const int nops = s_k;
	for (int p = tid; p < nops; p += T) op[p] = rev[nops - 1 - p];
}

// All pointers are device pointers.  Large jobs are processed in batches so that the
// direction bytes of one batch fit the budget.
int run_ksw2_jobs(gsa_ctx *c, i32 n, const uint8_t *pool1, const i64 *off1, const i32 *len1,
                  const uint8_t *pool2, const i64 *off2, const i32 *len2, uint8_t *ops, const i64 *ops_off, i32 *ops_len)
{
	if (n <= 0) return GSA_OK;
	hipStream_t st = c->stream;
	std::vector<i32> h_len1((size_t)n), h_len2((size_t)n); std::vector<i64> h_ooff((size_t)n);
	GSA_CHECK(c, hipMemcpyAsync(h_len2.data(), len2, (size_t)n * 4, hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipMemcpyAsync(h_len1.data(), len1, (size_t)n * 4, hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipMemcpyAsync(h_ooff.data(), ops_off, (size_t)n * 8, hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipStreamSynchronize(st));
	std::vector<i32> small, mid, big;
	i64 ops_total = 0;
	for (i32 i = 0; i < n; i++) {
		const i64 m = h_len1[i], nn = h_len2[i];
		c->counters[4] += (u64)(m * nn); c->counters[6] += (u64)(m + nn);
		if (h_ooff[i] + m + nn > ops_total) ops_total = h_ooff[i] + m + nn;
		if (m <= 0 || nn <= 0) mid.push_back(i);
		else if (nn <= 64 && m + nn - 1 <= SMALL_ROWS) small.push_back(i);
		else if (nn > 512) big.push_back(i);
		else mid.push_back(i);
	}
	c->counters[5] += (u64)n;
	auto by_cells = [&](i32 a, i32 b) { const i64 ca = (i64)h_len1[a] * h_len2[a], cb = (i64)h_len1[b] * h_len2[b]; return ca != cb ? ca > cb : a < b; };
	std::sort(small.begin(), small.end(), by_cells); std::sort(mid.begin(), mid.end(), by_cells); std::sort(big.begin(), big.end(), by_cells);
	// big first (longest critical path), then mid, then the many small ones fill the machine around them
	i32 *d_order = dev_ensure<i32>(c, c->d_flag2, (size_t)n + 1);
	uint8_t *rev = dev_ensure<uint8_t>(c, c->d_i64a, (size_t)ops_total + 64);
	if (!d_order || !rev) return GSA_ERR_NOMEM;
	std::vector<i32> order; order.reserve((size_t)n);
	order.insert(order.end(), big.begin(), big.end()); order.insert(order.end(), mid.begin(), mid.end()); order.insert(order.end(), small.begin(), small.end());
	GSA_CHECK(c, hipMemcpyAsync(d_order, order.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
	// One direction buffer for both workgroup classes (offsets continue), so the three kernels can run
	// CONCURRENTLY on three streams: the few huge jobs set the critical path, everything else fills
	// the machine around them.  If the direction bytes exceed the budget the classes are batched.
	const i64 budget = 12ll << 30;
	std::vector<i32> wg_jobs(big); wg_jobs.insert(wg_jobs.end(), mid.begin(), mid.end());
	std::vector<i64> h_diroff(wg_jobs.size() + 1, 0);
	hipEvent_t ev_fork = c->ev[10], ev_j1 = c->ev[11], ev_j2 = c->ev[12];
	size_t first = 0;
	bool small_done = small.empty();
	while (first < wg_jobs.size() || !small_done) {
		size_t last = first; i64 bytes = 0; int nmax_big = 1, nmax_mid = 1, mmax_big = 1, mmax_mid = 1; size_t nbig = 0;
		while (last < wg_jobs.size()) {
			const i32 jb = wg_jobs[last]; const i64 cells = (i64)h_len1[jb] * h_len2[jb];
			if (last > first && bytes + cells > budget) break;
			h_diroff[last] = bytes; bytes += cells;
			if (last < big.size()) { nbig++; if (h_len2[jb] > nmax_big) nmax_big = h_len2[jb]; if (h_len1[jb] > mmax_big) mmax_big = h_len1[jb]; }
			else { if (h_len2[jb] > nmax_mid) nmax_mid = h_len2[jb]; if (h_len1[jb] > mmax_mid) mmax_mid = h_len1[jb]; }
			last++;
		}
		uint8_t *dir = dev_ensure<uint8_t>(c, c->d_scan2, (size_t)bytes + 64);
		i64 *d_diroff = dev_ensure<i64>(c, c->j_cells, wg_jobs.size() + 1);
		if (!dir || !d_diroff) return GSA_ERR_NOMEM;
		if (last > first) GSA_CHECK(c, hipMemcpyAsync(d_diroff + first, h_diroff.data() + first, (last - first) * 8, hipMemcpyHostToDevice, st));
		const int npad_big = (nmax_big + 63) & ~63, npad_mid = (nmax_mid + 63) & ~63, mpad_big = (mmax_big + 63) & ~63, mpad_mid = (mmax_mid + 63) & ~63;
		if (nmax_big > 5 * 1024 || mpad_big > 140 * 1024 || mpad_mid > 140 * 1024) return gsa_fail(c, GSA_ERR_LIMIT, "DP fragment too long (query side > 5120 or reference side > 143360 bases)");
		(void)npad_big; (void)npad_mid;
		GSA_CHECK(c, hipEventRecord(ev_fork, st));
		const size_t nmid = (last - first) - nbig;
		if (nbig) hipLaunchKernelGGL((k_dp_wg<1024, 5>), dim3((unsigned)nbig), dim3(1024), (size_t)mpad_big, st, (i32)nbig, d_order + first, d_diroff + first, pool1, off1, len1, pool2, off2, len2, dir, rev, ops, ops_off, ops_len, mpad_big);
		if (nmid) {
			GSA_CHECK(c, hipStreamWaitEvent(c->stream_aux[0], ev_fork, 0));
			hipLaunchKernelGGL((k_dp_wg<256, 2>), dim3((unsigned)nmid), dim3(256), (size_t)mpad_mid, c->stream_aux[0], (i32)nmid, d_order + first + nbig, d_diroff + first + nbig, pool1, off1, len1, pool2, off2, len2, dir, rev, ops, ops_off, ops_len, mpad_mid);
			GSA_CHECK(c, hipEventRecord(ev_j1, c->stream_aux[0]));
			GSA_CHECK(c, hipStreamWaitEvent(st, ev_j1, 0));
		}
		if (!small_done) {
			const unsigned nb = (unsigned)((small.size() + SMALL_WAVES - 1) / SMALL_WAVES);
			GSA_CHECK(c, hipStreamWaitEvent(c->stream_aux[1], ev_fork, 0));
			hipLaunchKernelGGL(k_dp_small, dim3(nb), dim3(64 * SMALL_WAVES), 0, c->stream_aux[1], (i32)small.size(), d_order + wg_jobs.size(), pool1, off1, len1, pool2, off2, len2, ops, ops_off, ops_len);
			GSA_CHECK(c, hipEventRecord(ev_j2, c->stream_aux[1]));
			GSA_CHECK(c, hipStreamWaitEvent(st, ev_j2, 0));
			small_done = true;
		}
		GSA_CHECK(c, hipGetLastError());
		first = last;
		if (first < wg_jobs.size()) GSA_CHECK(c, hipStreamSynchronize(st));       // the direction buffer is reused by the next batch
	}
	GSA_CHECK(c, hipStreamSynchronize(st));       // staging vectors (order, diroff) go out of scope
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
