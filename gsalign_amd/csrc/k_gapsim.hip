// gsalign_amd/csrc/k_gapsim.hip -- gap similarity test (a9).
//
// Replaces CalGapSimilarity / CreateKmerVecFromReadSeq / CreateKmerID
// (reference src/KmerAnalysis.cpp:10-17,32-76,78-121).  One 256-thread workgroup per gap:
//   * flanks on one diagonal: count positions with equal codes (or an N on
//     either side); similar if count >= q_len*0.5;
//   * otherwise (both windows <= 5000): multiset intersection of the 5-mer ids of
//     the two windows, similar if |intersection| > (q_len+r_len)*0.1.
// The sorted-vector set_intersection of the reference equals sum(min(h1,h2))
// over two histograms, which live in LDS.  The reference's id arithmetic is kept
// bit-exact, including what a code 4 (n, IUPAC) does to the rolling id and the
// literal-'N' rescan quirk, which is replayed sequentially
// by one lane because it is a state machine.
#include "gsa_ctx.h"
#include "gsa_fm.h"

#define KBINS 1376          // max id: 4*(256+64+16+4+1) = 1364
#define GS_T 256            // threads per gap (the longest window, 5000 bases, sets the kernel's duration)

__device__ __forceinline__ u32 kmer_id_direct(const uint8_t *s, int pos)       // CreateKmerID, no masking
{
	u32 id = 0;
	for (int i = pos; i < pos + 5; i++) id = (id << 2) + gsa_nt4(s[i]);
	return id;
}

// one window -> histogram (all GS_T threads call)
__device__ void kmer_hist(const uint8_t *__restrict__ s, int len, u32 *hist, int lane, int *s_flag)
{
	// does the window contain a literal 'N'?
	if (lane == 0) *s_flag = 0;
	__syncthreads();
	int hasN = 0;
	// (eight positions per thread and pass: one 16-byte window read; both sequence buffers are padded)
	for (int p = lane * 8; p < len; p += GS_T * 8) {
		unsigned long long w0; __builtin_memcpy(&w0, s + p, 8);
		for (int b = 0; b < 8; b++) hasN |= (p + b < len && (uint8_t)(w0 >> (8 * b)) == 'N');
	}
	if (hasN) *s_flag = 1;
	__syncthreads();
	hasN = *s_flag;
	__syncthreads();
	if (!hasN) {
		// wid_0 = direct id; wid_p (p>=1) = ((wid_{p-1} & 0xFF) << 2) + v[p+4], which depends on v[p..p+4] only
		for (int p0 = lane * 8; p0 + 5 <= len; p0 += GS_T * 8) {
			unsigned long long w0, w1; __builtin_memcpy(&w0, s + p0, 8); __builtin_memcpy(&w1, s + p0 + 8, 8);
			u32 cd[12];
#pragma unroll
			for (int b = 0; b < 12; b++) cd[b] = (u32)gsa_nt4((uint8_t)((b < 8 ? w0 >> (8 * b) : w1 >> (8 * (b - 8)))));
#pragma unroll
			for (int b = 0; b < 8; b++) {
				const int p = p0 + b;
				if (p + 5 > len) break;
				u32 id;
				if (p == 0) { id = 0; for (int i = 0; i < 5; i++) id = (id << 2) + cd[i]; }
				else {
					u32 t = 0;
					for (int i = 0; i < 4; i++) t = (t << 2) + cd[b + i];
					id = ((t & 0xFF) << 2) + cd[b + 4];
				}
				atomicAdd(&hist[id], 1u);
			}
		}
	} else if (lane == 0) {
		u32 wid, count = 0, head = 0, tail = 0;
		while (count < 5 && tail < (u32)len) { if (s[tail++] != 'N') count++; else count = 0; }
		if (count == 5) {
			wid = kmer_id_direct(s, (short)head); hist[wid]++;
			for (head += 1; tail < (u32)len; head++, tail++) {
				if (s[tail] != 'N') { wid = ((wid & 0xFF) << 2) + gsa_nt4(s[tail]); hist[wid]++; }
				else {
					count = 0; tail++;
					while (count < 5 && tail < (u32)len) { if (s[tail++] != 'N') count++; else count = 0; }
					if (count == 5) { wid = kmer_id_direct(s, (short)head); hist[wid]++; }
					else break;
				}
			}
		}
	}
}

__global__ void __launch_bounds__(GS_T) k_gapsim(DevIndex di, const uint8_t *__restrict__ query, i32 n_host, const i32 *__restrict__ d_n, const i32 *__restrict__ q1a, const i32 *__restrict__ q2a,
                                                const i64 *__restrict__ r1a, const i64 *__restrict__ r2a, i32 *res,
                                                const i32 *__restrict__ jseed, i32 *cut4)
{
	__shared__ u32 h1[KBINS], h2[KBINS];
	__shared__ int s_acc, s_flag;
	const int lane = threadIdx.x;
	const i32 n = d_n ? *d_n : n_host;                  // (job count on the device when the host has not looked yet)
	for (int job = blockIdx.x; job < n; job += gridDim.x) {
		const i32 q1 = q1a[job], q2 = q2a[job]; const i64 r1 = r1a[job], r2 = r2a[job];
		const int q_len = q2 - q1, r_len = (int)(r2 - r1);
		bool sim = false;
		if (lane == 0) s_acc = 0;
		__syncthreads();
		if (r1 - q1 == r2 - q2) {
			int idy = 0;
			for (int p = lane; p < q_len; p += GS_T) {
				const int a = gsa_nt4(di.ref[r1 + p]), b = gsa_nt4(query[q1 + p]);
				idy += (a == b || a == 4 || b == 4);
			}
			for (int o = 32; o; o >>= 1) idy += __shfl_xor(idy, o);
			if ((lane & 63) == 0 && idy) atomicAdd(&s_acc, idy);
			__syncthreads();
			if ((double)s_acc >= q_len * 0.5) sim = true;
			__syncthreads();
			if (lane == 0) s_acc = 0;
			__syncthreads();
		}
		if (!sim && q_len <= GSA_MAX_SEED_GAP && r_len <= GSA_MAX_SEED_GAP) {
			for (int b = lane; b < KBINS; b += GS_T) { h1[b] = 0; h2[b] = 0; }
			__syncthreads();
			kmer_hist(query + q1, q_len, h1, lane, &s_flag);
			kmer_hist(di.ref + r1, r_len, h2, lane, &s_flag);
			__syncthreads();
			int common = 0;
			for (int b = lane; b < KBINS; b += GS_T) common += (int)(h1[b] < h2[b] ? h1[b] : h2[b]);
			for (int o = 32; o; o >>= 1) common += __shfl_xor(common, o);
			if ((lane & 63) == 0 && common) atomicAdd(&s_acc, common);
			__syncthreads();
			if ((double)s_acc > (q_len + r_len) * 0.1) sim = true;
		}
		// (stage 4 applies the verdict on the spot: a gap whose sides are not similar cuts the block in front of seed jseed[job],
		//  CheckGapsBetweenSeeds :120-156 -- was a kernel of its own behind this one)
		if (lane == 0) { res[job] = sim ? 1 : 0; if (cut4 && !sim) cut4[jseed[job]] = 1; }
		__syncthreads();
	}
}

int run_gapsim_jobs(gsa_ctx *c, i32 n, const i32 *d_n, const i32 *d_q1, const i32 *d_q2, const i64 *d_r1, const i64 *d_r2, i32 *d_res, const i32 *d_jseed, i32 *d_cut4)
{
	// n = job count, or with d_n != nullptr an upper bound (the grid is capped, workgroups loop over the jobs)
	if (n <= 0) return GSA_OK;
	const i32 grid = d_n ? (n < 1024 ? n : 1024) : n;
	hipLaunchKernelGGL(k_gapsim, dim3(grid), dim3(GS_T), 0, c->stream, c->di, c->q_dev, n, d_n, d_q1, d_q2, d_r1, d_r2, d_res, d_jseed, d_cut4);
	GSA_CHECK(c, hipGetLastError());
	return GSA_OK;
}

extern "C" int gsa_gap_similarity_batch(gsa_ctx *c, int32_t n, const int32_t *q1, const int32_t *q2, const int64_t *r1, const int64_t *r2, int32_t *similar)
{
	if (!c || n < 0) return GSA_ERR_ARG;
	if (c->qlen <= 0) return gsa_fail(c, GSA_ERR_STATE, "gsa_set_query first");
	if (n == 0) return GSA_OK;
	for (int i = 0; i < n; i++)
		if (q1[i] < 0 || q2[i] < q1[i] || q2[i] > c->qlen || r1[i] < 0 || r2[i] < r1[i] || r2[i] > 2 * c->G) return gsa_fail(c, GSA_ERR_ARG, "gap window out of range");
	hipStream_t st = c->stream;
	i32 *dq1 = dev_ensure<i32>(c, c->leaf[0], (size_t)n), *dq2 = dev_ensure<i32>(c, c->leaf[1], (size_t)n), *dres = dev_ensure<i32>(c, c->leaf[2], (size_t)n);
	i64 *dr1 = dev_ensure<i64>(c, c->leaf[3], (size_t)n), *dr2 = dev_ensure<i64>(c, c->leaf[4], (size_t)n);
	if (!dq1 || !dq2 || !dres || !dr1 || !dr2) return GSA_ERR_NOMEM;
	GSA_CHECK(c, hipMemcpyAsync(dq1, q1, n * 4, hipMemcpyHostToDevice, st)); GSA_CHECK(c, hipMemcpyAsync(dq2, q2, n * 4, hipMemcpyHostToDevice, st));
	GSA_CHECK(c, hipMemcpyAsync(dr1, r1, n * 8, hipMemcpyHostToDevice, st)); GSA_CHECK(c, hipMemcpyAsync(dr2, r2, n * 8, hipMemcpyHostToDevice, st));
	int rc = run_gapsim_jobs(c, n, nullptr, dq1, dq2, dr1, dr2, dres, nullptr, nullptr);
	if (rc == GSA_OK) { GSA_CHECK(c, hipMemcpyAsync(similar, dres, n * 4, hipMemcpyDeviceToHost, st)); GSA_CHECK(c, hipStreamSynchronize(st)); }
	return rc;
}
