// gsalign_amd/csrc/gsa_dp.h -- device-side global affine-gap DP with traceback (a13).
//
// Restates ksw_extz2_sse + ksw_backtrack as called by ksw2_alignment
// (reference src/ksw2_alignment.cpp:25-68,70-249,251-273): m=5, q=2, e=1, w=-1,
// i.e. full matrix, match +1, mismatch -1, N 0, gap k costs 2+k.  Suzuki-Kasahara
// difference recurrence, one byte of direction flags per cell, exact tie rules
// (diagonal beats a, a beats b), cell-exact (no SIMD padding): SURVEY.md App. A.6.
//
// Mapping: ONE 64-lane wavefront per alignment.  Anti-diagonal r = i + j; the
// cells of a diagonal are independent, lane l takes cell t = st + 64c + l.  The
// four 8-bit state arrays u,v,x,y (indexed by t) live in LDS; the direction bytes
// go to HBM diagonal-major (coalesced 64-byte stores).  Chunks of a diagonal are
// processed from high t to low t so that the (t-1) neighbour is still the value
// of diagonal r-1 when it is read.  No MFMA: this is 8-bit max/add with a
// data-dependent traceback.
#ifndef GSA_DP_H
#define GSA_DP_H
#include "gsa_fm.h"

// number of cells on diagonals < r of an m x n problem (m = |s1| ksw query, n = |s2| ksw target)
__device__ __forceinline__ i64 dp_rowoff(i64 r, i64 m, i64 n)
{
	const i64 a = m < n ? m : n, b = m < n ? n : m;
	if (r <= a) return r * (r + 1) / 2;
	if (r <= b) return a * (a + 1) / 2 + (r - a) * a;
	return a * (a + 1) / 2 + (b - a) * a + (r - b) * (m + n - 1) - (b + r - 1) * (r - b) / 2;
}

// Fill the direction matrix.  s1/s2 are raw ASCII; lds holds 4*npad bytes where
// npad >= n rounded up to 64.  All 64 lanes of the wave must call this.
__device__ __forceinline__ void dp_fill(const uint8_t *__restrict__ s1, int m, const uint8_t *__restrict__ s2, int n,
                                        int8_t *lds, int npad, uint8_t *__restrict__ dir)
{
	const int lane = threadIdx.x & 63;
	int8_t *U = lds, *V = lds + npad, *X = lds + 2 * npad, *Y = lds + 3 * npad;
	const int nr = m + n - 1;
	i64 off = 0;
	for (int r = 0; r < nr; r++) {
		const int st = r - m + 1 > 0 ? r - m + 1 : 0, en = r < n - 1 ? r : n - 1;
		const int len = en - st + 1;
		if (lane == 0 && en >= r) { Y[r] = 0; U[r] = r ? 2 : 0; }     // boundary (r-1, t=r): ksw2_alignment.cpp:165
		__syncthreads();   // single-wave workgroup: orders LDS traffic between lanes
		for (int c = (len - 1) >> 6; c >= 0; c--) {
			const int t = st + (c << 6) + lane;
			if (t <= en) {
				int8_t xt1, vt1;
				if (t > 0) { xt1 = X[t - 1]; vt1 = V[t - 1]; } else { xt1 = 0; vt1 = r ? 2 : 0; }   // :157-164
				const int8_t ut = U[t], yt = Y[t];
				const int a_ = gsa_nt4(s2[t]), b_ = gsa_nt4(s1[r - t]);
				const int sc = (a_ == 4 || b_ == 4) ? 0 : (a_ == b_ ? 1 : -1);
				int z = sc + 6;
				int a = xt1 + vt1, b = yt + ut;
				int d = a > z ? 1 : 0; z = z > a ? z : a;
				if (b > z) d = 2;
				z = z > b ? z : b;
				z = z < 7 ? z : 7;
				const int un = z - vt1, vn = z - ut;
				z -= 2; a -= z; b -= z;
				if (a > 0) d |= 0x08; else a = 0;
				if (b > 0) d |= 0x10; else b = 0;
				U[t] = (int8_t)un; V[t] = (int8_t)vn; X[t] = (int8_t)a; Y[t] = (int8_t)b;
				dir[off + (t - st)] = (uint8_t)d;
			}
			__syncthreads();   // single-wave workgroup: orders LDS traffic between lanes
		}
		off += len;
	}
}

// ksw_backtrack by one lane: writes the op string REVERSED into rev[0..), returns its length
__device__ __forceinline__ int dp_backtrack(const uint8_t *__restrict__ dir, int m, int n, uint8_t *rev)
{
	int i = n - 1, j = m - 1, state = 0, k = 0;
	while (i >= 0 && j >= 0) {
		const int r = i + j;
		const int st = r - m + 1 > 0 ? r - m + 1 : 0;
		const u32 tmp = dir[dp_rowoff(r, m, n) + (i - st)];
		if (state == 0) state = tmp & 7;
		else if (!((tmp >> (state + 2)) & 1)) state = 0;
		if (state == 0) state = tmp & 7;
		if (state == 0) { rev[k++] = 'M'; --i; --j; }
		else if (state == 1 || state == 3) { rev[k++] = 'D'; --i; }
		else { rev[k++] = 'I'; --j; }
	}
	for (; i >= 0; --i) rev[k++] = 'D';
	for (; j >= 0; --j) rev[k++] = 'I';
	return k;
}

#endif
