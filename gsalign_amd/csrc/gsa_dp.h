// gsalign_amd/csrc/gsa_dp.h -- device-side global affine-gap DP with traceback (a13).
//
// Restates ksw_extz2_sse + ksw_backtrack as called by ksw2_alignment
// (reference src/ksw2_alignment.cpp:25-68,70-249,251-273): m=5, q=2, e=1, w=-1,
// i.e. full matrix, match +1, mismatch -1, N 0, gap k costs 2+k.  Suzuki-Kasahara
// difference recurrence, one byte of direction flags per cell, exact tie rules
// (diagonal beats a, a beats b), cell-exact (no SIMD padding): SURVEY.md App. A.6.
// s1 = reference fragment (ksw "query", index j, length m), s2 = query fragment
// (ksw "target", index i = t, length n); anti-diagonal r = i + j.
// No MFMA: this is 8-bit max/add with a data-dependent traceback.
#ifndef GSA_DP_H
#define GSA_DP_H
#include "gsa_fm.h"

// one cell of the recurrence (ksw2_alignment.cpp:74-95,184-199).
// in : xt1,vt1 = x,v of (r-1,t-1); ut,yt = u,y of (r-1,t); a_,b_ = codes of s2[t], s1[r-t]
// out: new u,v,x,y of (r,t) and the direction byte
__device__ __forceinline__ int dp_cell(int xt1, int vt1, int ut, int yt, int a_, int b_, int &un, int &vn, int &xn, int &yn)
{
	const int sc = (a_ == 4 || b_ == 4) ? 0 : (a_ == b_ ? 1 : -1);
	int z = sc + 6;
	int a = xt1 + vt1, b = yt + ut;
	int d = a > z ? 1 : 0; z = z > a ? z : a;
	if (b > z) d = 2;
	z = z > b ? z : b;
	z = z < 7 ? z : 7;
	un = z - vt1; vn = z - ut;
	z -= 2; a -= z; b -= z;
	if (a > 0) d |= 0x08; else a = 0;
	if (b > 0) d |= 0x10; else b = 0;
	xn = a; yn = b;
	return d;
}

// the traceback automaton (ksw_backtrack :25-68); with the full band the force_state paths never fire
__device__ __forceinline__ int dp_bt_step(u32 tmp, int &state, int &i, int &j)
{
	if (state == 0) state = tmp & 7;
	else if (!((tmp >> (state + 2)) & 1)) state = 0;
	if (state == 0) state = tmp & 7;
	if (state == 0) { --i; --j; return 'M'; }
	if (state == 1 || state == 3) { --i; return 'D'; }
	--j; return 'I';
}

// number of cells on diagonals < r of an m x n problem (diagonal-major direction matrix)
__device__ __forceinline__ i64 dp_rowoff(i64 r, i64 m, i64 n)
{
	const i64 a = m < n ? m : n, b = m < n ? n : m;
	if (r <= a) return r * (r + 1) / 2;
	if (r <= b) return a * (a + 1) / 2 + (r - a) * a;
	return a * (a + 1) / 2 + (b - a) * a + (r - b) * (m + n - 1) - (b + r - 1) * (r - b) / 2;
}

// shift a value one lane up across the whole 64-lane wave (lane t receives lane t-1's value;
// lane 0 receives `fill`): DPP wave_shr:1, a VALU-rate move instead of an LDS round trip
__device__ __forceinline__ int wave_shr1(int v, int fill)
{
	return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);
}

#endif
