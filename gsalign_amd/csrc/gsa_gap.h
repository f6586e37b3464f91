// gsalign_amd/csrc/gsa_gap.h -- what a gap between two consecutive seeds of a block is (shared by the
// record pass of stage 6 and the early launch of the large DP gaps from the leaf table).
#ifndef GSA_GAP_H
#define GSA_GAP_H
#include "gsa_fm.h"

enum { FT_SEED = 0, FT_DEL = 1, FT_INS = 2, FT_EQ = 3, FT_DP = 4 };
#define SMALL_ROWS 128      // k_dp_small: n <= 64 target columns and at most this many anti-diagonals

// gap of qg query and rg reference bases (both already clamped at 0): GenerateFragAlignment :311-342
__device__ __forceinline__ i32 classify_gap(const uint8_t *__restrict__ query, const uint8_t *__restrict__ ref, i32 qpos, i64 rpos, i32 qg, i32 rg, i32 &mism)
{
	mism = 0;
	if (qg == 0) return FT_DEL;
	if (rg == 0) return FT_INS;
	if (qg == rg) {
		// CheckFragPairMismatch: positions where the QUERY is ambiguous are skipped.  Eight bases per pair of
		// loads (both buffers are padded): the count only matters while it stays <= GSA_MAX_MISMATCH.
		const uint8_t *qs = query + qpos, *rs = ref + rpos;
		for (i32 x = 0; x < qg && mism <= GSA_MAX_MISMATCH; x += 8) {
			unsigned long long wq, wr;
			__builtin_memcpy(&wq, qs + x, 8); __builtin_memcpy(&wr, rs + x, 8);
			const i32 lim = qg - x < 8 ? qg - x : 8;
#pragma unroll
			for (int b = 0; b < 8; b++) {
				const int a = gsa_nt4((uint8_t)(wq >> (8 * b))), r = gsa_nt4((uint8_t)(wr >> (8 * b)));
				if (b < lim && a != 4 && a != r) mism++;
			}
		}
		if (mism <= GSA_MAX_MISMATCH) return FT_EQ;
	}
	return FT_DP;
}

// does a DP job (m reference bases x n query bases) need the striped kernel?
__device__ __forceinline__ bool dp_is_large(i32 m, i32 n) { return !(n <= 64 && m + n - 1 <= SMALL_ROWS); }

#endif
