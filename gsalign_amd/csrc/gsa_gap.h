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
		// CheckFragPairMismatch: positions where the QUERY is ambiguous are skipped.  32 positions per round,
		// all 64 loads in flight before the first compare (one thread does this inside a fused pass, so the
		// longest gap's load latency is the pass's duration); the count only matters while <= GSA_MAX_MISMATCH.
		const uint8_t *qs = query + qpos, *rs = ref + rpos;
		if (qg <= 8) {
			// the usual case, a SNP or two: a handful of loads, not a 32-wide round
			uint8_t bq[8], br[8];
#pragma unroll
			for (int b = 0; b < 8; b++) { if (b < qg) { bq[b] = qs[b]; br[b] = rs[b]; } else { bq[b] = 'N'; br[b] = 'N'; } }
#pragma unroll
			for (int b = 0; b < 8; b++) { const int a = gsa_nt4(bq[b]); if (b < qg && a != 4 && a != gsa_nt4(br[b])) mism++; }
		} else {
			for (i32 x = 0; x < qg && mism <= GSA_MAX_MISMATCH; x += 32) {
				uint8_t bq[32], br[32];
#pragma unroll
				for (int b = 0; b < 32; b++) { const i32 p = x + b < qg ? x + b : qg - 1; bq[b] = qs[p]; br[b] = rs[p]; }
#pragma unroll
				for (int b = 0; b < 32; b++) { const int a = gsa_nt4(bq[b]); if (x + b < qg && a != 4 && a != gsa_nt4(br[b])) mism++; }
			}
		}
		if (mism <= GSA_MAX_MISMATCH) return FT_EQ;
	}
	return FT_DP;
}

// does a DP job (m reference bases x n query bases) need the striped kernel?
__device__ __forceinline__ bool dp_is_large(i32 m, i32 n) { return !(n <= 64 && m + n - 1 <= SMALL_ROWS); }

#endif
