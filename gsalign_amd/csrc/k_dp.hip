// gsalign_amd/csrc/k_dp.hip -- batched gap-closing DP (a13) and the public leaf
// operator gsa_ksw2_batch.  Cell recurrence and traceback automaton: gsa_dp.h.
//
// Two kernels, chosen per job by size (jobs are launched largest first, the two kernels run concurrently):
//  * k_dp_small  n <= 64 and m+n-1 <= 128 (the bulk of the jobs: median 11 x 11).
//    One wavefront per alignment, lane t owns target column t.  The (u,v,x,y) state
//    lives in REGISTERS; the left neighbour and the reference base travel one lane
//    up per anti-diagonal with DPP wave shifts (systolic array); direction bytes
//    and the traceback stay in LDS.  No barrier, no global traffic but the result.
//  * k_dp_stripe  everything else (up to 5000 x 5000): the target columns are cut
//    into 64-wide stripes, one wavefront per PAIR of stripes (packed 16-bit VALU),
//    boundary columns handed over through LDS / HBM; see the comment at the kernel.
#include <algorithm>
#include <cstring>
#include "gsa_ctx.h"
#include "gsa_dp.h"
#include "gsa_scan.h"
#include "gsa_gap.h"

#define SMALL_WAVES 4

__global__ void __launch_bounds__(64 * SMALL_WAVES) k_dp_small(i32 n_jobs, const i32 *__restrict__ order, const uint8_t *__restrict__ pool1, const i64 *__restrict__ off1,
                                                                const i32 *__restrict__ len1, const uint8_t *__restrict__ pool2, const i64 *__restrict__ off2,
                                                                const i32 *__restrict__ len2, uint8_t *ops, const i64 *__restrict__ ops_off, i32 *ops_len,
                                                                const i32 *__restrict__ jfrag, gsa_frag *frag)
{
	// direction flags as NIBBLES, two anti-diagonals per byte (only bits 0-1 and 3-4 of ksw2's flag byte are ever set): half
	// the LDS per alignment -- LDS is what limits how many of these waves a CU holds -- and half the LDS stores
	__shared__ uint8_t s_dir[SMALL_WAVES][(SMALL_ROWS / 2) * 64];
	__shared__ uint8_t s_rev[SMALL_WAVES][SMALL_ROWS + 64];
	__shared__ int s_n[SMALL_WAVES];
	const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const i32 slot = blockIdx.x * SMALL_WAVES + w;
	if (slot >= n_jobs) return;
	// (the job and its two lengths are the same in all lanes: said so, they live in scalar registers, the loop over the anti-diagonals is
	//  uniform and the reference base that enters at lane 0 is READ from its lane (v_readlane) instead of fetched through the LDS crossbar
	//  (ds_bpermute): that fetch sat in front of every diagonal's dependent chain)
	const i32 job = __builtin_amdgcn_readfirstlane(order[slot]);
	const int m = __builtin_amdgcn_readfirstlane(len1[job]), n = __builtin_amdgcn_readfirstlane(len2[job]);
	const uint8_t *s1 = pool1 + off1[job], *s2 = pool2 + off2[job];
	uint8_t *dir = s_dir[w], *rev = s_rev[w];
	const int cq = lane < n ? gsa_nt4(s2[lane]) : 4;
	// reference base for lane t at diagonal r is s1[r - t]: it enters at lane 0 and moves one lane up per diagonal
	const int c1a = lane < m ? gsa_nt4(s1[lane]) : 4, c1b = lane + 64 < m ? gsa_nt4(s1[lane + 64]) : 4;
	int u = lane ? 2 : 0, v = 0, x = 0, y = 0, wref = 4, dacc = 0;
	const int nr = m + n - 1;
	for (int r = 0; r < nr; r++) {
		const int inb = r < m ? (r < 64 ? __builtin_amdgcn_readlane(c1a, r) : __builtin_amdgcn_readlane(c1b, r - 64)) : 4;      // s1[r] broadcast
		wref = wave_shr1(wref, inb);
		const int xt1 = wave_shr1(x, 0), vt1 = wave_shr1(v, r ? 2 : 0);                     // (r-1,t-1); boundary for t = 0 (:157-164)
		const int jj = r - lane;
		int d = 0;
		if (lane < n && jj >= 0 && jj < m) {
			int un, vn, xn, yn;
			d = dp_cell(xt1, vt1, u, y, cq, wref, un, vn, xn, yn);
			u = un; v = vn; x = xn; y = yn;
		}
		const int nib = (d & 3) | ((d & 0x18) >> 1);
		// (every lane stores, cells outside the matrix are never read)
		if (r & 1) dir[(r >> 1) * 64 + lane] = (uint8_t)(dacc | (nib << 4)); else dacc = nib;
	}
	if (nr & 1) dir[(nr >> 1) * 64 + lane] = (uint8_t)dacc;
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	if (lane == 0) {
		int i = n - 1, j = m - 1, state = 0, k = 0;
		while (i >= 0 && j >= 0) {
			const u32 nb = ((u32)dir[((i + j) >> 1) * 64 + i] >> (((i + j) & 1) << 2)) & 15u;
			const u32 tmp = (nb & 3u) | ((nb & 0xCu) << 1);
			int ns = state;                                                 // ksw_backtrack automaton (:38-52), branch-free
			if (ns != 0 && !((tmp >> (ns + 2)) & 1)) ns = 0;
			if (ns == 0) ns = (int)(tmp & 7);
			state = ns;
			const int isM = ns == 0 ? 1 : 0, isD = (ns == 1 || ns == 3) ? 1 : 0;
			rev[k++] = (uint8_t)(isM ? 'M' : (isD ? 'D' : 'I'));
			i -= isM | isD; j -= isM | (1 - isD);
		}
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
	if (lane == 0) { ops_len[job] = nops; if (frag) frag[jfrag[job]].aln_len = nops; }      // (the job's record is final with this)
}

// ---------------------------------------------------------------------------
// k_dp_tiny: the bulk of the jobs is a handful of bases on either side (median 11 x 11), so FOUR
// alignments share a wavefront: n <= 16 target columns each, one per 16-lane DPP row (row_shr:1 never
// crosses a row), at most TINY_ROWS anti-diagonals.  Same systolic scheme as k_dp_small; the four
// tracebacks run on four lanes at once.
// ---------------------------------------------------------------------------
#define TINY_ROWS 64
#define TINY_WAVES 4
__device__ __forceinline__ int row_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x111, 0xf, 0xf, false); }   // lane t of a row <- lane t-1, row start <- fill

__global__ void __launch_bounds__(64 * TINY_WAVES) k_dp_tiny(i32 n_jobs, const i32 *__restrict__ order, const uint8_t *__restrict__ pool1, const i64 *__restrict__ off1,
                                                              const i32 *__restrict__ len1, const uint8_t *__restrict__ pool2, const i64 *__restrict__ off2,
                                                              const i32 *__restrict__ len2, uint8_t *ops, const i64 *__restrict__ ops_off, i32 *ops_len,
                                                              const i32 *__restrict__ jfrag, gsa_frag *frag)
{
	__shared__ uint8_t s_dir[TINY_WAVES][TINY_ROWS * 64];
	__shared__ uint8_t s_ref[TINY_WAVES][4][TINY_ROWS];         // nt4 codes of the four reference fragments
	__shared__ uint8_t s_rev[TINY_WAVES][4][TINY_ROWS + 32];
	__shared__ int s_n[TINY_WAVES][4];
	const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, tl = lane & 15;
	const i32 slot = (blockIdx.x * TINY_WAVES + w) * 4 + q;
	if ((blockIdx.x * TINY_WAVES + w) * 4 >= n_jobs) return;
	const bool have = slot < n_jobs;
	const i32 job = have ? order[slot] : 0;
	const int m = have ? len1[job] : 0, n = have ? len2[job] : 0;
	const uint8_t *s1 = pool1 + off1[job], *s2 = pool2 + off2[job];
	uint8_t *dir = s_dir[w];
	for (int k = tl; k < TINY_ROWS; k += 16) s_ref[w][q][k] = (uint8_t)(k < m ? gsa_nt4(s1[k]) : 4);
	const int cq = tl < n ? gsa_nt4(s2[tl]) : 4;
	int u = tl ? 2 : 0, v = 0, x = 0, y = 0, wref = 4;
	// the longest of the four decides how many diagonals the wavefront runs
	int nr = have ? m + n - 1 : 0;
	nr = max(max(__builtin_amdgcn_readlane(nr, 0), __builtin_amdgcn_readlane(nr, 16)), max(__builtin_amdgcn_readlane(nr, 32), __builtin_amdgcn_readlane(nr, 48)));
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	int inb = s_ref[w][q][0];
	for (int r = 0; r < nr; r++) {
		const int inb_next = s_ref[w][q][r + 1 < TINY_ROWS ? r + 1 : TINY_ROWS - 1];      // (one diagonal ahead: off the recurrence chain)
		wref = row_shr1(wref, r < m ? inb : 4);
		const int xt1 = row_shr1(x, 0), vt1 = row_shr1(v, r ? 2 : 0);                     // (r-1,t-1); boundary for t = 0 (:157-164)
		const int jj = r - tl;
		if (tl < n && jj >= 0 && jj < m) {
			int un, vn, xn, yn;
			const int d = dp_cell(xt1, vt1, u, y, cq, wref, un, vn, xn, yn);
			u = un; v = vn; x = xn; y = yn;
			dir[r * 64 + lane] = (uint8_t)d;
		}
		inb = inb_next;
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	if (tl == 0 && have) {
		uint8_t *rev = s_rev[w][q];
		int i = n - 1, j = m - 1, state = 0, k = 0;
		while (i >= 0 && j >= 0) {
			const u32 tmp = dir[(i + j) * 64 + (q << 4) + i];
			int ns = state;                                                 // ksw_backtrack automaton (:38-52), branch-free
			if (ns != 0 && !((tmp >> (ns + 2)) & 1)) ns = 0;
			if (ns == 0) ns = (int)(tmp & 7);
			state = ns;
			const int isM = ns == 0 ? 1 : 0, isD = (ns == 1 || ns == 3) ? 1 : 0;
			rev[k++] = (uint8_t)(isM ? 'M' : (isD ? 'D' : 'I'));
			i -= isM | isD; j -= isM | (1 - isD);
		}
		for (; i >= 0; --i) rev[k++] = 'D';
		for (; j >= 0; --j) rev[k++] = 'I';
		s_n[w][q] = k;
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	if (have) {
		const int nops = s_n[w][q];
		uint8_t *op = ops + ops_off[job];
		for (int p = tl; p < nops; p += 16) op[p] = s_rev[w][q][nops - 1 - p];
		if (tl == 0) { ops_len[job] = nops; if (frag) frag[jfrag[job]].aln_len = nops; }
	}
}

// ---------------------------------------------------------------------------
// k_dp_lane (round 3): ONE LANE PER ALIGNMENT for everything below the striped kernel (n <= 64, m + n - 1 <= 128: 360 000 jobs and
// 104 M cells of a human-sized contig -- 7 % of the cells, but k_dp_small / k_dp_tiny spent 4.7 VALU instructions per cell on them,
// ten times the striped kernel: a systolic wave has 23 of 64 lanes busy on the median job, moves three values by DPP per step and
// walks back on one lane).  A lane needs no neighbour: cell (i, j) takes x, v from the cell on its left (registers) and u, y from
// the cell above (one 16-bit LDS entry per column, with the query base's code), row by row -- the same recurrence in another
// order (dp_cell is order-free: gsa_dp.h), so every lane is busy on every instruction.  What has to be managed is balance: a
// wavefront runs as long as its largest job.  A workgroup takes a tile of 512 jobs, counting-sorts it by cells (128 logarithmic
// bins in LDS) and its four waves draw batches of 64 size-neighbours, largest first.  Direction nibbles go to a per-wave arena in
// global memory (L2-resident: eight cells per dword, dword k of lane l at [k][l] -- coalesced), the traceback automaton reads them
// back, the reversed op string is staged in the LDS of the (dead) column entries.
// ---------------------------------------------------------------------------
#define LANE_TILE 512
#define LANE_BINS 128
#define LANE_KMAX 576           // direction dwords of one job at most: m * ceil(n / 8) with n <= 64, m + n - 1 <= 128 (n = 57, m = 72)
#ifndef LANE_WGS
#define LANE_WGS 1024           // persistent workgroups (four per CU: 38 KB of LDS each)
#endif
#define LANE_LDS_WAVE 8448      // 64 columns x 64 lanes x 2 bytes (forward)  |  (128 + 2) op bytes x 64 lanes (traceback)
__device__ __forceinline__ u32 lane_bin(u32 cells)      // floor(8 log2 cells): 1 <= cells < 8192 -> 0 .. 103
{
	const int msb = 31 - __clz((int)cells);
	const u32 frac = msb >= 3 ? (cells >> (msb - 3)) & 7u : (cells << (3 - msb)) & 7u;
	return ((u32)msb << 3) | frac;
}

__global__ void __launch_bounds__(256) k_dp_lane(i32 n_tiny, const i32 *__restrict__ order_tiny, i32 n_small, const i32 *__restrict__ order_small,
                                                  const uint8_t *__restrict__ pool1, const i64 *__restrict__ off1, const i32 *__restrict__ len1,
                                                  const uint8_t *__restrict__ pool2, const i64 *__restrict__ off2, const i32 *__restrict__ len2,
                                                  uint8_t *ops, const i64 *__restrict__ ops_off, i32 *ops_len, const i32 *__restrict__ jfrag, gsa_frag *frag, u32 *arena_all, u32 kstride)
{
	__shared__ u32 s_hist[LANE_BINS];
	__shared__ i32 s_sorted[LANE_TILE];
	__shared__ int s_next;
	__shared__ __attribute__((aligned(16))) uint8_t s_work[4][LANE_LDS_WAVE];
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const i64 n_all = (i64)n_tiny + n_small;
	u32 *arena = arena_all + ((size_t)blockIdx.x * 4 + w) * 64 * kstride + lane;      // dword k of my job: arena[k * 64]; kstride = the most dwords a job of this launch's class can need
	uint16_t *col = (uint16_t *)s_work[w] + lane;                                         // column i of my job: col[i * 64]
	uint8_t *revb = s_work[w] + lane;                                                     // reversed op k of my job: revb[k * 64]
	for (i64 t0 = (i64)blockIdx.x * LANE_TILE; t0 < n_all; t0 += (i64)gridDim.x * LANE_TILE) {
		// ---- the tile's jobs, largest first (counting sort by cells) ----
		if (tid < LANE_BINS) s_hist[tid] = 0;
		if (tid == 0) s_next = 0;
		__syncthreads();
		i32 jb[LANE_TILE / 256]; u32 key[LANE_TILE / 256], rk[LANE_TILE / 256];
#pragma unroll
		for (int k = 0; k < LANE_TILE / 256; k++) {
			const i64 idx = t0 + k * 256 + tid;
			jb[k] = -1; key[k] = 0; rk[k] = 0;
			if (idx < n_all) {
				const i32 job = idx < n_tiny ? order_tiny[idx] : order_small[idx - n_tiny];
				jb[k] = job; key[k] = lane_bin((u32)len1[job] * (u32)len2[job]); rk[k] = atomicAdd(&s_hist[key[k]], 1u);
			}
		}
		__syncthreads();
		if (tid < 64) {      // exclusive prefix over the bins in descending order (two bins per lane)
			const u32 a = s_hist[LANE_BINS - 1 - 2 * lane], b = s_hist[LANE_BINS - 2 - 2 * lane];
			u32 inc = a + b;
			for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(inc, o); if (lane >= o) inc += t; }
			const u32 ex = inc - (a + b);
			s_hist[LANE_BINS - 1 - 2 * lane] = ex; s_hist[LANE_BINS - 2 - 2 * lane] = ex + a;
		}
		__syncthreads();
#pragma unroll
		for (int k = 0; k < LANE_TILE / 256; k++) if (jb[k] >= 0) s_sorted[s_hist[key[k]] + rk[k]] = jb[k];
		__syncthreads();
		const int nt = (int)(n_all - t0 < LANE_TILE ? n_all - t0 : LANE_TILE), nb = (nt + 63) >> 6;
		for (;;) {
			int b = 0;
			if (lane == 0) b = atomicAdd(&s_next, 1);
			b = __builtin_amdgcn_readfirstlane(b);
			if (b >= nb) break;
			const int p = b * 64 + lane;
			const bool have = p < nt;
			const i32 job = have ? s_sorted[p] : 0;
			const int m = have ? len1[job] : 0, n = have ? len2[job] : 0;
			const uint8_t *s1 = pool1 + off1[job], *s2 = pool2 + off2[job];
			const int nw = (n + 7) >> 3;
			// column entries: u (5 bits) | y << 5 (5 bits) | 4 * code of the query base << 10; before row 0: u = 2 (0 in column 0), y = 0
			int nmax = n;
			for (int o = 32; o; o >>= 1) { const int t = __shfl_xor(nmax, o); nmax = t > nmax ? t : nmax; }
			for (int i = 0; i < nmax; i++) if (i < n) col[i * 64] = (uint16_t)((i ? 2 : 0) | (gsa_nt4(s2[i]) << 12));
			// ---- forward: row j = reference base, columns in GROUPS OF EIGHT (one direction dword; its eight entries are read together,
			//      the cells follow one another through x, v in registers, the loop control is paid once per group) ----
			bool act = have && m > 0 && n > 0;
			int g = 0, j = 0, x = 0, v = 0, kk = 0;
			// z = score + q + e of (query code a, row base b) as a nibble table over a: 7 match, 5 mismatch, 6 when either is N
			auto row_table = [](int b) -> u32 { return b == 4 ? 0x66666u : 0x65555u + (2u << (4 * b)); };
			u32 tbl = row_table(act ? gsa_nt4(s1[0]) : 4);
			uint8_t raw_next = (act && m > 1) ? s1[1] : (uint8_t)'N';      // (the next row's base: loaded a row ahead, decoded when the row starts)
			while (__any(act)) {
				if (act) {
					uint16_t *cg = col + g * 8 * 64;
					u32 e[8];
#pragma unroll
					for (int k = 0; k < 8; k++) e[k] = cg[k * 64];
					u32 acc = 0;
					const int left = n - g * 8;                            // valid columns of this group (>= 1; 8 or more: all)
#pragma unroll
					for (int k = 0; k < 8; k++) {
						if (k < left) {
							const int u = (int)(e[k] & 31u), y = (int)((e[k] >> 5) & 31u);
							const u32 c4 = e[k] >> 10;
							int z = (int)((tbl >> c4) & 15u);
							int a = x + v, b = y + u;
							int d = a > z ? 1 : 0; z = z > a ? z : a;
							if (b > z) d = 2;
							z = z > b ? z : b;
							z = z < 7 ? z : 7;
							const int un = z - v, vn = z - u;
							z -= 2; a -= z; b -= z;
							if (a > 0) d |= 0x08; else a = 0;
							if (b > 0) d |= 0x10; else b = 0;
							cg[k * 64] = (uint16_t)((u32)un | ((u32)b << 5) | (c4 << 10));
							x = a; v = vn;
							acc |= (u32)((d & 3) | ((d & 0x18) >> 1)) << (4 * k);
						}
					}
					arena[(size_t)kk * 64] = acc; kk++;
					g++;
					if (g == nw) {
						g = 0; j++; x = 0; v = 2;      // (left boundary of row j > 0: x = 0, v = 2; ksw2_alignment.cpp:157-164)
						if (j >= m) act = false;
						else { tbl = row_table(gsa_nt4(raw_next)); raw_next = j + 1 < m ? s1[j + 1] : (uint8_t)'N'; }
					}
				}
			}
			// ---- traceback (ksw_backtrack automaton, gsa_dp.h), one lane per job; the reversed ops go where the column entries were ----
			int ti = n - 1, tj = m - 1, state = 0, k = 0;
			bool tb = have && ti >= 0 && tj >= 0;
			while (__any(tb)) {
				if (tb) {
					const u32 wd = arena[(size_t)(tj * nw + (ti >> 3)) * 64];
					const u32 nbv = (wd >> ((ti & 7) << 2)) & 15u;
					const u32 tmp = (nbv & 3u) | ((nbv & 0xCu) << 1);
					int ns = state;
					if (ns != 0 && !((tmp >> (ns + 2)) & 1)) ns = 0;
					if (ns == 0) ns = (int)(tmp & 7);
					state = ns;
					const int isM = ns == 0 ? 1 : 0, isD = (ns == 1 || ns == 3) ? 1 : 0;
					revb[k * 64] = (uint8_t)(isM ? 'M' : (isD ? 'D' : 'I'));
					k++;
					ti -= isM | isD; tj -= isM | (1 - isD);
					tb = ti >= 0 && tj >= 0;
				}
			}
			if (have) {
				for (; ti >= 0; --ti) { revb[k * 64] = 'D'; k++; }
				for (; tj >= 0; --tj) { revb[k * 64] = 'I'; k++; }
				uint8_t *op = ops + ops_off[job];
				for (int q = 0; q < k; q++) op[q] = revb[(k - 1 - q) * 64];
				ops_len[job] = k;
				if (frag) frag[jfrag[job]].aln_len = k;      // (the job's record is final with this)
			}
		}
		__syncthreads();      // (the next tile reuses the bins and the sorted list)
	}
}

// ---------------------------------------------------------------------------
// k_dp_stripe: every alignment that does not fit the small kernel.  The n target
// columns are cut into stripes of 64; ONE WAVEFRONT PER PAIR OF STRIPES, the pairs of a
// job on whatever CUs the dispatcher picks (four pairs per workgroup), so a 1.5k x 1.5k
// problem runs on a dozen SIMDs instead of one.
// A wave keeps stripe 2pp in the low 16-bit halves of its registers and stripe 2pp+1, 64
// steps behind, in the high halves: the state (u, v, x, y <= 7 + q + e) fits, and the whole
// recurrence is packed 16-bit VALU (v_pk_add/sub/max/min/mad_u16: one instruction for both
// cells, 29 VALU instructions per step = 14.5 per cell; the one-stripe version had 34).
// On step s lane l handles rows s - l (A) and s - 64 - l (B); the left neighbours arrive by
// one DPP wave rotate of the packed (x | v << 8) pairs, unpacked with per-lane v_perm
// selectors that give lane 0 the boundary row (A) and A's lane 63 (B).  The substitution
// score is a v_perm over two 4-byte tables (one per stripe: my query base against A, C, G,
// T) with a selector per reference row staged in LDS.
// The only dependency between waves is the (x,v) pair of stripe 2pp+1's last column per row:
// it is handed over as self-validating 4-byte granules {tag,x|v<<8}, through LDS inside a
// workgroup and through HBM between workgroups (agent-scope relaxed atomics: write-through /
// L1-bypassing, so no fence and no separate flag; the tag is a 16-bit launch epoch, so the
// granules need no clearing between launches), 8 rows per store; the consumer fetches 8 rows
// per poll, one block ahead.
// Direction NIBBLES go to HBM STRIPE-LOCAL: stripe p owns (m+63) steps of 64 nibbles, eight
// steps per stored dword (assembled by packed multiply-adds), and every traceback tile is
// one contiguous block.
// The wave that finishes last (LDS ticket inside a workgroup, agent-scope release/acquire
// around a global ticket between workgroups) runs the traceback.  It keeps a DP_TILE_ROWS x 64
// tile of the current stripe in LDS and walks it RUN BY RUN: the lanes look ahead along the
// three possible directions (21 cells each) in one LDS read, a ballot gives the length
// of the run the automaton of ksw_backtrack would take step by step, and the run
// is emitted at once.
// Forward progress: a workgroup takes its place in the launch from a TICKET drawn when it starts (not from its
// workgroup index), so the pair pp-1 that pair pp waits for belongs to a workgroup that is already running,
// whatever order the dispatcher starts workgroups in and whatever else competes for the CUs.  The wait is still
// bounded (2 s of wall clock): a launch that trips it is repeated job by job (gsa_align_contig, dp_safe).
// ---------------------------------------------------------------------------
struct StripeJob { i32 job, m, n, P; i64 diroff, bndoff; i32 ctr, first_block; };
#define DP_TILE_ROWS 160        // local diagonals of a traceback tile (64 diagonal steps need 128); a multiple of 8
#define DP_STRIPE_BYTES(M) ((((size_t)(M) + 63 + 7) >> 3) << 8)      // (M + 63) anti-diagonals of 64 nibbles, in blocks of eight
#define DP_C1_PAD 320           // code bytes around the reference fragment: 128 "N" rows in front (stripe B starts 64 steps late, lane 63 another 63), the rest behind
#define DP_TILE_SLACK 16        // the prefetched tile reaches this far past the predicted entry
#ifndef DP_G
#define DP_G 8               // boundary rows per hand-off block (4, 8 or 16)
#endif
// -DGSA_DP_TIMING: in-kernel phase timers for tools/dp_probe.py (single-job launches only)
#ifdef GSA_DP_TIMING
#define DPT(...) __VA_ARGS__
#else
#define DPT(...)
#endif
// -DDP_EXP=bits: timing experiments only (results are wrong): 1 = no direction stores, 2 = no traceback
#ifndef DP_EXP
#define DP_EXP 0
#endif
#if DP_EXP & 1
#define DPX_STORE(...)
#else
#define DPX_STORE(...) __VA_ARGS__
#endif
#if DP_EXP & 2
#define DPX_TB(...) __VA_ARGS__
#else
#define DPX_TB(...)
#endif
#define DP_LOOK 21
#define DP_WAIT_TICKS 200000000ull   // bound of a hand-off wait: 2 s of the 100 MHz wall clock
#define DP_CLASS_M 768          // size classes of a long job list: reference fragments above / up to this (see launch_stripes)
#define DP_CLASS_TOP 1536       // ... and, round 5, the upper class cut once more: fragments above this keep the 64 KB layout, (768, 1536] run with 26 KB
#define DP_CLASS_MIN_JOBS 4096
#define DP_LDS_M 3968         // longest reference fragment for which four waves share a workgroup (selectors + 3 boundary columns in 64 KB of LDS)

// packed 16-bit arithmetic on the two halves of a register, spelled out: the compiler rewrites min(x, 1) and friends into
// per-half compares and selects (five instructions for one)
// Round 4: the packed instructions come from vector builtins (the compiler knows what they are: no `s_nop` behind every one of them, as there was
// behind each inline-asm statement -- 475 in the kernel).  What made round 2 spell them as inline asm -- `min(x, 1)` on packed shorts is rewritten into
// per-half compares and selects -- is avoided by keeping the constants opaque (DP_OPAQUE: a register the optimiser cannot see through).
#ifdef DP_PK_ASM
#define PK2(NAME, INS) __device__ __forceinline__ u32 NAME(u32 a, u32 b) { u32 d; asm(INS " %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
PK2(pk_add, "v_pk_add_u16") PK2(pk_sub, "v_pk_sub_u16") PK2(pk_max, "v_pk_max_u16") PK2(pk_min, "v_pk_min_u16")
__device__ __forceinline__ u32 pk_sub_sat(u32 a, u32 b) { u32 d; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }      // max(a - b, 0)
__device__ __forceinline__ u32 pk_mad(u32 a, u32 b, u32 c) { u32 d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ u32 pk_shl(u32 a, u32 sh) { u32 d; asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(d) : "v"(sh), "v"(a)); return d; }
#define DP_OPAQUE(X)
#else
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
#define V2(X) __builtin_bit_cast(v2u16, (u32)(X))
#define U1(X) __builtin_bit_cast(u32, (v2u16)(X))
__device__ __forceinline__ u32 pk_add(u32 a, u32 b) { return U1(V2(a) + V2(b)); }
__device__ __forceinline__ u32 pk_sub(u32 a, u32 b) { return U1(V2(a) - V2(b)); }
__device__ __forceinline__ u32 pk_max(u32 a, u32 b) { return U1(__builtin_elementwise_max(V2(a), V2(b))); }
__device__ __forceinline__ u32 pk_min(u32 a, u32 b) { return U1(__builtin_elementwise_min(V2(a), V2(b))); }
__device__ __forceinline__ u32 pk_sub_sat(u32 a, u32 b) { return U1(__builtin_elementwise_sub_sat(V2(a), V2(b))); }      // max(a - b, 0)
__device__ __forceinline__ u32 pk_mad(u32 a, u32 b, u32 c) { return U1(V2(a) * V2(b) + V2(c)); }
__device__ __forceinline__ u32 pk_shl(u32 a, u32 sh) { return U1(V2(a) << V2(sh)); }
#define DP_OPAQUE(X) asm volatile("" : "+v"(X))
#endif

template <int WPB>
__global__ void __launch_bounds__(64 * WPB) k_dp_stripe(const i32 *__restrict__ blk2job, const StripeJob *__restrict__ sjobs, const uint8_t *__restrict__ pool1, const i64 *__restrict__ off1,
                                                   const uint8_t *__restrict__ pool2, const i64 *__restrict__ off2, uint8_t *dirbase, u32 *bndbase, u32 *ctr,
                                                   uint8_t *revbase, uint8_t *ops, const i64 *__restrict__ ops_off, i32 *ops_len, u32 ep, i32 lds_c1, i32 lds_rows, u32 *err, i32 tick_slot)
{
	extern __shared__ __attribute__((aligned(16))) u32 C2[];           // the reference fragment as byte selectors of the two stripes' score tables
	// The traceback tile ALIASES the forward pass's LDS (codes + boundary columns): the wave that walks back drew the last
	// ticket of its job, so every stripe of the job -- every other wave of this workgroup -- is through with them.  LDS
	// per workgroup is what limits how many jobs (and which other kernels of the contig) a CU holds.
	uint8_t *tile = (uint8_t *)C2;
	// The LARGEST jobs are the contig's latency floor (the list is sorted by cells: they are the first workgroups): their
	// waves issue ahead of whatever else shares the SIMD.  The mass of smaller jobs behind them does not get that: on a
	// 50 Mb contig they are 10 000 workgroups, and at raised priority they starve the record / small-DP path beside them
	// (its passes ran 5-10x slower), which is the longer path there.
	__shared__ u32 s_bid, s_tick;
	if (threadIdx.x == 0) {
		s_tick = 0;
		const u32 tk = __hip_atomic_fetch_add(&ctr[tick_slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (tk == gridDim.x - 1) __hip_atomic_store(&ctr[tick_slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // last ticket of this launch: clean for the next one
		s_bid = tk;
	}
	__syncthreads();
	const u32 bid = s_bid;
#ifndef DP_PRIO_BLOCKS
#define DP_PRIO_BLOCKS 96      // the first workgroups of a launch (its largest jobs: the list is sorted by cells) issue at raised priority
#endif
	if (bid < DP_PRIO_BLOCKS) __builtin_amdgcn_s_setprio(3);
	// which job / pair of stripes am I (uniform).  Both tables are read where the host wrote them (pinned memory): two
	// dependent reads across the link cost less than a copy operation in front of the launch
	const StripeJob sj = sjobs[blk2job[bid]];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int m = sj.m, n = sj.n, P = sj.P, PP = (P + 1) >> 1;
	const int pp = ((int)bid - sj.first_block) * WPB + wave;            // stripes 2pp ("A", low halves) and 2pp + 1 ("B", high halves)
	const uint8_t *s1 = pool1 + off1[sj.job], *s2 = pool2 + off2[sj.job];
	const size_t pitch = (size_t)DP_STRIPE_BYTES(m);                    // direction nibbles of one stripe: 256 bytes per eight anti-diagonals
	const int nblk = (m + 63 + 7) >> 3;                                 // ... in so many blocks
	uint8_t *dir = dirbase + sj.diroff;
	u32 *bnd_in = bndbase + sj.bndoff + (size_t)(pp - 1) * m, *bnd_out = bndbase + sj.bndoff + (size_t)pp * m;
	// C2[128 + j] = v_perm selector of step-row j: byte 0 = code of reference row j (stripe A's table: bytes 0-3 of the pair),
	// byte 2 = 4 + code of row j - 64 (stripe B lags 64 steps: its table is bytes 4-7), 0x0c ("constant 0") for N and for
	// the rows in front of and behind the fragment, and in bytes 1, 3.  On step s lane l reads entry s - l.
	{
		int fe = (m + DP_C1_PAD + 63) & ~63; fe = fe < (lds_c1 >> 2) ? fe : (lds_c1 >> 2);
		for (int t = threadIdx.x; t < fe; t += 64 * WPB) {
			const int j = t - 128, jb = j - 64;
			const u32 ca = (j >= 0 && j < m) ? (u32)gsa_nt4(s1[j]) : 4u, cb = (jb >= 0 && jb < m) ? (u32)gsa_nt4(s1[jb]) : 4u;
			C2[t] = (ca < 4 ? ca : 0x0cu) | 0x0c000c00u | ((cb < 4 ? 4u + cb : 0x0cu) << 16);
		}
	}
	// WPB > 1: the waves of one workgroup hand their boundary column over through LDS (same granules, tag 0 = not yet)
	u32 *lds_bnd = C2 + (lds_c1 >> 2);
	if (WPB > 1) for (int t = threadIdx.x; t < (WPB - 1) * lds_rows; t += 64 * WPB) lds_bnd[t] = 0;
	u32 *lin = lds_bnd + (size_t)(wave > 0 ? wave - 1 : 0) * lds_rows, *lout = lds_bnd + (size_t)(wave < WPB - 1 ? wave : 0) * lds_rows;
	const bool out_lds = WPB > 1 && wave < WPB - 1;
	// TWO STRIPES PER WAVE, as the two 16-bit halves of every register: u, v, x, y are 0 ... 7 + q + e, so the recurrence runs on
	// packed 16-bit instructions (v_pk_add / max / min / sub: one instruction for both cells).  Stripe B lags 64 steps behind A:
	// on step s lane l holds A's cell (row s - l, column 128 pp + l) and B's cell (row s - 64 - l, column 128 pp + 64 + l), and
	// B's lane 0 takes its left neighbour -- A's lane 63, one step earlier -- from a readlane.
	const int tA = pp * 128 + lane, tB = tA + 64;
	const bool hasB = 2 * pp + 1 < P;
	const int WpB = !hasB ? 0 : (n - pp * 128 - 64 < 64 ? n - pp * 128 - 64 : 64);
	const int WpA = n - pp * 128 < 64 ? n - pp * 128 : 64;
	const int cqA = tA < n ? gsa_nt4(s2[tA]) : 4, cqB = tB < n ? gsa_nt4(s2[tB]) : 4;
	// z = score + q + e (ksw2_alignment.cpp:74-95: match 1, mismatch -1, N 0 -> 7, 5, 6) as a byte table over the reference
	// code, stored XOR 6 so that the selector's "constant 0" is the N row: z = v_perm(tables, selector) ^ 6 in both halves
	u32 tblA = 0, tblB = 0;
#pragma unroll
	for (int cc = 0; cc < 4; cc++) {
		tblA |= (u32)(cqA == 4 ? 0 : (cqA == cc ? 7 ^ 6 : 5 ^ 6)) << (8 * cc);
		tblB |= (u32)(cqB == 4 ? 0 : (cqB == cc ? 7 ^ 6 : 5 ^ 6)) << (8 * cc);
	}
	const u32 uinit2 = (tA ? 2u : 0u) | (2u << 16);
	u32 u2 = uinit2, y2 = 0;
	u32 bin = 0, gnext = 0;
	__syncthreads();
	if (pp >= PP) return;                               // (a workgroup's spare waves only helped to stage the fragment)
	const int S_end = hasB ? 64 + m + WpB - 1 : m + WpA - 1;      // steps of this wave
	DPT(const unsigned long long T0c = wall_clock64();)
	u32 *dirA = (u32 *)(dir + (size_t)(2 * pp) * pitch), *dirB = (u32 *)(dir + (size_t)(2 * pp + 1) * pitch);
	// boundary granules are fetched ONE BLOCK AHEAD (8 rows per block) so their L2 latency overlaps the block before
	if (pp > 0) {
		const int row = (lane & (DP_G - 1)) < m ? (lane & (DP_G - 1)) : m - 1;
		gnext = (WPB > 1 && wave > 0) ? __hip_atomic_load(&lin[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : __hip_atomic_load(&bnd_in[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	asm volatile("" :: "v"(gnext));                      // the first prefetch is complete before the loop: inside it, waits then only count stores issued after a prefetch
	const bool pub_stripe = pp < PP - 1;                // (then stripe B is full and its lane 63 owns the boundary column)
	u32 pk2 = 0;                                        // x | v << 8 of my two columns after the current step: bytes xA, vA, xB, vB
	u32 hb = 0;                                         // lanes 0-7: stripe B's boundary column (x | v << 8 of its lane 63), the eight rows of the current block
	const u32 selx = lane ? 0x0c060c04u : 0x0c040c00u, selv = lane ? 0x0c070c05u : 0x0c050c01u;      // x, v of (lane - 1 | boundary, A's lane 63) from (rotated pairs, boundary row)
	u32 acc = 0, r0 = 0;                   // direction nibbles of the last four steps (per half, oldest on top); those of the four before
	u32 c1 = 0x00010001u, c2 = 0x00020002u, c4 = 0x00040004u, c7 = 0x00070007u, c16 = 0x00100010u;
	DP_OPAQUE(c1); DP_OPAQUE(c2); DP_OPAQUE(c4); DP_OPAQUE(c7); DP_OPAQUE(c16);
	// one step; K2 is the position inside the 16-step block (a literal in the unrolled body).  GUARD = 1: the first 128 steps (lanes
	// that have not reached row 0 yet are put back to the initial state after every step) and the last blocks (stripe A's
	// direction blocks have an end).  Nothing masks the cells a lane computes outside the matrix -- rows >= m, columns >= n:
	// their values only ever reach other such cells, and their direction nibbles are never read.
// lane K of HB takes the wave-uniform VAL (v_writelane_b32; K a literal / not)
#define DP_WL_LIT(HB, VAL, K) asm("v_writelane_b32 %0, %1, %2" : "+v"(HB) : "s"(VAL), "n"(K));
#define DP_WL_VAR(HB, VAL, K) if (lane == (K)) HB = (VAL);
#define DP_LOADG(MODE, ROW) ((MODE) == 2 ? __hip_atomic_load(&lin[ROW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : __hip_atomic_load(&bnd_in[ROW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
#define DP_STEP(K2, MODE, WSEL, GUARD, WL)                                                                       \
	{                                                                                                           \
		const int s_ = s0 + (K2);                                                                               \
		if (((K2) & (DP_G - 1)) == 0) {                                                                         \
			if ((MODE) == 0) bin = (s_ == 0 && lane == 0) ? 0u : 0x200u;    /* t = 0 boundary: x1 = 0, v1 = q, except for the very first cell (:157-164) */ \
			else if (s_ < m) {                                                                                  \
				/* boundary rows s_ .. s_+DP_G-1 from the pair in front: spin until every granule carries its tag */ \
				const int row = s_ + (lane & (DP_G - 1));                                                       \
				const bool need = lane < DP_G && row < m;                                                       \
				u32 g = gnext;                                                                                  \
				if (!__all(!need || (g >> 16) == ep)) {      /* (first look outside the loop: its wait only covers the prefetch) */ \
					u32 spins = 0; const unsigned long long t_wait0 = wall_clock64();                          \
					do {                                                                                        \
						if ((++spins & 255) == 0 && (wall_clock64() - t_wait0 > DP_WAIT_TICKS || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; } \
						__builtin_amdgcn_s_sleep(1);                                                            \
						if (need) g = DP_LOADG(MODE, row);                                                      \
					} while (!__all(!need || (g >> 16) == ep));                                                  \
				}                                                                                               \
				bin = g & 0xffffu;                                                                              \
				/* every lane loads (clamped row): an unconditional load lands in gnext without a copy that would wait for it */ \
				const int rown = row + DP_G < m ? row + DP_G : m - 1;                                           \
				gnext = DP_LOADG(MODE, rown);                                                                   \
			}                                                                                                   \
		}                                                                                                       \
		/* left neighbours: lane l-1's pair; lane 0 takes (boundary row of A | A's lane 63 for B) */               \
		/* left neighbours: the pairs rotate one lane up (DPP wave_ror:1); lane 0 -- its own byte selectors -- takes the   \
		   boundary row for A and A's lane 63 (row s_ - 64, computed one step ago) for B */                          \
		const u32 bin0 = (u32)__builtin_amdgcn_readlane((int)bin, (K2) & (DP_G - 1));                           \
		const u32 rot = (u32)__builtin_amdgcn_mov_dpp((int)pk2, 0x13C, 0xf, 0xf, true);                         \
		WL(hb, (u32)__builtin_amdgcn_readlane((int)pk2, 63) >> 16, (K2) & 7)      /* B's lane 63: boundary row s_ - 128 */ \
		const u32 x1 = __builtin_amdgcn_perm(rot, bin0, selx), v1 = __builtin_amdgcn_perm(rot, bin0, selv);     \
		const u32 z0 = __builtin_amdgcn_perm(tblB, tblA, (WSEL)) ^ 0x00060006u;                                 \
		const u32 a = pk_add(x1, v1), b = pk_add(y2, u2);                                                       \
		const u32 z1 = pk_max(z0, a), z2 = pk_max(z1, b);                                                       \
		const u32 ta = pk_min(pk_sub(z1, z0), c1), tb = pk_min(pk_sub(z2, z1), c1);      /* a > z; b > max(z, a): the direction is tb ? 2 : ta */ \
		const u32 zc = pk_min(z2, c7);                                                                          \
		const u32 un = pk_sub(zc, v1), vn = pk_sub(zc, u2), zz = pk_sub(zc, c2);                                \
		const u32 xa = pk_sub_sat(a, zz), yb = pk_sub_sat(b, zz);                                               /* x, y */ \
		const u32 fa = pk_min(xa, c1), fb = pk_min(yb, c1);                                                     /* the "x / y is positive" flags (0x08, 0x10 of ksw2) */ \
		/* direction NIBBLES ta | tb << 1 | fa << 2 | fb << 3, four steps per 16-bit half, eight steps per stored dword: the wave \
		   stores 256 bytes per stripe and eight steps (a byte store per step kept the address unit busier than the ALU) */ \
		acc = pk_mad(acc, c16, pk_mad(pk_mad(fb, c2, fa), c4, pk_mad(tb, c2, ta)));                             \
		u2 = un; y2 = yb;                                                                                       \
		pk2 = __builtin_amdgcn_perm(vn, xa, 0x06020400u);      /* bytes xA, vA, xB, vB (values outside the matrix may not fit a byte: cut, not carried into the neighbour) */ \
		if (GUARD) {                                                                                            \
			const u32 keep = (lane <= s_ ? 0xffffu : 0u) | (lane <= s_ - 64 ? 0xffff0000u : 0u);               \
			u2 = (u2 & keep) | (uinit2 & ~keep); y2 &= keep;                                                    \
		}                                                                                                       \
		if (((K2) & 7) == 3) r0 = acc;                                                                  \
		if (((K2) & 7) == 7) {                                                                                  \
			const int blk = s_ >> 3;                                                                            \
			if (!(GUARD) || blk < nblk) { DPX_STORE(dirA[((size_t)blk << 6) + lane] = __builtin_amdgcn_perm(acc, r0, 0x05040100u);) } \
			if (hasB && (!(GUARD) || (blk >= 8 && blk - 8 < nblk))) { DPX_STORE(dirB[((size_t)(blk - 8) << 6) + lane] = __builtin_amdgcn_perm(acc, r0, 0x07060302u);) } \
		}                                                                                                       \
		if (((K2) & 7) == 7 && pub_stripe && s_ >= 135) {                                                       \
			/* rows s_-135 .. s_-128 of B's boundary column are complete: one store of eight tagged granules */   \
			const int row = s_ - 135 + lane;                                                                    \
			if (lane < 8 && row < m) {                                                                          \
				if (out_lds) __hip_atomic_store(&lout[row], (ep << 16) | hb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
				else __hip_atomic_store(&bnd_out[row], (ep << 16) | hb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
			}                                                                                                   \
		}                                                                                                       \
	}
	// (three copies of the loop: the first pair has no boundary loads in flight, and keeping it apart keeps its waits off the stores)
#define DP_LOOP(MODE)                                                                                           \
	for (int s0 = 0; s0 < S_end; s0 += 16) {                                                                    \
		const u32 *crow = C2 + 128 + s0 - lane;                /* my selectors from step s0 on */                  \
		if (s0 + 16 <= S_end) {                                                                                 \
			u32 w[16];                                                                                          \
			_Pragma("unroll") for (int k2 = 0; k2 < 16; k2++) w[k2] = crow[k2];                                   \
			if (s0 >= 128 && s0 + 16 <= 8 * nblk) {                                                             \
				/* the steady state of a long stripe pair: every lane is under way, every direction block exists */ \
				DP_STEP(0, MODE, w[0], 0, DP_WL_LIT) DP_STEP(1, MODE, w[1], 0, DP_WL_LIT) DP_STEP(2, MODE, w[2], 0, DP_WL_LIT) DP_STEP(3, MODE, w[3], 0, DP_WL_LIT) DP_STEP(4, MODE, w[4], 0, DP_WL_LIT) DP_STEP(5, MODE, w[5], 0, DP_WL_LIT) DP_STEP(6, MODE, w[6], 0, DP_WL_LIT) DP_STEP(7, MODE, w[7], 0, DP_WL_LIT) \
				DP_STEP(8, MODE, w[8], 0, DP_WL_LIT) DP_STEP(9, MODE, w[9], 0, DP_WL_LIT) DP_STEP(10, MODE, w[10], 0, DP_WL_LIT) DP_STEP(11, MODE, w[11], 0, DP_WL_LIT) DP_STEP(12, MODE, w[12], 0, DP_WL_LIT) DP_STEP(13, MODE, w[13], 0, DP_WL_LIT) DP_STEP(14, MODE, w[14], 0, DP_WL_LIT) DP_STEP(15, MODE, w[15], 0, DP_WL_LIT) \
			} else {                                                                                            \
				DP_STEP(0, MODE, w[0], 1, DP_WL_LIT) DP_STEP(1, MODE, w[1], 1, DP_WL_LIT) DP_STEP(2, MODE, w[2], 1, DP_WL_LIT) DP_STEP(3, MODE, w[3], 1, DP_WL_LIT) DP_STEP(4, MODE, w[4], 1, DP_WL_LIT) DP_STEP(5, MODE, w[5], 1, DP_WL_LIT) DP_STEP(6, MODE, w[6], 1, DP_WL_LIT) DP_STEP(7, MODE, w[7], 1, DP_WL_LIT) \
				DP_STEP(8, MODE, w[8], 1, DP_WL_LIT) DP_STEP(9, MODE, w[9], 1, DP_WL_LIT) DP_STEP(10, MODE, w[10], 1, DP_WL_LIT) DP_STEP(11, MODE, w[11], 1, DP_WL_LIT) DP_STEP(12, MODE, w[12], 1, DP_WL_LIT) DP_STEP(13, MODE, w[13], 1, DP_WL_LIT) DP_STEP(14, MODE, w[14], 1, DP_WL_LIT) DP_STEP(15, MODE, w[15], 1, DP_WL_LIT) \
			}                                                                                                   \
		} else {                                                                                                \
			for (int k2 = 0; s0 + k2 < S_end; k2++) DP_STEP(k2, MODE, crow[k2], 1, DP_WL_VAR)                                 \
		}                                                                                                       \
	}
	if (pp == 0) { DP_LOOP(0) } else if (WPB > 1 && wave > 0) { DP_LOOP(2) } else { DP_LOOP(1) }
#undef DP_LOOP
#undef DP_STEP
#undef DP_LOADG
	if (S_end & 7) {
		// the last, partial dwords of the two stripes
		if (S_end & 3) acc = pk_shl(acc, (u32)(4 * (4 - (S_end & 3))) * 0x00010001u);
		if ((S_end & 7) < 4) r0 = acc;
		const int blk = S_end >> 3;
		if (blk < nblk) { DPX_STORE(dirA[((size_t)blk << 6) + lane] = __builtin_amdgcn_perm(acc, r0, 0x05040100u);) }
		if (hasB && blk >= 8 && blk - 8 < nblk) { DPX_STORE(dirB[((size_t)(blk - 8) << 6) + lane] = __builtin_amdgcn_perm(acc, r0, 0x07060302u);) }
	}
	if (pub_stripe) {
		// the last (partial) block of boundary rows: row m - 1 is lane 63's value after the last step (S_end = m + 127 here)
		if (lane == (S_end & 7)) hb = (u32)__builtin_amdgcn_readlane((int)pk2, 63) >> 16;
		const int row = (S_end & ~7) - 128 + lane;
		if (lane <= (S_end & 7) && row >= 0 && row < m) {
			if (out_lds) __hip_atomic_store(&lout[row], (ep << 16) | hb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			else __hip_atomic_store(&bnd_out[row], (ep << 16) | hb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	DPT(if (pp == 0 && lane == 0) ctr[sj.ctr + 40] = (u32)(wall_clock64() - T0c); if (pp == PP - 1 && lane == 0) ctr[sj.ctr + 41] = (u32)(wall_clock64() - T0c);)
	// ---- ticket: the last wave to finish does the traceback ----
	// A job whose stripes all sit in THIS workgroup (n <= 128 WPB: most of the 22 thousand striped jobs of a human-sized contig)
	// synchronises at workgroup scope with an LDS ticket; agent scope -- stripes in workgroups on other XCDs, whose L2s are not
	// coherent with each other -- means an L2 write-back per stripe and an invalidate in front of the traceback.
	// Longer jobs: the waves of a workgroup first count themselves in LDS, only the last one of each workgroup pays the
	// agent-scope release and draws the job's global ticket (one per workgroup instead of one per wave).
	const bool one_wg = WPB > 1 && PP <= WPB;
	const int first_p = ((int)bid - sj.first_block) * WPB;                 // pairs of this workgroup: first_p .. first_p + mine - 1
	const int mine = PP - first_p < WPB ? PP - first_p : WPB, n_wg = (PP + WPB - 1) / WPB;
	u32 ticket = 0;
	if (WPB > 1) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (lane == 0) ticket = __hip_atomic_fetch_add(&s_tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		ticket = (u32)__builtin_amdgcn_readfirstlane((int)ticket);
		if ((int)ticket != mine - 1) return;
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	}
	if (!one_wg) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (lane == 0) ticket = __hip_atomic_fetch_add(&ctr[sj.ctr], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		ticket = (u32)__builtin_amdgcn_readfirstlane((int)ticket);
		if ((int)ticket != (WPB > 1 ? n_wg : PP) - 1) return;
		if (lane == 0) ctr[sj.ctr] = 0;                                     // (nobody else looks again: the counters stay clean for the next launch)
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	}
	DPX_TB(if (lane == 0) ops_len[sj.job] = 0; return;)
	DPT(const unsigned long long T1c = wall_clock64(); int ntile = 0, nrun = 0;)
	uint8_t *rev = revbase + ops_off[sj.job], *op = ops + ops_off[sj.job];
	int i = n - 1, j = m - 1, state = 0, k = 0;
	// lane l looks at the cell e = l % 21 steps ahead along direction g = l / 21 (0: M, 1: D, 2: I)
	const int g = lane / DP_LOOK, e = lane - g * DP_LOOK;
	const int dlc = g == 2 ? 0 : -e, drl = g == 0 ? -2 * e : -e;
	// The next tile is fetched while the current one is walked: an alignment path runs along the diagonal, so from (i, j)
	// it will enter the stripe to the left near row j - (lc + 1); those diagonals (+-DP_TILE_SLACK for indels on the way)
	// are loaded into registers now and only written to LDS when the walker gets there.  A wrong guess costs nothing but
	// the load: the tile is then fetched the plain way.
	// (a tile = DP_TILE_ROWS / 8 blocks of eight diagonals = 5 KB; block-aligned)
	constexpr int TB_BLKS = DP_TILE_ROWS / 8, TB_VEC = TB_BLKS * 16 / 64;
	static_assert(TB_VEC == 5, "the prefetched tile is five named registers (an array was kept in scratch: 96 bytes per lane)");
	uint4 pf0 = {0, 0, 0, 0}, pf1 = pf0, pf2 = pf0, pf3 = pf0, pf4 = pf0;
#define PF_EACH(X) X(0, pf0) X(1, pf1) X(2, pf2) X(3, pf3) X(4, pf4)
	int pf_sp = -1, pf_lo = 0, pf_hi = -1;
	const int rl_max = m - 1 + 63;                                      // last local diagonal of a stripe
	const u32 *tile32 = (const u32 *)tile;
	while (i >= 0 && j >= 0) {
		i = __builtin_amdgcn_readfirstlane(i); j = __builtin_amdgcn_readfirstlane(j);
		// tile: stripe sp, local diagonals rl_lo .. rl_hi
		DPT(ntile++;)
		const int sp = i >> 6, rl_hi = j + (i & 63);
		int rl_lo;
		uint4 *dst = (uint4 *)tile;
		if (sp == pf_sp && rl_hi <= pf_hi && rl_hi - pf_lo >= 64) {
			rl_lo = pf_lo;
			DPT(nrun += 1 << 16;)
#define PF_PUT(Q, R) dst[(Q) * 64 + lane] = R;
			PF_EACH(PF_PUT)
#undef PF_PUT
		} else {
			const int b_hi = rl_hi >> 3, b_lo = b_hi - (TB_BLKS - 1) > 0 ? b_hi - (TB_BLKS - 1) : 0;
			rl_lo = b_lo << 3;
			const uint4 *src = (const uint4 *)(dir + (size_t)sp * pitch + ((size_t)b_lo << 8));
			const int nvec = (b_hi - b_lo + 1) * 16;
#pragma unroll
			for (int q2 = 0; q2 < TB_VEC; q2++) { const int id = q2 * 64 + lane; if (id < nvec) dst[id] = src[id]; }
		}
		pf_sp = -1;
		{
			const int jp = j - ((i & 63) + 1);
			if (sp > 0 && jp >= 0) {
				int hi = jp + 63 + DP_TILE_SLACK; hi = hi < rl_max ? hi : rl_max;
				const int pb_hi = hi >> 3, pb_lo = pb_hi - (TB_BLKS - 1) > 0 ? pb_hi - (TB_BLKS - 1) : 0;
				const uint4 *src = (const uint4 *)(dir + (size_t)(sp - 1) * pitch + ((size_t)pb_lo << 8));
				const int nvec = (pb_hi - pb_lo + 1) * 16;
#define PF_GET(Q, R) { const int id = (Q) * 64 + lane; R = src[id < nvec ? id : 0]; }
				PF_EACH(PF_GET)
#undef PF_GET
				pf_sp = sp - 1; pf_lo = pb_lo << 3; pf_hi = (pb_hi << 3) + 7 < rl_max ? (pb_hi << 3) + 7 : rl_max;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		for (;;) {
			// the walker state is wave-uniform: pin it to scalar registers
			i = __builtin_amdgcn_readfirstlane(i); j = __builtin_amdgcn_readfirstlane(j);
			state = __builtin_amdgcn_readfirstlane(state); k = __builtin_amdgcn_readfirstlane(k);
			if (i < 0 || j < 0) break;
			const int lc = i - (sp << 6), rl = j + lc;
			if (lc < 0 || rl < rl_lo) break;                       // left the tile: reload
			const int lc2 = lc + dlc, rl2 = rl + drl;
			const bool valid = lane < 3 * DP_LOOK && lc2 >= 0 && rl2 >= rl_lo && rl2 - lc2 >= 0;
			u32 tmp = 0xffu;
			if (valid) {      // back to ksw2's flag byte (nibble of step k: half k / 4, oldest on top)
				const u32 nb = (tile32[(((rl2 - rl_lo) >> 3) << 6) + lc2] >> ((((rl2 & 7) >> 2) << 4) + ((3 - (rl2 & 3)) << 2))) & 15u;
				tmp = ((nb & 2u) ? 2u : (nb & 1u)) | ((nb & 0xCu) << 1);
			}
			const u32 cur = (u32)__builtin_amdgcn_readfirstlane((int)tmp);
			// the automaton of ksw_backtrack (:38-52) for the current cell ...
			int S = state;
			if (S != 0 && !((cur >> (S + 2)) & 1)) S = 0;
			if (S == 0) S = (int)(cur & 7);
			const int isM = S == 0 ? 1 : 0, isD = (S == 1 || S == 3) ? 1 : 0;
			const int gS = isM ? 0 : (isD ? 1 : 2);
			// ... and for the cells behind it while they keep the same state
			const bool cont = valid && g == gS && (isM ? (tmp & 7) == 0 : (((tmp >> (S + 2)) & 1) != 0 || (int)(tmp & 7) == S));
			const unsigned long long bal = __ballot(cont) >> (gS * DP_LOOK + 1);
			int run = __builtin_ctzll(~bal);
			run = run < DP_LOOK - 1 ? run : DP_LOOK - 1;
			const int L = 1 + run;
			if (lane < L) rev[k + lane] = (uint8_t)(isM ? 'M' : (isD ? 'D' : 'I'));
			k += L; state = S; DPT(nrun++;)
			i -= (isM | isD) ? L : 0; j -= (isM | (1 - isD)) ? L : 0;
		}
	}
	DPT(if (lane == 0) { ctr[sj.ctr + 42] = (u32)(wall_clock64() - T1c); ctr[sj.ctr + 43] = ntile; ctr[sj.ctr + 44] = nrun; ctr[sj.ctr + 45] = (u32)(wall_clock64() - T0c); })
	if (lane == 0) {
		for (; i >= 0; --i) rev[k++] = 'D';
		for (; j >= 0; --j) rev[k++] = 'I';
		ops_len[sj.job] = k;
	}
	k = __builtin_amdgcn_readfirstlane(k);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	for (int q2 = lane; q2 < k; q2 += 64) op[q2] = rev[k - 1 - q2];
}

// Size classes on the device: one fused pass (gsa_scan.h) lists the jobs that need the striped kernel
// as (job, m, n) triples and packs the others into the small kernel's order array.
struct OpClassify {
	const i32 *len1, *len2; i32 *order, *order_tiny, *lg, *jlarge, *mail;
	static constexpr bool clamped = true;      // (gsa_scan.h: every element's loads unconditional and together; beyond the job count the lengths are stale, not used)
	struct Item { i32 in, m, n; };
	__device__ Item load(i64 j) const { Item it; it.in = 0; it.m = len1[j]; it.n = len2[j]; return it; }
	__device__ void prep(Item &it, i64 j) const { it.in = j < mail[M_NJOB] ? 1 : 0; if (!it.in) it.m = it.n = 0; }
	__device__ i32 value(const Item &it, i64, int c) const
	{
		if (!it.in) return 0;
		const i32 m = it.m, n = it.n;
		if (c == 0) return dp_is_large(m, n) ? 1 : 0;
		if (dp_is_large(m, n)) return 0;
		if (lane_cells > 0) return m * n <= lane_cells ? 1 : 0;         // one lane each (k_dp_lane); the rest of the class: one wavefront each (k_dp_small)
		return (n <= 16 && m + n - 1 <= TINY_ROWS) ? 1 : 0;            // (GSA_DP_LANE=0) four of these share a wavefront
	}
	__device__ void emit(const Item &it, i64 j, const i32 *v, const i32 *ex) const
	{
		if (!it.in) return;
		const i32 m = it.m, n = it.n;
		if (m <= 0 || n <= 0) lb_pub(&mail[M_DPERR], 2);
		jlarge[j] = v[0];
		if (v[0]) { i32 *e = lg + 3 * (size_t)ex[0]; lb_pub(&e[0], (i32)j); lb_pub(&e[1], m); lb_pub(&e[2], n); }      // (finish() reads the list)
		else if (v[1]) order_tiny[ex[1]] = (i32)j;
		else order[j - ex[0] - ex[1]] = (i32)j;
	}
	__device__ void done(const i32 *t) const { lb_pub(&mail[M_NLARGE], t[0]); lb_pub(&mail[M_NTINY], t[1]); }
	// the last tile puts the mailbox and the head of the large-job list into pinned memory (the host launches from there)
	i32 *h_out; i32 h_cap; i32 lane_cells;
	__device__ void finish(int tid) const
	{
		if (tid < MAIL_N) h_out[tid] = __hip_atomic_load(&mail[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		i32 nl = __hip_atomic_load(&mail[M_NLARGE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (nl > h_cap) nl = h_cap;
		lb_copy_out(h_out + MAIL_N, lg, 3 * nl, tid);
	}
};

// sums of m*n and m+n over the jobs (measurement only)
__global__ void k_dp_cells(const i32 *__restrict__ mail, const i32 *__restrict__ len1, const i32 *__restrict__ len2, unsigned long long *out)
{
	const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	const bool in = j < mail[M_NJOB];
	unsigned long long v = in ? (unsigned long long)len1[j] * (unsigned long long)len2[j] : 0, w = in ? (unsigned long long)(len1[j] + len2[j]) : 0;
	for (int o = 32; o; o >>= 1) { v += __shfl_xor(v, o); w += __shfl_xor(w, o); }
	if ((threadIdx.x & 63) == 0 && v) { atomicAdd(out, v); atomicAdd(out + 1, w); }
}

void dp_count_cells(gsa_ctx *c, i32 n_ub, const i32 *len1, const i32 *len2, hipStream_t stream)
{
	hipLaunchKernelGGL(k_dp_cells, dim3(grid_for((size_t)n_ub, 256)), dim3(256), 0, stream, c->d_mail.as<i32>(), len1, len2, (unsigned long long *)(c->d_mail.as<i32>() + M_CELLS));
}

#define LG_CHUNK 2048       // large-job triples copied together with the mailbox (more -> a second copy)

// Striped kernel for a list of large jobs on stream `ss` (batches so that the direction bytes of one batch fit the
// budget).  The direction / boundary / ticket buffers are shared: two launches must not be in flight together.
int launch_stripes(gsa_ctx *c, hipStream_t st, std::vector<LgJob> &large, const uint8_t *pool1, const i64 *off1, const uint8_t *pool2, const i64 *off2,
                   uint8_t *ops, const i64 *ops_off, i32 *ops_len, uint8_t *rev, int err_slot)
{
	if (large.empty()) return GSA_OK;
	i32 *mail = c->d_mail.as<i32>();
	// largest first: they are the critical path (only the head of a long list is ordered: the rest fills the machine anyway)
	{
		auto by_cells = [](const LgJob &a, const LgJob &b) { const i64 ca = (i64)a.m * a.n, cb = (i64)b.m * b.n; return ca != cb ? ca > cb : a.job < b.job; };
		if (large.size() > 512) std::partial_sort(large.begin(), large.begin() + 256, large.end(), by_cells);
		else std::sort(large.begin(), large.end(), by_cells);
	}
	// Size classes.  Every workgroup of a launch reserves the LDS the launch's LONGEST reference fragment needs (the code
	// string + three boundary columns): with thousands of jobs that caps the chip at 3 workgroups per CU although almost
	// all of them are small.  Long lists are therefore launched as two kernels, back to back on the stream: fragments
	// above DP_CLASS_M first (the critical ones), the rest behind them with a quarter of the LDS.  Short lists (a bacterial
	// contig: 700 jobs) stay one launch -- there the second kernel would only wait for the longest job of the first.
	// (in front of both: the few fragments above DP_LDS_M, whose boundary columns do not fit LDS -- one wave per workgroup, hand-off
	//  through HBM; they used to drag the whole upper class down to that layout)
	// Round 5: the upper class is cut once more at DP_CLASS_TOP.  Its workgroups hold 16 bytes of LDS per reference row of the class's LONGEST fragment
	// (53 - 64 KB: two or three workgroups per CU), while nine in ten of its jobs are shorter than 1 536 rows (26 KB: six per CU) -- the class ran
	// four rounds of ~0.5 ms on a 250 Mb contig although its longest job needs one.
	size_t n_xl = 0, n_top = large.size(), n_hi = large.size();
	{
		auto it0 = std::stable_partition(large.begin(), large.end(), [](const LgJob &g) { return g.m > DP_LDS_M; });
		n_xl = (size_t)(it0 - large.begin());
		if (large.size() >= DP_CLASS_MIN_JOBS) {
			auto it1 = std::stable_partition(it0, large.end(), [](const LgJob &g) { return g.m > DP_CLASS_TOP; });
			auto it = std::stable_partition(it1, large.end(), [](const LgJob &g) { return g.m > DP_CLASS_M; });
			n_top = (size_t)(it1 - large.begin()); n_hi = (size_t)(it - large.begin());
		}
	}
	for (const LgJob &g : large) if ((((g.m + 63) & ~63) + DP_C1_PAD) * 4 > 150 * 1024) return gsa_fail(c, GSA_ERR_LIMIT, "DP reference-side fragment longer than 38000 bases");
	const i64 budget = 12ll << 30;
	size_t first = 0;
	while (first < large.size()) {
		// descriptors are staged in pinned memory: the upload is asynchronous
		size_t cnt = 0;
		if (c->dp_safe) cnt = 1;      // (retry after a hand-off time-out: one job per launch, every stripe of it resident at once)
		else { size_t l = first; i64 db = 128; while (l < large.size()) { const i64 cells = (((i64)large[l].n + 63) / 64) * (i64)DP_STRIPE_BYTES(large[l].m); if (l > first && db + cells > budget) break; db += cells + 128; l++; } cnt = l - first; }
		// (the early launch and a late one may be in flight together: each has its own table)
		DevBuf &psj = err_slot == M_DPERR3 ? c->p_sj_early : c->p_sj;
		// the segments of this batch: [first, s0) above DP_LDS_M, [s0, st) above DP_CLASS_TOP, [st, s1) above DP_CLASS_M, [s1, first + cnt) below
		const size_t s0 = std::min(std::max(n_xl, first), first + cnt), stp = std::min(std::max(n_top, first), first + cnt), s1 = std::min(std::max(n_hi, first), first + cnt);
		constexpr int NSEG = 4;
		struct Seg { size_t b, e; int mmax, wpb, mpad, lds_rows; size_t dyn_lds; i32 *b2j; i32 nblocks; } seg[NSEG] = { { first, s0 }, { s0, stp }, { stp, s1 }, { s1, first + cnt } };
		size_t nb_ub = 0;
		for (Seg &sg : seg) {
			sg.mmax = 1;
			for (size_t k = sg.b; k < sg.e; k++) if (large[k].m > sg.mmax) sg.mmax = large[k].m;
			sg.mpad = ((sg.mmax + 63) & ~63) + DP_C1_PAD;      // + the "N" rows in front and behind (see the kernel)
			sg.wpb = sg.mmax <= DP_LDS_M ? 4 : 1;      // reference fragments up to DP_LDS_M bases: four waves (eight stripes) per workgroup, boundary columns through LDS
			sg.lds_rows = (sg.mmax + 15) & ~15;
			sg.dyn_lds = (size_t)sg.mpad * 4 + (sg.wpb > 1 ? (size_t)(sg.wpb - 1) * sg.lds_rows * 4 : 0);      // (one selector dword per row)
			if (sg.dyn_lds < (size_t)DP_TILE_ROWS * 32) sg.dyn_lds = (size_t)DP_TILE_ROWS * 32;      // (the traceback tile -- nibbles -- lives in the same bytes)
			// option dp_occupancy (experiment): a four-wave workgroup of the lower classes asks for 1/occ of a CU's LDS, so that at most `occ` of them sit on a
			// CU and one or two workgroup slots stay free for the fused passes that run beside the striped kernel (they wait for wave slots otherwise)
			if (c->opt.dp_occupancy > 0 && sg.wpb == 4) { const size_t want = (size_t)(160 * 1024) / (size_t)c->opt.dp_occupancy - 1024; if (sg.dyn_lds < want && want <= 64 * 1024) sg.dyn_lds = want; }
			for (size_t k = sg.b; k < sg.e; k++) nb_ub += (size_t)((((large[k].n + 63) / 64 + 1) / 2 + sg.wpb - 1) / sg.wpb);      // (a wave takes two stripes)
		}
		if (!pin_ensure<char>(c, psj, (cnt + 1) * sizeof(StripeJob) + (nb_ub + 2) * 4)) return GSA_ERR_NOMEM;
		StripeJob *sj = psj.as<StripeJob>();
		i32 *b2j_all = (i32 *)(sj + cnt + 1);
		i64 dbytes = 128, bwords = 0; i32 nctr = NSEG; size_t b2j_used = 0;      // (ctr[0 .. NSEG-1]: launch tickets of the size classes)
		for (Seg &sg : seg) {
			sg.b2j = b2j_all + b2j_used; sg.nblocks = 0;
			for (size_t k = sg.b; k < sg.e; k++) {
				const LgJob &g = large[k];
				const i64 cells = (((i64)g.n + 63) / 64) * (i64)DP_STRIPE_BYTES(g.m);   // stripe-local direction nibbles
				StripeJob s; s.job = g.job; s.m = g.m; s.n = g.n; s.P = (g.n + 63) / 64;
				s.diroff = dbytes; dbytes += cells + 128;
				s.bndoff = bwords; bwords += (i64)(s.P - 1) * g.m;
				s.ctr = nctr++; s.first_block = sg.nblocks;
				for (int b = 0; b < ((s.P + 1) / 2 + sg.wpb - 1) / sg.wpb; b++) sg.b2j[sg.nblocks++] = (i32)(k - first);
				sj[k - first] = s;
			}
			b2j_used += (size_t)sg.nblocks;
		}
		const size_t last = first + cnt;
		uint8_t *dir = dev_ensure<uint8_t>(c, c->d_scan2, (size_t)dbytes + 512);
		const size_t bnd_cap0 = c->d_dp_bnd.cap;
		u32 *bnd = dev_ensure<u32>(c, c->d_dp_bnd, (size_t)bwords + 64);
		const size_t ctr_cap0 = c->d_dp_ctr.cap;
		u32 *ctr = dev_ensure<u32>(c, c->d_dp_ctr, (size_t)nctr + 64);
		if (!dir || !bnd || !ctr) return GSA_ERR_NOMEM;
		// boundary granules carry the launch epoch as their tag: cleared only when the buffer is new or the epoch wraps
		c->dp_epoch = (c->dp_epoch + 1) & 0xffffu;
		if (c->dp_epoch == 0 || c->d_dp_bnd.cap != bnd_cap0) { GSA_CHECK(c, hipMemsetAsync(bnd, 0, c->d_dp_bnd.cap, st)); if (c->dp_epoch == 0) c->dp_epoch = 1; }
		// (the ticket counters are put back to zero by the wave that draws the last ticket; the error word lives in the mailbox)
		if (c->d_dp_ctr.cap != ctr_cap0 || c->dp_dirty) { GSA_CHECK(c, hipMemsetAsync(ctr, 0, c->d_dp_ctr.cap, st)); GSA_CHECK(c, hipMemsetAsync(mail + err_slot, 0, 4, st)); c->dp_dirty = false; }
		// (the two classes back to back on one stream.  Side by side on two streams -- the few long jobs at raised priority --
		//  was measured at 250 Mb: same step time, the refinement passes beside them starve instead: the chip is busy either way)
		// (option dp_side: the lower class on a stream of its own -- it then starts with the upper one instead of behind it; the classes share
		//  nothing but the error word: tickets per class, direction / boundary bytes per job)
		const bool side = c->opt.dp_side && c->stream_aux[3] && seg[NSEG - 1].nblocks > 0 && (seg[0].nblocks > 0 || seg[1].nblocks > 0 || seg[2].nblocks > 0);
		if (side) { GSA_CHECK(c, hipEventRecord(c->ev[24], st)); GSA_CHECK(c, hipStreamWaitEvent(c->stream_aux[3], c->ev[24], 0)); }
		hipStream_t st_main = st;
		for (int si = 0; si < NSEG; si++) {
			const Seg &sg = seg[si];
			if (sg.nblocks == 0) continue;
			hipStream_t st = (side && si == NSEG - 1) ? c->stream_aux[3] : st_main;
			if (sg.wpb == 4) hipLaunchKernelGGL(k_dp_stripe<4>, dim3((unsigned)sg.nblocks), dim3(256), sg.dyn_lds, st, (const i32 *)sg.b2j, (const StripeJob *)sj, pool1, off1, pool2, off2, dir + 256, bnd, ctr, rev, ops, ops_off, ops_len, c->dp_epoch, (i32)(sg.mpad * 4), (i32)sg.lds_rows, (u32 *)(mail + err_slot), si);
			else hipLaunchKernelGGL(k_dp_stripe<1>, dim3((unsigned)sg.nblocks), dim3(64), sg.dyn_lds, st, (const i32 *)sg.b2j, (const StripeJob *)sj, pool1, off1, pool2, off2, dir + 256, bnd, ctr, rev, ops, ops_off, ops_len, c->dp_epoch, (i32)(sg.mpad * 4), (i32)sg.lds_rows, (u32 *)(mail + err_slot), si);
		}
		if (side) { GSA_CHECK(c, hipEventRecord(c->ev[25], c->stream_aux[3])); GSA_CHECK(c, hipStreamWaitEvent(st, c->ev[25], 0)); }
		GSA_CHECK(c, hipGetLastError());
		DPT(GSA_CHECK(c, hipStreamSynchronize(st)); if (cnt == 1) { u32 hh[6]; hipMemcpy(hh, ctr + sj[0].ctr + 40, 24, hipMemcpyDeviceToHost); fprintf(stderr, "[dp] %d x %d: fwd0 %.1f us  fwdlast %.1f us  traceback %.1f us (tiles %u runs %u)  total %.1f us\n", sj[0].m, sj[0].n, hh[0] * 0.01, hh[1] * 0.01, hh[2] * 0.01, hh[3], hh[4], hh[5] * 0.01); })
		if (last < large.size()) {
			// the staging buffer and the direction bytes are reused by the next batch
			i32 *h = c->h_mail;
			GSA_CHECK(c, hipMemcpyAsync(h, mail, MAIL_N * sizeof(i32), hipMemcpyDeviceToHost, st));
			GSA_CHECK(c, hipStreamSynchronize(st));
			if (h[err_slot]) { c->dp_dirty = c->dp_timeout = true; return gsa_fail(c, GSA_ERR_STATE, "internal: DP stripe hand-off timed out"); }
		}
		first = last;
	}
	return GSA_OK;
}

// All pointers are device pointers; the job count sits in mail[M_NJOB] (<= n_ub).  Jobs that do not fit
// the small kernel are processed in batches so that the direction bytes of one batch fit the budget.
// Returns with the work enqueued: errors of the last batch land in the mailbox (M_DPERR, M_DPERR2).
int run_ksw2_jobs(gsa_ctx *c, i32 n_ub, const uint8_t *pool1, const i64 *off1, const i32 *len1,
                  const uint8_t *pool2, const i64 *off2, const i32 *len2, uint8_t *ops, const i64 *ops_off, i32 *ops_len, i64 ops_total, Ksw2Launch *out,
                  const i32 *jfrag, gsa_frag *frag, bool mail_clean)
{
	*out = Ksw2Launch();
	if (n_ub <= 0) return GSA_OK;
	hipStream_t st = c->stream;
	i32 *mail = c->d_mail.as<i32>();
	i32 *d_order = dev_ensure<i32>(c, c->d_flag2, (size_t)n_ub + 2);
	i32 *d_order_tiny = dev_ensure<i32>(c, c->d_dp_tiny, (size_t)n_ub + 2);
	i32 *d_lg = dev_ensure<i32>(c, c->d_dp_large, 4 * ((size_t)n_ub + 1));      // (job, m, n) triples of the large jobs, then one flag per job
	i32 *d_jlarge = d_lg ? d_lg + 3 * ((size_t)n_ub + 1) : nullptr;
	uint8_t *rev = dev_ensure<uint8_t>(c, c->d_i64a, (size_t)ops_total + 64);
	if (!d_order || !d_order_tiny || !d_lg || !rev) return GSA_ERR_NOMEM;
	if (!pin_ensure<i32>(c, c->p_dp, (size_t)MAIL_N + 3 * LG_CHUNK)) return GSA_ERR_NOMEM;
	if (!mail_clean) GSA_CHECK(c, hipMemsetAsync(mail + M_DPERR, 0, 8 * sizeof(i32), st));                    // M_DPERR, M_NLARGE, M_DPERR2, -, M_CELLS (2 x u64): one aligned fill (28 bytes at an odd offset were three)
	i32 *h = c->p_dp.as<i32>();
	const size_t first_lg = (size_t)std::min<i64>(n_ub, LG_CHUNK);
	// Size classes below the striped kernel: alignments of at most GSA_DP_LANE cells (default 512; swept 256 .. 8192: profiles/archive/r03_dp_lane_sweep.txt) go one per LANE (k_dp_lane: every lane busy
	// on every instruction; a lane walks its cells one after the other, so the largest job of a launch is its latency floor --
	// 35 instructions per cell), the larger ones one per wavefront (k_dp_small).  GSA_DP_LANE=0: round 2's tiny / small split.
	const int dp_lane = c->opt.dp_lane;
	{ OpClassify op = { len1, len2, d_order, d_order_tiny, d_lg, d_jlarge, mail, h, (i32)first_lg, dp_lane }; int rc = lb_launch<2>(c, n_ub, op); if (rc) return rc; }
	GSA_CHECK(c, hipStreamSynchronize(st));
	if (h[M_LBERR]) return gsa_fail(c, GSA_ERR_STATE, "internal: look-back scan timed out");
	if (h[M_DPERR]) return gsa_fail(c, GSA_ERR_ARG, "DP job with an empty side");
	const i32 n = h[M_NJOB], nlarge = h[M_NLARGE], ntiny = h[M_NTINY], nsmall = n - nlarge - ntiny;
	out->n = n; out->nsmall = nsmall + ntiny; out->nlarge = nlarge;
	if (n <= 0) return GSA_OK;
	c->counters[5] += (u64)n;
	if ((size_t)nlarge > first_lg) {
		i32 keep[MAIL_N]; memcpy(keep, h, sizeof(keep));      // the mailbox copy must survive the reallocation (the caller reads it too)
		if (!pin_ensure<i32>(c, c->p_dp, (size_t)MAIL_N + 3 * (size_t)nlarge)) return GSA_ERR_NOMEM;
		h = c->p_dp.as<i32>(); memcpy(h, keep, sizeof(keep));
		GSA_CHECK(c, hipMemcpyAsync(h + MAIL_N, d_lg, (size_t)nlarge * 12, hipMemcpyDeviceToHost, st));
		GSA_CHECK(c, hipStreamSynchronize(st));
	}
	std::vector<LgJob> large((const LgJob *)(h + MAIL_N), (const LgJob *)(h + MAIL_N) + nlarge);
	hipEvent_t ev_fork = c->ev[10], ev_j2 = c->ev[12];
	// the many small jobs run on a second stream, concurrently with the striped ones
#ifdef GSA_EXPERIMENTS
	static const int dp_order = [] { const char *e = getenv("GSA_DP_ORDER"); return e ? atoi(e) : 0; }();      // 1 = tiny, small, then the stripes, one after the other on the caller's stream
	static const int dp_after_early = [] { const char *e = getenv("GSA_DP_AFTER_EARLY"); return e ? atoi(e) : 0; }();      // the small classes wait for the early striped launch
#else
	const int dp_order = 0, dp_after_early = 0;
#endif
	if (nsmall + ntiny > 0 && dp_lane > 0) {
		GSA_CHECK(c, hipEventRecord(ev_fork, st));
		GSA_CHECK(c, hipStreamWaitEvent(c->stream_aux[1], ev_fork, 0));
		if (dp_after_early && c->early_in_flight) GSA_CHECK(c, hipStreamWaitEvent(c->stream_aux[1], c->ev[14], 0));
		if (ntiny > 0) {
			const i64 tiles = ((i64)ntiny + LANE_TILE - 1) / LANE_TILE;
			const unsigned nwg = (unsigned)(tiles < LANE_WGS ? tiles : LANE_WGS);
			// direction dwords of one job at most: m * ceil(n / 8) over the shapes of the class (n <= 64, m + n - 1 <= 128, m * n <= dp_lane cells): 128 for
			// the default 512 cells, LANE_KMAX = 576 without a cell limit -- the arena was always sized for the latter: 604 MB per context instead of 134
			u32 kstride = 1;
			for (int nn = 1; nn <= 64; nn++) { int mm = 128 - nn + 1; if (dp_lane > 0 && dp_lane / nn < mm) mm = dp_lane / nn; if (mm < 1) continue; const u32 kd = (u32)mm * (u32)((nn + 7) / 8); if (kd > kstride) kstride = kd; }
			if (kstride > LANE_KMAX) kstride = LANE_KMAX;
			u32 *arena = dev_ensure<u32>(c, c->d_dp_arena, (size_t)nwg * 4 * 64 * kstride);
			if (!arena) return GSA_ERR_NOMEM;
			hipLaunchKernelGGL(k_dp_lane, dim3(nwg), dim3(256), 0, c->stream_aux[1], ntiny, d_order_tiny, 0, d_order, pool1, off1, len1, pool2, off2, len2, ops, ops_off, ops_len, jfrag, frag, arena, kstride);
		}
		if (nsmall > 0) {
			// (option dp_small_side: the one-per-wavefront kernel on a stream of its own instead of the caller's -- the late striped launch below then starts beside it
			//  instead of behind it; with 16 hardware queues a fifth stream per context no longer shares one)
			const bool own = c->opt.dp_small_side && c->stream_aux[3] && ntiny > 0;
			hipStream_t s2 = own ? c->stream_aux[3] : (ntiny > 0 ? st : c->stream_aux[1]);
			if (own) GSA_CHECK(c, hipStreamWaitEvent(s2, ev_fork, 0));
			const unsigned nb = (unsigned)((nsmall + SMALL_WAVES - 1) / SMALL_WAVES);
			hipLaunchKernelGGL(k_dp_small, dim3(nb), dim3(64 * SMALL_WAVES), 0, s2, nsmall, d_order, pool1, off1, len1, pool2, off2, len2, ops, ops_off, ops_len, jfrag, frag);
			if (ntiny > 0) { GSA_CHECK(c, hipEventRecord(c->ev[18], s2)); GSA_CHECK(c, hipStreamWaitEvent(c->stream_aux[1], c->ev[18], 0)); }
		}
		GSA_CHECK(c, hipGetLastError());
		GSA_CHECK(c, hipEventRecord(ev_j2, c->stream_aux[1]));
		out->small_in_flight = true;
	} else if (nsmall + ntiny > 0 && dp_order == 1) {
		if (ntiny > 0) {
			const unsigned nb = (unsigned)((ntiny + 4 * TINY_WAVES - 1) / (4 * TINY_WAVES));
			hipLaunchKernelGGL(k_dp_tiny, dim3(nb), dim3(64 * TINY_WAVES), 0, st, ntiny, d_order_tiny, pool1, off1, len1, pool2, off2, len2, ops, ops_off, ops_len, jfrag, frag);
		}
		if (nsmall > 0) {
			const unsigned nb = (unsigned)((nsmall + SMALL_WAVES - 1) / SMALL_WAVES);
			hipLaunchKernelGGL(k_dp_small, dim3(nb), dim3(64 * SMALL_WAVES), 0, st, nsmall, d_order, pool1, off1, len1, pool2, off2, len2, ops, ops_off, ops_len, jfrag, frag);
		}
		GSA_CHECK(c, hipGetLastError());
		GSA_CHECK(c, hipEventRecord(ev_j2, st));
		GSA_CHECK(c, hipStreamWaitEvent(c->stream_aux[1], ev_j2, 0));
		out->small_in_flight = true;
	} else
	if (nsmall + ntiny > 0) {
		GSA_CHECK(c, hipEventRecord(ev_fork, st));
		GSA_CHECK(c, hipStreamWaitEvent(c->stream_aux[1], ev_fork, 0));
		if (ntiny > 0) {
			const unsigned nb = (unsigned)((ntiny + 4 * TINY_WAVES - 1) / (4 * TINY_WAVES));
			hipLaunchKernelGGL(k_dp_tiny, dim3(nb), dim3(64 * TINY_WAVES), 0, c->stream_aux[1], ntiny, d_order_tiny, pool1, off1, len1, pool2, off2, len2, ops, ops_off, ops_len, jfrag, frag);
		}
		if (nsmall > 0) {
			// (the two size classes side by side: the one-per-wavefront kernel stays on the caller's stream, joined below.
			//  Not a stream of its own: the runtime maps streams onto four hardware queues, a fifth stream shares one --
			//  with the striped kernel, if it is unlucky)
			hipStream_t s2 = ntiny > 0 ? st : c->stream_aux[1];
			const unsigned nb = (unsigned)((nsmall + SMALL_WAVES - 1) / SMALL_WAVES);
			hipLaunchKernelGGL(k_dp_small, dim3(nb), dim3(64 * SMALL_WAVES), 0, s2, nsmall, d_order, pool1, off1, len1, pool2, off2, len2, ops, ops_off, ops_len, jfrag, frag);
			if (ntiny > 0) { GSA_CHECK(c, hipEventRecord(c->ev[18], s2)); GSA_CHECK(c, hipStreamWaitEvent(c->stream_aux[1], c->ev[18], 0)); }
		}
		GSA_CHECK(c, hipGetLastError());
		GSA_CHECK(c, hipEventRecord(ev_j2, c->stream_aux[1]));
		out->small_in_flight = true;
	}
	if (nlarge > 0) {
		// (large gaps are normally launched early, from the leaf table; whatever turns up here shares their buffers)
		if (c->early_in_flight) GSA_CHECK(c, hipStreamWaitEvent(st, c->ev[14], 0));
		int rc = launch_stripes(c, st, large, pool1, off1, pool2, off2, ops, ops_off, ops_len, rev, M_DPERR2);
		if (rc) return rc;
	}
	// (no join: the caller decides what else runs behind the small kernel on stream_aux[1]; event ev[12] marks its end)
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
	uint8_t *d_p1 = dev_ensure<uint8_t>(c, c->leaf[0], (size_t)p1 + 1), *d_p2 = dev_ensure<uint8_t>(c, c->leaf[1], (size_t)p2 + 1), *d_ops = dev_ensure<uint8_t>(c, c->leaf[2], (size_t)po + 1);
	i64 *d_o1 = dev_ensure<i64>(c, c->leaf[3], n), *d_o2 = dev_ensure<i64>(c, c->leaf[4], n), *d_oo = dev_ensure<i64>(c, c->leaf[5], n);
	i32 *d_l1 = dev_ensure<i32>(c, c->leaf[6], n), *d_l2 = dev_ensure<i32>(c, c->leaf[7], n), *d_ol = dev_ensure<i32>(c, c->leaf[8], n);
	if (!d_p1 || !d_p2 || !d_ops || !d_o1 || !d_o2 || !d_oo || !d_l1 || !d_l2 || !d_ol) return GSA_ERR_NOMEM;
	GSA_CHECK(c, hipMemcpyAsync(d_p1, pool1, p1, hipMemcpyHostToDevice, st)); GSA_CHECK(c, hipMemcpyAsync(d_p2, pool2, p2, hipMemcpyHostToDevice, st));
	GSA_CHECK(c, hipMemcpyAsync(d_o1, off1, n * 8, hipMemcpyHostToDevice, st)); GSA_CHECK(c, hipMemcpyAsync(d_o2, off2, n * 8, hipMemcpyHostToDevice, st));
	GSA_CHECK(c, hipMemcpyAsync(d_oo, ops_off, n * 8, hipMemcpyHostToDevice, st));
	GSA_CHECK(c, hipMemcpyAsync(d_l1, len1, n * 4, hipMemcpyHostToDevice, st)); GSA_CHECK(c, hipMemcpyAsync(d_l2, len2, n * 4, hipMemcpyHostToDevice, st));
	const i32 cnts[2] = { n_pairs, (i32)po };
	GSA_CHECK(c, hipMemcpyAsync(c->d_mail.as<i32>() + M_NJOB, cnts, 8, hipMemcpyHostToDevice, st));      // M_NJOB, M_OPSTOT
	Ksw2Launch kl;
	int rc = run_ksw2_jobs(c, n_pairs, d_p1, d_o1, d_l1, d_p2, d_o2, d_l2, d_ops, d_oo, d_ol, po, &kl);
	if (rc == GSA_OK && kl.small_in_flight) GSA_CHECK(c, hipStreamWaitEvent(st, c->ev[12], 0));
	if (rc == GSA_OK) {
		i32 err = 0;
		GSA_CHECK(c, hipMemcpyAsync(ops, d_ops, po, hipMemcpyDeviceToHost, st));
		GSA_CHECK(c, hipMemcpyAsync(ops_len, d_ol, n * 4, hipMemcpyDeviceToHost, st));
		GSA_CHECK(c, hipMemcpyAsync(&err, c->d_mail.as<i32>() + M_DPERR2, 4, hipMemcpyDeviceToHost, st));
		GSA_CHECK(c, hipStreamSynchronize(st));
		if (err) { c->dp_dirty = c->dp_timeout = true; rc = gsa_fail(c, GSA_ERR_STATE, "internal: DP stripe hand-off timed out"); }
	}
	return rc;
}
