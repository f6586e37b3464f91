// gsalign_amd/csrc/k_seed.hip -- stage 1: seed exploration (a4, a5), locate (a3),
// ordering by (PosDiff, qPos) and SeedGrouping (a6).
//
// Replaces IdentifyLocalMEM + BWT_Search + bwt_sa + SeedGrouping
// (reference src/GSAlign.cpp:51-107,126-143; src/bwt_search.cpp:121-185).
#include <cstring>
#include <mutex>
#include <condition_variable>
#include <vector>
#include "gsa_ctx.h"
#include "gsa_fm.h"
#include "gsa_scan.h"

enum { CNT_OCCBLK = 0, CNT_DONE = 1, CNT_HITS = 2, CNT_SEEDS = 3, CNT_DPCELLS = 4, CNT_DPJOBS = 5, CNT_DPMN = 6, CNT_CAND = 8, CNT_OVERFLOW = 9, CNT_OCCBLK_ALL = 10, CNT_HEAVY = 12 };

#ifndef SEED_WG
#define SEED_WG 64              // lanes per chunk: ONE WAVE (round 3, late: 128 lanes = two waves per chunk spent 27 % more VALU wave-instructions on the same
                                // searches -- a wave iterates as long as its slowest lane, and the tail of a chunk, a few lanes deep in the Occ walk of
                                // a repeat copy, kept both waves turning; profiles/archive/r03_seed_shape_sweep.txt.  256 lanes before that: 0.157 -> 0.139 ms)
#endif
#ifndef NSUB
#define NSUB 96                // speculative sub-ranges per chunk (work items of the workgroup, drawn from an LDS queue: 64 lanes take the first 64, a lane
                               // that is through early takes one of the other 32).  Swept 64 .. 512 on the bench: with 64-base text windows and four
                               // presence bits per round trip fewer, longer walks win -- 384 was best before them, 128 with two waves per chunk)
#endif
#ifndef ADV_STEPS
#define ADV_STEPS 4             // advance steps (hops over memoised / ambiguous positions, opening a search) per round trip
#endif
#ifndef PLOOK
#define PLOOK 3                 // presence bits looked up beside the one of the current start
#endif
#define PATH_WORDS 320          // 10240 on-path bits per chunk
#define QP_WORDS (GSA_CHUNK / 16 + 4)
#define QN_WORDS (GSA_CHUNK / 32 + 4)
enum { M_DONE = 0, M_FM = 1, M_TEXT = 2, M_KMER = 3, M_LOC = 4, M_ADV = 5, M_KLO = 6, M_MLOC = 7, M_MTEXT = 8 };
#ifndef SEED_MULTI
#define SEED_MULTI 1            // intervals of at most this many rows (<= 4) are finished by text comparison of all their rows (seed_chunk, round 4); 1 = off.
#endif                          // MEASURED at 4 (parity green, 50 cases): 28 more VGPRs (168 + spills), seed stage of a 250 Mb contig 3.03 -> 3.27 ms, of the
                                // full human set 55.2 -> 54.0 ms: the Occ steps that weigh are the walks through repeat copies with thousands of rows, not the
                                // last two steps of a chance interval.  Off in production; kept behind this switch.

// ---- 2-bit packed sequences: base p sits at bits (2*(p&15)) of word p>>4 (LSB first) ----
__device__ __forceinline__ int q_code(const u32 *qp, int p) { return (qp[p >> 4] >> ((p & 15) << 1)) & 3; }
__device__ __forceinline__ int q_isn(const u32 *qn, int p) { return (qn[p >> 5] >> (p & 31)) & 1; }
__device__ __forceinline__ u64 funnel64(u32 w0, u32 w1, u32 w2, int sh)      // 64 bits starting sh (even, < 32) bits into w0
{
	const u64 lo = (u64)w0 | ((u64)w1 << 32);
	return sh ? (lo >> sh) | ((u64)w2 << (64 - sh)) : lo;
}
__device__ __forceinline__ u64 q_bits64(const u32 *qp, int p) { const int w = p >> 4; return funnel64(qp[w], qp[w + 1], qp[w + 2], (p & 15) << 1); }
__device__ __forceinline__ u32 q_nbits32(const u32 *qn, int p) { const int w = p >> 5; return (u32)((((u64)qn[w + 1] << 32) | qn[w]) >> (p & 31)); }

// Unique interval (x2 == 1): how many of the next (at most 32) query bases continue the only
// occurrence, i.e. pos+t < clen, tp+t < tend, query base t unambiguous and equal to text base t.
// Equivalent to that many successful bwt_2occ4 steps, which leave x0 and x2 = 1 unchanged
// (see DESIGN.md section 4).  r0..r2 = packed reference words starting at word tp>>4.
__device__ __forceinline__ int text_match32(u32 r0, u32 r1, u32 r2, i64 tp, i64 tend, const u32 *qp, const u32 *qn, int pos, int clen)
{
	int avail = clen - pos;
	if (tend - tp < (i64)avail) avail = (int)(tend - tp);
	if (avail > 32) avail = 32;
	if (avail <= 0) return 0;
	const u64 d = funnel64(r0, r1, r2, (int)(tp & 15) << 1) ^ q_bits64(qp, pos);
	const u64 dm = (d | (d >> 1)) & 0x5555555555555555ull;
	const u32 nm = q_nbits32(qn, pos);
	int n = dm ? (__ffsll((unsigned long long)dm) - 1) >> 1 : 32;
	const int fn = nm ? __ffs((int)nm) - 1 : 32;
	n = n < fn ? n : fn;
	return n < avail ? n : avail;
}

// Presence table: does a pres_k-mer occur in the indexed text?  BWT_Search from s reaches MinSeedLength iff the first
// MinSeedLength bases occur, so an absent pres_k-mer (pres_k <= MinSeedLength) settles a search that yields no seed with ONE read.
// GROUPED layout (round 3): a walk crosses the ~14 starts in front of a mismatch one by one, so the kernel asks about s, s+1,
// s+2, s+3 together -- as a plain bitmap indexed by the k-mer those were four reads of four unrelated cache lines (most of the
// seed kernel's 6.2 GB of fetches per 250 Mb contig, profiles/archive/r02_pmc_human.json).  The four k-mers share the K-3 bases
// q[s+3 .. s+K): that CORE selects a 32-byte line, and bit 64 i + e_i of the line answers for start s+i, where e_i (6 bits) are
// the three bases of that k-mer outside the core -- q[s+i .. s+3) and q[s+K .. s+K+i).  A k-mer of the text is therefore entered
// four times, once per role i.  4^(K-3) lines: 512 MiB for K = 15.  (pres4_line / pres4_bits take the query's 2-bit window from
// the group's first base; the builder derives the same numbers from the k-mer alone.)
__device__ __forceinline__ u32 pres4_line(u64 qb, int K) { return (u32)((qb >> 6) & ((1ull << (2 * (K - 3))) - 1)); }
__device__ __forceinline__ u32 pres4_bit(u64 qb, int K, int i)      // 0 .. 255: position inside the line for start s + i
{
	const u32 head = (u32)(qb >> (2 * i)) & ((1u << (2 * (3 - i))) - 1);          // q[s+i .. s+3)
	const u32 tail = (u32)(qb >> (2 * K)) & ((1u << (2 * i)) - 1);                // q[s+K .. s+K+i)
	return (u32)i * 64u + (head | (tail << (2 * (3 - i))));
}

// ---------------------------------------------------------------------------
// Seed exploration.  Work unit = one 10 000-bp chunk (absolute position, App. B
// #1).  Inside a chunk the reference walks a CHAIN: next start = start+len+1
// after an accepted match (start+5 with -sen), start+1 otherwise, so the walk is
// sequential -- but "next start" is a pure function of the start position, i.e.
// the chunk is a functional graph whose paths merge (two walks that are inside
// the same exact match end at the same mismatch).  One 256-lane workgroup owns a
// chunk and cuts it into NSUB sub-ranges.  Round 1: lanes pull sub-ranges from an
// LDS queue and walk each one speculatively from its left edge, memoising next(s)
// per position.  Later rounds: a sub-range whose true entry (= the previous
// sub-range's exit) differs from where it was entered is re-walked along the memo
// (new searches only until it merges) until no exit moves.  The result is exactly
// the reference's chain; the on-path bit per position selects which memoised
// matches become seeds.  Query (2-bit packed + N bitmap), memo, exits and on-path
// bits live in LDS.
// ---------------------------------------------------------------------------
// Exclusive prefix of the per-chunk hit counts (n1 = chunks + 1 entries, the last one is 0), by the workgroup that is through
// LAST in a seed kernel: the counts were stored with agent-scope atomics and are read the same way (the other workgroups ran on
// other XCDs), 8 or 16 loads in flight per lane.  Was a rocPRIM scan behind the kernel: two more GPU operations per contig.
template <int TPB>
__device__ __forceinline__ void wg_exscan_hits(const i32 *hits, i32 *base, int n1)
{
	constexpr int V = TPB <= 64 ? 16 : 8;             // loads in flight per lane
	__shared__ i32 s_ws[TPB / 64], s_run;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	if (tid == 0) s_run = 0;
	__syncthreads();
	for (int b0 = 0; b0 < n1; b0 += TPB * V) {
		i32 v[V], tsum = 0;
#pragma unroll
		for (int k = 0; k < V; k++) { const int idx = b0 + tid * V + k; v[k] = idx < n1 ? __hip_atomic_load(&hits[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0; }
#pragma unroll
		for (int k = 0; k < V; k++) tsum += v[k];
		i32 inc = tsum;
		for (int o = 1; o < 64; o <<= 1) { const i32 t = __shfl_up(inc, o); if (lane >= o) inc += t; }
		if (lane == 63) s_ws[wv] = inc;
		__syncthreads();
		i32 wo = 0, tot = 0;
		for (int w = 0; w < TPB / 64; w++) { const i32 x = s_ws[w]; if (w < wv) wo += x; tot += x; }
		i32 e = s_run + wo + inc - tsum;
#pragma unroll
		for (int k = 0; k < V; k++) { const int idx = b0 + tid * V + k; if (idx < n1) base[idx] = e; e += v[k]; }
		__syncthreads();
		if (tid == 0) s_run += tot;
		__syncthreads();
	}
}

// The same by ONE WAVE of a workgroup whose other waves are busy with chunks of their own (k_seed_wg with SEED_WPW > 1): no LDS, no workgroup barrier.
__device__ __forceinline__ void wave_exscan_hits(const i32 *hits, i32 *base, int n1)
{
	constexpr int V = 16;
	const int lane = threadIdx.x & 63;
	i32 run = 0;
	for (int b0 = 0; b0 < n1; b0 += 64 * V) {
		i32 v[V], tsum = 0;
#pragma unroll
		for (int k = 0; k < V; k++) { const int idx = b0 + lane * V + k; v[k] = idx < n1 ? __hip_atomic_load(&hits[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0; }
#pragma unroll
		for (int k = 0; k < V; k++) tsum += v[k];
		i32 inc = tsum;
		for (int o = 1; o < 64; o <<= 1) { const i32 t = __shfl_up(inc, o); if (lane >= o) inc += t; }
		i32 e = run + inc - tsum;
#pragma unroll
		for (int k = 0; k < V; k++) { const int idx = b0 + lane * V + k; if (idx < n1) base[idx] = e; e += v[k]; }
		run += __shfl(inc, 63);
	}
}

// next(s) - s per position of a chunk, 0 = unknown, as NIBBLES (round 3; bytes in round 2, u16 before): hops of 15 and more --
// every accepted match, a few hundred per chunk counting the speculative walks -- keep their value in a hash table beside the
// nibbles.  LDS per workgroup is what limits how many chunks a CU works on at once (and with them the random reads in flight):
// 18.7 KB -> 15.7 KB = ten workgroups per CU instead of eight.  Lanes set different nibbles of one word at the same time
// (atomic OR); a position is only ever given ONE value (next(s) is a function of s), so setting it twice is harmless.
#ifndef SEED_MIN_WAVES
#define SEED_MIN_WAVES 5        // waves per SIMD the register allocation must allow: 5 = 96 VGPRs (37 of them spilled into 152 bytes of scratch; the loop wants ~150).  Round 6: what the
                                // kernel leaves FREE on a CU is worth more than what the spills cost it -- see k_seed_wg.  (3 until round 6: 149 VGPRs, no scratch.)
#endif
#ifndef LHOP_N
#define LHOP_N 1024
#endif
// Round 4: two bits per position.  next(s) - s only ever takes three kinds of value -- 1 (no seed from s), 5 (an accepted match under -sen:
// GSAlign.cpp:88-91) and len + 1 >= MinSeedLength + 1 (an accepted match) -- so the codes are 0 unknown, 1 -> +1, 2 -> +5, 3 -> the hop sits in the
// hash table; anything else (hops of 2-4, 6-...: the accounting build, MinSeedLength below 5) goes to the table as well.  5 KB -> 2.5 KB per chunk:
// LDS x time is what the seed kernel costs the chip (a CU's LDS holds its chunks and nothing else's meanwhile), see DESIGN section 4.
#define MEMO_WORDS (GSA_CHUNK / 16)
__device__ __forceinline__ int memo_nib(const u32 *memo, int s) { return (int)((memo[s >> 4] >> ((s & 15) << 1)) & 3u); }      // the code: 0 = unknown
__device__ __forceinline__ void memo_one(u32 *memo, int s) { atomicOr(&memo[s >> 4], 1u << ((s & 15) << 1)); }
__device__ __forceinline__ int memo_get(const u32 *memo, const u32 *lhop, int s)
{
	const int v = memo_nib(memo, s);
	if (v < 2) return v;
	if (v == 2) return 5;
	for (u32 h = ((u32)s * 40503u) >> 6;; h++) { const u32 e = lhop[h & (LHOP_N - 1)]; if ((e >> 16) == (u32)s + 1) return (int)(e & 0xffffu); }
}
__device__ __forceinline__ void memo_set(u32 *memo, u32 *lhop, int s, int d, int *abort_flag)
{
	if (d == 1 || d == 5) { atomicOr(&memo[s >> 4], (d == 1 ? 1u : 2u) << ((s & 15) << 1)); return; }
	const u32 e = ((u32)(s + 1) << 16) | (u32)d;
	u32 h = ((u32)s * 40503u) >> 6;
	for (int tries = 0; tries < LHOP_N; tries++, h++) {
		const u32 old = atomicCAS(&lhop[h & (LHOP_N - 1)], 0u, e);
		if (old == 0 || old == e) { atomicOr(&memo[s >> 4], 3u << ((s & 15) << 1)); return; }      // (two walks that reach the same start store the same hop: next(s) is a function of s)
	}
	*(volatile int *)abort_flag = 1;                                // table full: the chunk is redone by the dense kernels
}

// Four ASCII bases (one dword, first base in the low byte) -> their nt4 codes packed LSB first in bits 0-7 (an ambiguous base: code 0,
// as gsa_nt4's 4 & 3) | the "ambiguous" flags of the four in bits 8-11.  gsa_nt4 (nst_nt4_table, bntseq.c:40-57) byte by byte costs
// ~17 VALU instructions per base; staging a 10 000-base chunk that way was a tenth of the seed kernel's instructions.  Here: fold the
// case, code = ((b >> 1) & 3) ^ (its own high bit) -- a 0, c 1, g 3 ^ 1 = 2, t 2 ^ 1 = 3 -- look the letter that code stands for up with
// one v_perm and compare: anything that is not that letter is ambiguous.  ~5 instructions per base.
__device__ __forceinline__ u32 nt4_quad(u32 w)
{
	const u32 l = w | 0x20202020u;
	const u32 x = (l >> 1) & 0x03030303u;
	u32 code = x ^ ((x >> 1) & 0x01010101u);
	const u32 want = __builtin_amdgcn_perm(0u, 0x74676361u, code);               // bytes 'a' 'c' 'g' 't' selected by the four codes
	const u32 diff = l ^ want;
	const u32 bad = ((diff | ((diff & 0x7f7f7f7fu) + 0x7f7f7f7fu)) >> 7) & 0x01010101u;      // 1 per byte that is not the letter of its code
	code &= ~(bad * 3u);
	return ((code * 0x01041040u) >> 24) | (((bad * 0x01020408u) >> 24) & 15u) << 8;
}

// 32 bases from `src` (position p0 of a chunk of clen bases; behind the chunk: N) -> two words of 2-bit codes + the word of their N flags
__device__ __forceinline__ void stage32(const uint8_t *src, int p0, int clen, u32 &w0, u32 &w1, u32 &wn)
{
	u32 d[8];
	if (p0 + 32 <= clen) { const uint4 a = *(const uint4 *)src, b = *(const uint4 *)(src + 16); d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w; }
	else {
#pragma unroll
		for (int t = 0; t < 8; t++) { d[t] = 0; for (int k = 0; k < 4; k++) d[t] |= (u32)(p0 + 4 * t + k < clen ? src[4 * t + k] : (uint8_t)'N') << (8 * k); }
	}
	w0 = w1 = wn = 0;
#pragma unroll
	for (int t = 0; t < 8; t++) {
		const u32 r = nt4_quad(d[t]);
		if (t < 4) w0 |= (r & 0xffu) << (8 * t); else w1 |= (r & 0xffu) << (8 * (t - 4));
		wn |= (r >> 8) << (4 * t);
	}
}

// NCH chunks per wave (round 4; the round-3 verdict's "fill the lanes": a wave that owns TWO chunk states).  What the wave-iterations of
// ONE chunk hold on repeat-bearing text: 75 iterations with 14.7 of 64 lanes active (-DSEED_STATS) -- the body is over after ~25
// iterations, the rest is a handful of lanes deep in the Occ walks of repeat copies.  With NCH = 2 a wave owns two neighbouring chunks:
// 2 x 96 sub-ranges in one queue, the tails of the two chunks overlap in time.  Everything a chunk owns in LDS exists per chunk
// ([NCH][...]; 30 KB per pair: five pairs per CU = the same ten chunks in flight); an item is a (chunk, sub-range) pair, v = ch * NSUB +
// sub-range; the chains of the two chunks never touch (a match stops at its chunk's end and IdentifyLocalMEM restarts at every chunk,
// GSAlign.cpp:61-94), so the resolver works on both at once: two roots, exits past a chunk's end terminate.  Parity green (stage-1
// goldens, both layouts, 48 cases).  MEASURED (profiles/archive/r04_seed_nch.txt, 250 Mb bench workload, kernel alone): wave-iterations 1.87 M ->
// 1.23 M (-34 %), active lanes 14.7 -> 21.1 -- and the kernel 3.02 -> 3.79 ms, four contexts 30.3 -> 28.2 Gbp/s.  A wave-iteration takes
// ~4 us on the full chip whatever the number of waves per SIMD (1.25 or 2.5): it is a dependent random read over 20 GB of tables (a TLB
// walk each), not instruction issue, so what counts is requests in flight = waves x active lanes, and half the waves with 1.4x the lanes
// is fewer.  PRODUCTION STAYS AT NCH = 1 (SEED_NCH); the pair form is kept for the day the LDS per chunk halves (then ten PAIRS fit a CU).
// The accounting build (COUNT) keeps one chunk per wave in any case (its per-start Occ-block array is 20 KB per chunk).
// Everything a wave keeps in LDS for its chunk(s).  One object per WAVE: a workgroup of k_seed_wg is SEED_WPW independent waves (round 6), each with
// its own chunk, queue, memo and resolver state; nothing in seed_chunk synchronises across waves.
template <bool COUNT, int NCH>
struct SeedLds {
	static constexpr int NV = NCH * NSUB;
	u32 s_ncand[NCH], s_queue, s_hits[NCH];
	int changed, s_abort;
	u32 qp[NCH][QP_WORDS], qn[NCH][QN_WORDS];
	// next(s) - s per position as 2-bit codes + a hash table for the long hops (memo_get / memo_set above); a full table (never
	// seen) sends the chunk to the dense kernels like an exhausted budget does.
	u32 memo[NCH][MEMO_WORDS];
	u32 lhop[NCH][LHOP_N];            // (s + 1) << 16 | hop, 0 = free
	uint16_t mblk[COUNT ? GSA_CHUNK : 1];   // Occ blocks the search from s read (accounting build only: NCH = 1)
	u32 bits[NCH][PATH_WORDS];
	uint16_t entry_of[NV], exit_of[NV];
	uint16_t pend_it[SEED_WG];                          // items to walk for real in this pass
	uint16_t jmp[2][NV], walked_from[NV];               // pointer-jumping buffers; entry of the last real walk of a re-walked sub-range
	u32 rewalked[(NV + 31) / 32], onchain[(NV + 31) / 32], s_npend;
	int s_last;
};
// A wave's own synchronisation point.  One wave per workgroup: __syncthreads() as ever (the barrier itself is elided for a one-wave workgroup, the fences
// stay).  Several: the waves are independent, and LDS operations of ONE wave are performed in the order they were issued, so all that is needed is that
// the compiler keeps that order.
#define SEED_SYNC() do { if (WPW == 1) __syncthreads(); else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } } while (0)
#define s_ncand sl_.s_ncand
#define s_queue sl_.s_queue
#define s_hits sl_.s_hits
#define changed sl_.changed
#define s_abort sl_.s_abort
#define qp sl_.qp
#define qn sl_.qn
#define memo sl_.memo
#define lhop sl_.lhop
#define mblk sl_.mblk
#define bits sl_.bits
#define entry_of sl_.entry_of
#define exit_of sl_.exit_of
#define pend_it sl_.pend_it
#define jmp sl_.jmp
#define walked_from sl_.walked_from
#define rewalked sl_.rewalked
#define onchain sl_.onchain
#define s_npend sl_.s_npend
#define s_last sl_.s_last
template <bool COUNT, bool E16, int NCH, int WPW>
__device__ __forceinline__ void seed_chunk(const DevIndex &di, const uint8_t *__restrict__ q, i32 qlen, const Params &prm, u64 *cnt,
                                                      i32 *cand_s, i32 *cand_len, u64 *cand_x0, i32 *cand_freq, u32 cand_cap, u32 *cand_cnt, u32 *onpath, i32 *chunk_hits, u64 *hcnt,
                                                      u32 budget, u32 *heavy_list, i32 *chunk_base, const int chunk0, const u32 n_chunks, SeedLds<COUNT, NCH> &sl_)
{
	constexpr int NV = NCH * NSUB;                     // virtual items
	static_assert(!COUNT || NCH == 1, "the accounting build walks one chunk per wave");
	const int j = threadIdx.x & 63;      // (lane: a workgroup is SEED_WPW independent waves)
	const int nch = (u32)chunk0 + NCH <= n_chunks ? NCH : (int)(n_chunks - (u32)chunk0);      // chunks of this pair that exist (the last pair of an odd contig: one)
	i64 c0[NCH]; int clen[NCH], S[NCH], nitems[NCH]; size_t cbase[NCH];
#pragma unroll
	for (int ch = 0; ch < NCH; ch++) {
		c0[ch] = (i64)(chunk0 + ch) * GSA_CHUNK;
		clen[ch] = ch < nch ? (int)((i64)qlen - c0[ch] < GSA_CHUNK ? (i64)qlen - c0[ch] : GSA_CHUNK) : 0;
		S[ch] = (clen[ch] + NSUB - 1) / NSUB; if (S[ch] < 1) S[ch] = 1;       // sub-range length
		nitems[ch] = (clen[ch] + S[ch] - 1) / S[ch];
		cbase[ch] = (size_t)(chunk0 + ch) * cand_cap;                          // this chunk's private candidate segment
	}
	const int nitems_all = NCH == 1 ? nitems[0] : nitems[0] + nitems[NCH - 1];
#define CH_SEL(ARR, CH) (NCH == 1 ? ARR[0] : ((CH) ? ARR[NCH - 1] : ARR[0]))
	// stage the chunks: 32 bases per lane per pass -> two code words + one N word (16-byte global loads)
#pragma unroll
	for (int ch = 0; ch < NCH; ch++) {
		for (int g = j; g < QN_WORDS; g += SEED_WG) {
			u32 w0 = 0, w1 = 0, wn = 0;
			const int p0 = g << 5;
			if (p0 < clen[ch]) stage32(q + c0[ch] + p0, p0, clen[ch], w0, w1, wn);
			if (2 * g < QP_WORDS) qp[ch][2 * g] = w0;
			if (2 * g + 1 < QP_WORDS) qp[ch][2 * g + 1] = w1;
			qn[ch][g] = wn;
		}
		for (int p = j; p < MEMO_WORDS; p += SEED_WG) memo[ch][p] = 0;
		for (int p = j; p < LHOP_N; p += SEED_WG) lhop[ch][p] = 0;
		for (int p = j; p < PATH_WORDS; p += SEED_WG) bits[ch][p] = 0;
		for (int it = j; it < nitems[ch]; it += SEED_WG) entry_of[ch * NSUB + it] = (uint16_t)(it * S[ch]);
		if (j == 0) { s_ncand[ch] = 0; s_hits[ch] = 0; }
	}
	if (j == 0) { s_queue = 0; s_npend = 0; s_abort = 0; }
	if (j < (NV + 31) / 32) rewalked[j] = 0;
	u32 all_blocks = 0, rounds = 0, iters = 0;
#ifdef SEED_STATS
	u32 st_it = 0, st_fm_any = 0, st_fm_only = 0, st_act = 0, st_fm = 0;
#endif
	unsigned long long t_begin = wall_clock64(), t_round0 = 0, t_resolve = 0;
	u32 dirty = 0;                                      // rounds >= 2: this lane has one item (fb_item) to walk for real
	int fb_item = 0;
	SEED_SYNC();
	for (;;) {
		// One flat loop per wave.  Every iteration each lane has exactly ONE memory request pending
		// (two Occ blocks / 12 bytes of packed reference text / a k-mer table entry / an SA entry); all
		// lanes issue their requests together, wait once, then consume by mode -- so lanes that are in
		// different searches, or in different phases of a search, never serialise on each other's
		// memory latency.
		int item = -1, s = 0, bend = 0, pos = 0, mode = M_ADV; u32 kid = 0, pid = 0, pext = 0;
		int lclen = 0;                                        // length of the chunk my item belongs to
		const u32 *qp_l = qp[0], *qn_l = qn[0]; u32 *memo_l = memo[0], *lhop_l = lhop[0];      // ... and that chunk's arrays
		FmIntv ik = {0, 0, 0}; u32 blk = 0; i64 tp = 0;
		// Round 4: an interval of 2 .. SEED_MULTI rows is finished WITHOUT further Occ steps.  Its rows are suffix-array rows x0 .. x0 + x2 - 1,
		// sorted by suffix; extending the match by base c keeps the rows whose text continues with c -- a contiguous run, and the reference's
		// new x0 is the old one plus the rows in front of that run (bwt_search.cpp:159-165: ok[3] = A first, then C, G, T; the row whose suffix
		// ends is the `primary` adjustment).  So: read the rows' text positions from the dense SA (one round trip), compare every row's text with
		// the query 64 bases per round trip, and the match ends where the longest row ends: len = the longest common prefix, the final
		// interval = the rows that reach it.  Against a human-sized index most 15-mers that occur have 2-8 rows by chance (6.2 G rows over 4^15
		// k-mers) and took 2-3 Occ steps + a locate to get to the text comparison.
		i64 mp[4] = {0, 0, 0, 0}; u32 malive = 0;          // text positions of the rows at `pos`; the rows still matching (bit i: row x0 + i)
		bool need_item = true;
		while (!__all(mode == M_DONE)) {
			iters++;
			// A chunk whose walks exceed the budget (a tandem array with more than MaxSeedFreq copies: every start is searched for
			// ~100 bases, rejected and followed by start+1 -- thousands of dependent searches on a handful of lanes) is given up
			// here and searched from EVERY position in parallel by the dense kernels below (with its partner).
			if (!COUNT && budget && iters > budget) *(volatile int *)&s_abort = 1;
			if (*(volatile int *)&s_abort) break;
#ifdef SEED_STATS      // (experiments: what the wave-iterations are spent on -- sums over all waves instead of the maxima / timers)
			{ const u64 bf = __ballot(mode == M_FM), ba = __ballot(mode != M_DONE);
			  st_it++; st_fm_any += bf != 0; st_fm_only += bf != 0 && bf == ba; st_act += __popcll(ba); st_fm += __popcll(bf); }
#endif
			// ---- request phase (convergent) ----
			u64 kk = 0, ll = 0; bool kn = true, ln = true;
			if (mode == M_FM) {
				const u64 k = ik.x1 - 1, l = ik.x1 - 1 + ik.x2;
				kn = (k == (u64)-1); ln = (l == (u64)-1);
				kk = kn ? 0 : k - (k >= di.primary); ll = ln ? 0 : l - (l >= di.primary);
			}
#ifdef SEED_SKIP_IDLE      // (experiment: a kind of request no lane of the wave makes is not issued at all -- the other form reads entry 0 of the table)
#define ANY_IN(M) __any(mode == (M))
#else
#define ANY_IN(M) true
#endif
			FmBlock bk = {{0, 0, 0, 0}, {0, 0, 0, 0}, 0, 0, 0, 0}, bl = bk;
			if (ANY_IN(M_FM)) { bk = fm_load(di, kk >> 6); bl = fm_load(di, ll >> 6); }
			struct __attribute__((packed, aligned(4))) W5 { u32 a, b, c, d, e; };
			W5 w5 = {0, 0, 0, 0, 0};
			if (ANY_IN(M_TEXT)) w5 = *(const W5 *)(di.ref2 + (mode == M_TEXT ? (tp >> 4) : ((mode == M_MTEXT && (malive & 1u)) ? (mp[0] >> 4) : 0)));      // 20 bytes of packed text: a 64-base window
			const u32 r0 = w5.a, r1 = w5.b, r2 = w5.c, r3 = w5.d, r4 = w5.e;
			// one k-mer table entry: 16 bytes (one load) when the text is below 2^32, else 32
			ulonglong2 e0 = {0, 0}, e1 = {0, 0};
			if (!ANY_IN(M_KMER)) {}
			else if (E16) {
				const uint4 e = ((const uint4 *)(di.kmer ? di.kmer : (const u64 *)di.bwt))[mode == M_KMER ? kid : 0];
				e0.x = e.x; e0.y = e.y; e1.x = e.z; e1.y = e.w;
			} else {
				const ulonglong2 *pe = (const ulonglong2 *)((di.kmer ? di.kmer : (const u64 *)di.bwt) + (mode == M_KMER ? ((size_t)kid << 2) : 0));
				e0 = pe[0]; e1 = pe[1];
			}
			// presence bits of s and of the three starts behind it: a search that dies below MinSeedLength moves on by ONE base,
			// so walks cross the 14 bases in front of a mismatch start by start -- four of those per memory round trip, and since
			// round 3 all four from ONE 32-byte line (pres4_*)
			uint4 pl0 = {~0u, ~0u, ~0u, ~0u}, pl1 = {~0u, ~0u, ~0u, ~0u};
			if (di.pres && ANY_IN(M_KMER)) { const uint4 *pp = (const uint4 *)di.pres + 2 * (size_t)(mode == M_KMER ? pid : 0); pl0 = pp[0]; pl1 = pp[1]; }
			// (bit 64 i + e_i of the line: dword 2 i + (e_i >> 5).  e_i = pres4_bit(.., i) & 63 = the three bases outside the core = bits 2 i .. 2 i + 5 of
			//  the 12-bit number pext = q[s .. s+3) | q[s+K .. s+K+3) << 6: one number per open search instead of four packed ones)
#define PRES4_TEST(I) ((((((pext >> (2 * (I))) & 32u) ? ((I) == 0 ? pl0.y : (I) == 1 ? pl0.w : (I) == 2 ? pl1.y : pl1.w) : ((I) == 0 ? pl0.x : (I) == 1 ? pl0.z : (I) == 2 ? pl1.x : pl1.z)) >> ((pext >> (2 * (I))) & 31u)) & 1u) != 0)
			u64 sav = 0;
			if (ANY_IN(M_LOC)) sav = fm_locate(di, (mode == M_LOC || mode == M_MLOC) ? ik.x0 : 1);
#undef ANY_IN
			u64 sav1 = 0, sav2 = 0, sav3 = 0; W5 w5b = {0, 0, 0, 0, 0}, w5c = w5b, w5d = w5b;
			if (!COUNT && SEED_MULTI > 1) {
				if (__any(mode == M_MLOC)) {      // (rare enough per wave-iteration that the loads are only issued when somebody needs them)
					const int nr = mode == M_MLOC ? (int)ik.x2 : 0;
					sav1 = fm_locate(di, nr > 1 ? ik.x0 + 1 : 1); sav2 = fm_locate(di, nr > 2 ? ik.x0 + 2 : 1); sav3 = fm_locate(di, nr > 3 ? ik.x0 + 3 : 1);
				}
				if (__any(mode == M_MTEXT)) {
					const bool mt = mode == M_MTEXT;
					w5b = *(const W5 *)(di.ref2 + ((mt && (malive & 2u)) ? (mp[1] >> 4) : 0));
					w5c = *(const W5 *)(di.ref2 + ((mt && (malive & 4u)) ? (mp[2] >> 4) : 0));
					w5d = *(const W5 *)(di.ref2 + ((mt && (malive & 8u)) ? (mp[3] >> 4) : 0));
				}
			}
			// ---- consume phase: straight-line, one predicated block per mode ----
			bool ended = false;
			if (mode == M_KMER) {
				if (!PRES4_TEST(0)) {                  // the first MinSeedLength bases do not occur: no seed here, next start s+1
					memo_one(memo_l, s); s += 1; mode = M_ADV;
					// ... and the same for the starts behind it, as long as nothing else is known about them (the advance step
					// below owns every other rule: sub-range end, memoised hop, ambiguous bases, too close to the chunk end)
					const int L = prm.MinSeedLength < 32 ? prm.MinSeedLength : 32;
#pragma unroll
					for (int k2 = 0; k2 < PLOOK; k2++) {
						if (s >= bend || memo_nib(memo_l, s)) break;
						const u32 nb = q_nbits32(qn_l, s);
						if (s + prm.MinSeedLength > lclen || (nb & (L == 32 ? ~0u : (1u << L) - 1)) != 0) break;
						if (k2 == 0 ? PRES4_TEST(1) : k2 == 1 ? PRES4_TEST(2) : PRES4_TEST(3)) break;          // occurs: needs its table entry (next iteration)
						memo_one(memo_l, s); s += 1;
					}
				} else {
					const bool hit = e1.x != 0;         // absent k-mer: the match is shorter than k, walk it base by base
					if (hit) { ik.x0 = e0.x; ik.x1 = e0.y; ik.x2 = e1.x; pos = s + di.kmer_k; }
					else ik = fm_init(di, q_code(qp_l, s));      // (pos = s + 1 since the search was opened)
					mode = M_FM;
					if (hit && ik.x2 == 1) { tp = (i64)(e1.y - 1) + di.kmer_k; mode = M_TEXT; }      // unique: straight to the text comparison
					else if (!COUNT && SEED_MULTI > 1 && hit && ik.x2 <= SEED_MULTI) mode = M_MLOC;      // a few rows: their text positions, then the text
				}
			} else if (mode == M_LOC) {
				tp = (i64)sav + (pos - s); mode = M_TEXT;
			} else if (!COUNT && SEED_MULTI > 1 && mode == M_MLOC) {
				mp[0] = (i64)sav + (pos - s); mp[1] = (i64)sav1 + (pos - s); mp[2] = (i64)sav2 + (pos - s); mp[3] = (i64)sav3 + (pos - s);
				malive = (1u << (int)ik.x2) - 1u; mode = M_MTEXT;
			} else if (!COUNT && SEED_MULTI > 1 && mode == M_MTEXT) {
				int g0 = -1, g1 = -1, g2 = -1, g3 = -1;
#define MT_ROW(G, W, P) { int got = text_match32(W.a, W.b, W.c, P, (i64)di.seq_len, qp_l, qn_l, pos, lclen); if (got == 32) got += text_match32(W.c, W.d, W.e, P + 32, (i64)di.seq_len, qp_l, qn_l, pos + 32, lclen); G = got; }
				if (malive & 1u) MT_ROW(g0, w5, mp[0])
				if (malive & 2u) MT_ROW(g1, w5b, mp[1])
				if (malive & 4u) MT_ROW(g2, w5c, mp[2])
				if (malive & 8u) MT_ROW(g3, w5d, mp[3])
#undef MT_ROW
				int m = g0 > g1 ? g0 : g1; m = m > g2 ? m : g2; m = m > g3 ? m : g3;
				const u32 na = (g0 == m ? 1u : 0u) | (g1 == m ? 2u : 0u) | (g2 == m ? 4u : 0u) | (g3 == m ? 8u : 0u);      // (rows not alive hold -1 < m)
				pos += m; mp[0] += m; mp[1] += m; mp[2] += m; mp[3] += m;
				if (m < 64 || __popc(na) == 1) {
					// the rows that reach the longest match are the final interval (contiguous: the rows are sorted by suffix)
					const int first = __ffs((int)na) - 1;
					ik.x0 += (u64)first; ik.x2 = (u64)__popc(na);
					if (m == 64) { tp = first == 0 ? mp[0] : first == 1 ? mp[1] : first == 2 ? mp[2] : mp[3]; mode = M_TEXT; }      // one row left and still matching: the unique-interval path
					else ended = true;
				} else malive = na;
				if (mode == M_MTEXT && !ended && malive != na) malive = na;
			} else if (mode == M_TEXT) {
				int got = text_match32(r0, r1, r2, tp, (i64)di.seq_len, qp_l, qn_l, pos, lclen);
				if (got == 32) got += text_match32(r2, r3, r4, tp + 32, (i64)di.seq_len, qp_l, qn_l, pos + 32, lclen);
				pos += got; tp += got;
				ended = got < 64;
			} else if (mode == M_FM) {
				const bool can = pos < lclen && !q_isn(qn_l, pos < lclen ? pos : 0);
				const bool ok = can && fm_extend_loaded(di, ik, q_code(qp_l, pos < lclen ? pos : 0), bk, bl, kk, ll, kn, ln, blk);
				ended = !ok;
				if (ok) { pos++; if (!COUNT && ik.x2 == 1) mode = M_LOC; else if (!COUNT && SEED_MULTI > 1 && ik.x2 <= SEED_MULTI) mode = M_MLOC; }
			}
			if (ended) {
				const int len = pos - s;
				int d = 1;
				if (len >= prm.MinSeedLength && ik.x2 <= GSA_MAX_SEED_FREQ) {
					const int ch = NCH == 1 ? 0 : (item >= NSUB ? NCH - 1 : 0);
					const u32 slot = atomicAdd(&s_ncand[ch], 1u);                 // LDS counter: no global round trip in the loop
					const size_t cb = CH_SEL(cbase, ch);
					if (slot < cand_cap) { cand_s[cb + slot] = (i32)(CH_SEL(c0, ch) + s); cand_len[cb + slot] = len; cand_x0[cb + slot] = ik.x0; cand_freq[cb + slot] = (i32)ik.x2; }
					else cnt[CNT_OVERFLOW] = 1;
					d = prm.bSensitive ? 5 : len + 1;
				}
				memo_set(memo_l, lhop_l, s, d, &s_abort); if (COUNT) mblk[s] = (uint16_t)blk;
				all_blocks += blk;
				s += d; mode = M_ADV;
			}
			// ---- advance: ONE step per iteration (no inner loop): take an item / hop over a memoised or
			// ambiguous position / open the next search ----
			// (a few steps per iteration: hops over memoised / ambiguous positions cost no memory access)
			for (int step = 0; step < ADV_STEPS && mode == M_ADV; step++) {
				if (need_item) {
					if (rounds == 0) {
						// (the queue hands the two chunks' sub-ranges out alternately, so that both chunks' long walks start early)
						const u32 it_ = atomicAdd(&s_queue, 1u);
						if (NCH == 1) item = it_ < (u32)nitems[0] ? (int)it_ : -1;
						else {
							const u32 both = 2u * (u32)(nitems[0] < nitems[NCH - 1] ? nitems[0] : nitems[NCH - 1]);
							if (it_ < both) item = (int)(it_ >> 1) + ((it_ & 1u) ? NSUB : 0);
							else if (it_ < (u32)nitems_all) { const int r = (int)(it_ - both) + (int)(both >> 1); item = nitems[0] > nitems[NCH - 1] ? r : NSUB + r; }
							else item = -1;
						}
					}
					else if (dirty) { dirty = 0; item = fb_item; }
					else item = -1;
					need_item = false;
					if (item < 0) { mode = M_DONE; break; }
					const int ch = NCH == 1 ? 0 : (item >= NSUB ? NCH - 1 : 0), li = item - ch * NSUB;
					lclen = CH_SEL(clen, ch); qp_l = qp[ch]; qn_l = qn[ch]; memo_l = memo[ch]; lhop_l = lhop[ch];
					const int S_l = CH_SEL(S, ch);
					s = entry_of[item]; bend = (li + 1) * S_l < lclen ? (li + 1) * S_l : lclen;
				}
				if (s >= bend) { exit_of[item] = (uint16_t)s; need_item = true; continue; }
				const int m_ = memo_get(memo_l, lhop_l, s);
				if (m_) { s += m_; continue; }
				const u32 nb = q_nbits32(qn_l, s);
				const int L = prm.MinSeedLength < 32 ? prm.MinSeedLength : 32;
				if (nb & 1u) { memo_one(memo_l, s); if (COUNT) mblk[s] = 0; s += 1; }
				else if (!COUNT && (s + prm.MinSeedLength > lclen || (nb & (L == 32 ? ~0u : (1u << L) - 1)) != 0)) { memo_one(memo_l, s); s += 1; }      // cannot reach MinSeedLength
				else {
					pos = s + 1; blk = 0; mode = M_FM;
					// (the interval of the first base -- fm_init: three 5-way selects of 64-bit numbers -- only where the walk really starts
					//  at the first base: next to the chunk end / an N, or behind a k-mer entry that says "absent")
#ifdef SEED_EAGER_INIT      // (A/B switch: the interval of the first base computed for every search that is opened, as until late round 3)
					ik = fm_init(di, q_code(qp_l, s));
#endif
					if (COUNT || !(di.kmer_k > 1 && s + di.kmer_k <= lclen && (nb & ((1u << di.kmer_k) - 1)) == 0)) ik = fm_init(di, q_code(qp_l, s));
					else {
						const u64 qb = q_bits64(qp_l, s);
						kid = (u32)(qb & ((1ull << (2 * di.kmer_k)) - 1)); mode = M_KMER;
						// the line of the presence table that answers for s .. s+3, and the four positions inside it
						pid = di.pres_k ? pres4_line(qb, di.pres_k) : 0;
						pext = di.pres_k ? (((u32)qb & 63u) | (((u32)(qb >> (2 * di.pres_k)) & 63u) << 6)) : 0;
					}
				}
			}
		}
#undef PRES4_TEST
		rounds++;
		SEED_SYNC();
		if (s_abort) break;
		if (rounds == 1) t_round0 = wall_clock64() - t_begin;
		const unsigned long long t_r0 = wall_clock64();
		// True entries.  A walk that enters sub-range `it` on a memoised position leaves it at exit_of[it]
		// wherever it entered (all walks inside a sub-range merge), so at sub-range level the true chain is
		// the orbit of 0 under it -> exit_of[it] / S: found by pointer jumping (log2 NSUB parallel rounds in
		// LDS), the true entry of an on-chain sub-range being its predecessor's exit.  In round 1 only
		// `it`'s own walk wrote memo[] inside `it`, so an entry e with memo[e] != 0 lies on that walk.  An
		// entry the speculation never visited (a random >= MinSeedLength match made the speculative walk
		// jump over it) needs a real walk: such sub-ranges are collected, ASSUMED to keep their exit, walked
		// in parallel (one lane each) in another pass of the loop above, and the chain is resolved again.
		// (v = ch * NSUB + sub-range: the chains of the chunks of a pair are resolved together, one root each)
#define V_LIVE(V) ((V) < NSUB ? (V) < nitems[0] : (NCH > 1 && (V) - NSUB < nitems[NCH - 1]))
		for (int v = j; v < NV; v += SEED_WG) if (V_LIVE(v)) {
			const int ch = v >= NSUB ? NCH - 1 : 0;
			const int X = exit_of[v]; jmp[0][v] = (uint16_t)(X >= CH_SEL(clen, ch) ? 0xffff : ch * NSUB + X / CH_SEL(S, ch));
		}
		if (j < (NV + 31) / 32) onchain[j] = 0;
		if (j == 0) s_npend = 0;
		SEED_SYNC();
		if (j == 0) { onchain[0] = 1u; if (NCH > 1 && nitems[NCH - 1] > 0) atomicOr(&onchain[NSUB >> 5], 1u << (NSUB & 31)); }
		SEED_SYNC();
		int cur = 0;
		const int nmax = NCH == 1 ? nitems[0] : (nitems[0] > nitems[NCH - 1] ? nitems[0] : nitems[NCH - 1]);
		for (int span = 1; span < nmax; span <<= 1) {
			for (int v = j; v < NV; v += SEED_WG) if (V_LIVE(v)) {
				const int t = jmp[cur][v];
				if (t != 0xffff) {
					if ((onchain[v >> 5] >> (v & 31)) & 1u) atomicOr(&onchain[t >> 5], 1u << (t & 31));
					jmp[cur ^ 1][v] = jmp[cur][t];
				} else jmp[cur ^ 1][v] = 0xffff;
			}
			SEED_SYNC();
			cur ^= 1;
		}
		for (int v = j; v < NV; v += SEED_WG) if (V_LIVE(v)) { const int ch = v >= NSUB ? NCH - 1 : 0; entry_of[v] = (uint16_t)((v - ch * NSUB) ? CH_SEL(clen, ch) : 0); }      // off-chain: nothing to mark
		SEED_SYNC();
		for (int v = j; v < NV; v += SEED_WG) if (V_LIVE(v))
			if ((onchain[v >> 5] >> (v & 31)) & 1u) { const int ch = v >= NSUB ? NCH - 1 : 0; const int X = exit_of[v]; if (X < CH_SEL(clen, ch)) entry_of[ch * NSUB + X / CH_SEL(S, ch)] = (uint16_t)X; }
		SEED_SYNC();
		for (int v = j; v < NV; v += SEED_WG) if (V_LIVE(v)) {
			if (!((onchain[v >> 5] >> (v & 31)) & 1u)) continue;
			const int ch = v >= NSUB ? NCH - 1 : 0;
			const int e = entry_of[v];
			const bool known = ((rewalked[v >> 5] >> (v & 31)) & 1u) ? e == walked_from[v] : memo_nib(memo[ch], e) != 0;
			if (!known) {
				const u32 idx = atomicAdd(&s_npend, 1u);
				if (idx < SEED_WG) { pend_it[idx] = (uint16_t)v; walked_from[v] = (uint16_t)e; atomicOr(&rewalked[v >> 5], 1u << (v & 31)); }
			}
		}
		SEED_SYNC();
		if (j == 0) { if (s_npend > SEED_WG) s_npend = SEED_WG; changed = s_npend > 0 ? 1 : 0; }
		SEED_SYNC();
		t_resolve += wall_clock64() - t_r0;
		const int again = changed;
		if (again && j < (int)s_npend) { fb_item = pend_it[j]; dirty = 1; }
		SEED_SYNC();
		if (!again) break;
	}
	const bool heavy = s_abort != 0;
	if (heavy) {
		if (j == 0) for (int ch = 0; ch < nch; ch++) {
			const u32 hslot = (u32)atomicAdd((unsigned long long *)&cnt[CNT_HEAVY], 1ull);
			heavy_list[hslot] = (u32)(chunk0 + ch);
			s_ncand[ch] = 0; cand_cnt[chunk0 + ch] = 0; lb_pub(&chunk_hits[chunk0 + ch], 0); if (chunk0 + ch == 0) lb_pub(&chunk_hits[n_chunks], 0);
		}
		SEED_SYNC();
	}
	// mark the true path and count the Occ blocks the reference's walk reads
	u32 alg_blocks = 0;
	if (!heavy) for (int v = j; v < NV; v += SEED_WG) if (V_LIVE(v)) {
		const int ch = v >= NSUB ? NCH - 1 : 0, li = v - ch * NSUB;
		const int bend = (li + 1) * CH_SEL(S, ch) < CH_SEL(clen, ch) ? (li + 1) * CH_SEL(S, ch) : CH_SEL(clen, ch);
		for (int s = entry_of[v]; s < bend;) { atomicOr(&bits[ch][s >> 5], 1u << (s & 31)); if (COUNT) alg_blocks += mblk[s]; s += memo_get(memo[ch], lhop[ch], s); }
	}
#undef V_LIVE
	for (int o = 32; o; o >>= 1) { alg_blocks += __shfl_down(alg_blocks, o); all_blocks += __shfl_down(all_blocks, o); }
	if ((j & 63) == 0) {
		if (alg_blocks) atomicAdd((unsigned long long *)&cnt[CNT_OCCBLK], (unsigned long long)alg_blocks);
		if (all_blocks) atomicAdd((unsigned long long *)&cnt[CNT_OCCBLK_ALL], (unsigned long long)all_blocks);
#ifndef SEED_STATS
		atomicMax((unsigned long long *)&cnt[13], (unsigned long long)iters);
#endif
	}
#ifdef SEED_STATS
	if ((j & 63) == 0) { atomicAdd((unsigned long long *)&cnt[11], (unsigned long long)st_it); atomicAdd((unsigned long long *)&cnt[13], (unsigned long long)st_fm_any); atomicAdd((unsigned long long *)&cnt[14], (unsigned long long)st_fm_only);
	                     atomicAdd((unsigned long long *)&cnt[15], (unsigned long long)st_act); atomicAdd((unsigned long long *)&cnt[7], (unsigned long long)st_fm); }
	if (false)
#endif
	if (j == 0) { atomicMax((unsigned long long *)&cnt[11], (unsigned long long)rounds); atomicMax((unsigned long long *)&cnt[14], t_round0); atomicMax((unsigned long long *)&cnt[15], t_resolve); atomicMax((unsigned long long *)&cnt[7], wall_clock64() - t_begin); }
	SEED_SYNC();
	// per chunk: the on-path bits, and how many located hits it will contribute (so that the select kernel needs no global atomic)
	if (!heavy) {
#pragma unroll
		for (int ch = 0; ch < NCH; ch++) if (ch < nch) {
			for (int p = j; p < PATH_WORDS; p += SEED_WG) onpath[(size_t)(chunk0 + ch) * PATH_WORDS + p] = bits[ch][p];
			const u32 nc = s_ncand[ch] < cand_cap ? s_ncand[ch] : cand_cap;
			u32 h = 0;
			for (u32 i = j; i < nc; i += SEED_WG) { const i32 p = cand_s[cbase[ch] + i] - (i32)c0[ch]; if ((bits[ch][p >> 5] >> (p & 31)) & 1u) h += (u32)cand_freq[cbase[ch] + i]; }
			for (int o = 32; o; o >>= 1) h += __shfl_down(h, o);
			if ((j & 63) == 0 && h) atomicAdd(&s_hits[ch], h);
		}
		SEED_SYNC();
		if (j == 0) for (int ch = 0; ch < nch; ch++) {
			const u32 nc = s_ncand[ch] < cand_cap ? s_ncand[ch] : cand_cap;
			cand_cnt[chunk0 + ch] = nc; lb_pub(&chunk_hits[chunk0 + ch], (i32)s_hits[ch]); if (chunk0 + ch == 0) lb_pub(&chunk_hits[n_chunks], 0); atomicMax((unsigned long long *)&cnt[CNT_CAND], (unsigned long long)s_ncand[ch]);
			if (s_hits[ch]) atomicAdd((unsigned long long *)&cnt[CNT_HITS], (unsigned long long)s_hits[ch]);      // the contig's total: all the host needs to go on
		}
	}
#undef CH_SEL
	// the workgroup that is through last puts the counters into pinned memory (the host waits for this kernel, nothing
	// else) and leaves them at zero for the next contig
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (every wave: its counter atomics are done before the workgroup counts itself)
	SEED_SYNC();
	if (j == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (counters are device atomics; an agent-scope fence is an L2 write-back per workgroup here)
		s_last = atomicAdd((unsigned long long *)&cnt[CNT_DONE], (unsigned long long)nch) == (unsigned long long)n_chunks - (unsigned long long)nch ? 1 : 0;
	}
	SEED_SYNC();
	if (s_last && j < 16) {
		hcnt[j] = j == CNT_DONE ? 0 : __hip_atomic_load(&cnt[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		cnt[j] = 0;
	}
	if (s_last) { if (WPW == 1) wg_exscan_hits<SEED_WG>(chunk_hits, chunk_base, (int)n_chunks + 1); else wave_exscan_hits(chunk_hits, chunk_base, (int)n_chunks + 1); }
}
#undef s_ncand
#undef s_queue
#undef s_hits
#undef changed
#undef s_abort
#undef qp
#undef qn
#undef memo
#undef lhop
#undef mblk
#undef bits
#undef entry_of
#undef exit_of
#undef pend_it
#undef jmp
#undef walked_from
#undef rewalked
#undef onchain
#undef s_npend
#undef s_last
#undef SEED_SYNC


// The kernel: workgroups DRAW their chunk pairs from a ticket counter (cnt[SEED_TICKET], never reset: the host passes the value it has
// at launch, and a launch of g workgroups over n units leaves it n + g higher -- every workgroup's last draw is the one that fails).
// With g = n every workgroup takes one unit, as a plain grid would; with fewer the launch is PERSISTENT and holds at most g
// workgroups' worth of LDS and wave slots whatever the contig's size (what it leaves free the kernels of other contexts can take).
#define SEED_TICKET 16
#ifndef SEED_PERSIST
#define SEED_PERSIST 12        // (A/B builds with SEED_WPW = 1 only: one-wave workgroups per CU of round 5's launch shape -- what the LDS admits at 12.8 KB per chunk)
#endif
#ifndef SEED_NCH
#define SEED_NCH 1              // chunks per wave of the production kernel (2: measured slower, see seed_chunk; the accounting build: always 1)
#endif
// Round 6 -- THE KERNEL'S FOOTPRINT ON A CU.  Until round 5 twelve one-wave workgroups sat on every CU the kernel held: 154 KB of its 160 KB of LDS and 89 % of its
// vector registers (3 waves per SIMD x 152), for a kernel that issues VALU in 20 % of its cycles and waits for memory in 54 %.  Nothing else fitted beside it -- not a
// striped-DP workgroup (13 - 55 KB of LDS, 48 VGPRs), not a fused pass (four waves of ~100 VGPRs) -- so with four contexts in flight the CUs were handed back and forth
// between kernels that each use a fraction of them, and the step was the SUM of the stages (52 + 46 ms of the 98).  Now:
//  * a workgroup is SEED_WPW = 8 INDEPENDENT waves, each with its own SeedLds and its own tickets, and asks for more than half a CU's LDS (102 KB), so exactly ONE
//    workgroup fits a CU and a launch of n_cus workgroups lands on EVERY CU (the dispatcher fills a CU before it moves on: a short grid of one-wave workgroups meant
//    fewer CUs, not thinner ones);
//  * the register budget is 96 VGPRs per wave (SEED_MIN_WAVES = 5; the LDS is a launch parameter because with static LDS the compiler knows that the LDS holds the kernel
//    at three waves per SIMD and spends the registers that leaves, whatever the launch bound says: 149).
// A CU that runs the seed kernel keeps 58 KB of LDS and ~320 VGPRs per SIMD lane free: a top-class DP workgroup, or four low-class ones, or the fused passes of the other
// contexts run BESIDE it and issue while its waves wait.  Measured (profiles/r06_seed_footprint.txt, four contexts, human index): the seed stage ALONE 52 -> 60 ms (eight
// chunks per CU instead of twelve, spills), the STEP 98.7 -> 93.1 ms; 250 Mb contigs 35.7 -> 38.2 Gbp/s.  Either half alone does nothing: 8 waves at 155 VGPRs 98.7 ms,
// 12 one-wave workgroups at 96 VGPRs 97.9 ms.
#ifndef SEED_WPW
#define SEED_WPW 8              // waves (= chunks in flight) per workgroup = per CU (7 / 9 / 10: 96.9 / 95.2 / 94.7 ms; 1 = round 5's twelve one-wave workgroups per CU, an A/B build)
#endif
#ifndef SEED_WGS_PER_CU
#define SEED_WGS_PER_CU 1       // fat workgroups per CU (the LDS a workgroup asks for is more than 160 KB / (SEED_WGS_PER_CU + 1): one more never fits)
#endif
#define SEED_WG_LDS_MIN ((160 * 1024) / (SEED_WGS_PER_CU + 1) + 2048)
#define SEED_LB_WAVES(COUNT_) ((COUNT_) ? 1 : SEED_MIN_WAVES)
template <bool COUNT, bool E16>
__global__ void __launch_bounds__(SEED_WG * (COUNT ? 1 : SEED_WPW), SEED_LB_WAVES(COUNT)) k_seed_wg(DevIndex di, const uint8_t *__restrict__ q, i32 qlen, Params prm, u64 *cnt,
                                                      i32 *cand_s, i32 *cand_len, u64 *cand_x0, i32 *cand_freq, u32 cand_cap, u32 *cand_cnt, u32 *onpath, i32 *chunk_hits, u64 *hcnt,
                                                      u32 budget, u32 *heavy_list, i32 *chunk_base, u64 tk_base, u32 n_chunks)
{
	constexpr int NCH = COUNT ? 1 : SEED_NCH;
	constexpr int WPW = COUNT ? 1 : SEED_WPW;
	typedef SeedLds<COUNT, NCH> Lds;
	extern __shared__ __attribute__((aligned(16))) unsigned char seed_dyn_lds[];      // (a launch parameter: see the kernel's header)
	Lds *lds = (Lds *)seed_dyn_lds;
	Lds &L = lds[WPW == 1 ? 0 : (threadIdx.x >> 6)];
	for (;;) {
		u32 unit = 0;
		if ((threadIdx.x & 63) == 0) unit = (u32)(atomicAdd((unsigned long long *)&cnt[SEED_TICKET], 1ull) - tk_base);
		unit = (u32)__builtin_amdgcn_readfirstlane((int)unit);
		if ((u64)unit * NCH >= n_chunks) return;
		seed_chunk<COUNT, E16, NCH, WPW>(di, q, qlen, prm, cnt, cand_s, cand_len, cand_x0, cand_freq, cand_cap, cand_cnt, onpath, chunk_hits, hcnt, budget, heavy_list, chunk_base, (int)(unit * NCH), n_chunks, L);
	}
}


// ---------------------------------------------------------------------------
// Dense mode: BWT_Search from EVERY start position of a chunk, one lane per start, then the reference's chain
// (IdentifyLocalMEM, GSAlign.cpp:61-94) by pointer jumping over next(s).  next(s) is a pure function of s, so this is
// exact; it does up to 10 000 searches per chunk where the speculative kernel above does a few hundred, so it is used
// where nearly every start is on the chain anyway or the chain cannot be guessed:
//   * -sen (stride 5 after a seed: walks that start on different residues mod 5 only merge at the next mismatch, so the
//     true entry of every sub-range depends on its predecessor -- 127 resolver rounds per chunk were measured);
//   * chunks the speculative kernel gave up on (tandem arrays with more than MaxSeedFreq copies: every start is searched
//     for ~100 bases, rejected, and followed by start+1 -- 46 ms on one workgroup for a 6-kb array).
// Same search ladder as above: presence bitmap -> k-mer table -> (Occ steps until one row is left) -> dense SA ->
// 64-base text windows.  Consecutive lanes hold consecutive starts, so a wavefront's searches end at the same mismatch.
// ---------------------------------------------------------------------------
#define DENSE_TPB 256
// SPAN = starts per workgroup: 512 (two per lane) when every chunk is dense (-sen), 256 for the few chunks the speculative
// kernel gave up on (their searches are ~100 dependent Occ steps each: one per lane halves the latency of that detour)
#define DENSE_WGS(SPAN) ((GSA_CHUNK + (SPAN) - 1) / (SPAN))
template <bool E16, int DENSE_SPAN>
__global__ void __launch_bounds__(DENSE_TPB) k_dense_search(DevIndex di, const uint8_t *__restrict__ q, i32 qlen, Params prm, const u32 *__restrict__ chunk_list,
                                                              u32 *dn_lf, u64 *dn_x0, u64 *cnt)
{
	__shared__ u32 qp[QP_WORDS], qn[QN_WORDS];
	const u32 slot = blockIdx.x / DENSE_WGS(DENSE_SPAN), part = blockIdx.x % DENSE_WGS(DENSE_SPAN);
	const u32 chunk = chunk_list ? chunk_list[slot] : slot;
	const int j = threadIdx.x;
	const i64 c0 = (i64)chunk * GSA_CHUNK;
	const int clen = (int)((i64)qlen - c0 < GSA_CHUNK ? (i64)qlen - c0 : GSA_CHUNK);
	const int span0 = (int)part * DENSE_SPAN, span1 = span0 + DENSE_SPAN < clen ? span0 + DENSE_SPAN : clen;
	if (span0 >= clen) return;
	// stage the chunk from the first start of this workgroup to its end (a match may run that far)
	for (int g = (span0 >> 5) + j; g < QN_WORDS; g += DENSE_TPB) {
		u32 w0 = 0, w1 = 0, wn = 0;
		const int p0 = g << 5;
		if (p0 < clen) {
			stage32(q + c0 + p0, p0, clen, w0, w1, wn);
		}
		if (2 * g < QP_WORDS) qp[2 * g] = w0;
		if (2 * g + 1 < QP_WORDS) qp[2 * g + 1] = w1;
		qn[g] = wn;
	}
	__syncthreads();
	u32 *lf = dn_lf + (size_t)slot * GSA_CHUNK; u64 *x0o = dn_x0 + (size_t)slot * GSA_CHUNK;
	int nextp = span0 + j;                                  // this lane's starts: nextp, nextp + DENSE_TPB
	int s = 0, pos = 0, mode = M_ADV; u32 kid = 0, pid = 0, pext = 0, blk = 0, all_blocks = 0;
	FmIntv ik = {0, 0, 0}; i64 tp = 0;
	const int L = prm.MinSeedLength < 32 ? prm.MinSeedLength : 32;
	while (!__all(mode == M_DONE)) {
		// ---- request phase: one pending request per lane, all lanes issue together ----
		u64 kk = 0, ll = 0; bool kn = true, ln = true;
		if (mode == M_FM) {
			const u64 k = ik.x1 - 1, l = ik.x1 - 1 + ik.x2;
			kn = (k == (u64)-1); ln = (l == (u64)-1);
			kk = kn ? 0 : k - (k >= di.primary); ll = ln ? 0 : l - (l >= di.primary);
		}
		const FmBlock bk = fm_load(di, kk >> 6), bl = fm_load(di, ll >> 6);
		struct __attribute__((packed, aligned(4))) W5 { u32 a, b, c, d, e; };
		const W5 w5 = *(const W5 *)(di.ref2 + (mode == M_TEXT ? (tp >> 4) : 0));
		ulonglong2 e0 = {0, 0}, e1 = {0, 0};
		if (E16) {
			const uint4 e = ((const uint4 *)(di.kmer ? di.kmer : (const u64 *)di.bwt))[mode == M_KMER ? kid : 0];
			e0.x = e.x; e0.y = e.y; e1.x = e.z; e1.y = e.w;
		} else {
			const ulonglong2 *pe = (const ulonglong2 *)((di.kmer ? di.kmer : (const u64 *)di.bwt) + (mode == M_KMER ? ((size_t)kid << 2) : 0));
			e0 = pe[0]; e1 = pe[1];
		}
		// the short table: for a start whose kmer_k-mer does not occur (its match ends before kmer_k bases) -- the common case
		// under -sen, where a match of 10..14 bases is a seed -- the interval after kmer_lo_k bases, instead of walking them
		ulonglong2 l0 = {0, 0}, l1 = {0, 0};
		if (di.kmer_lo) {
			if (E16) { const uint4 e = ((const uint4 *)di.kmer_lo)[mode == M_KLO ? kid : 0]; l0.x = e.x; l0.y = e.y; l1.x = e.z; l1.y = e.w; }
			else { const ulonglong2 *pe = (const ulonglong2 *)(di.kmer_lo + (mode == M_KLO ? ((size_t)kid << 2) : 0)); l0 = pe[0]; l1 = pe[1]; }
		}
		// (one start per lane here: role 0 of the group that starts at s -- dword pid of the grouped presence table, bit pext)
		const u32 pw = di.pres ? di.pres[mode == M_KMER ? pid : 0] : ~0u;
		const u64 sav = fm_locate(di, mode == M_LOC ? ik.x0 : 1);
		// ---- consume phase ----
		bool ended = false;
		if (mode == M_KMER) {
			if (!((pw >> pext) & 1u)) { ended = true; pos = s; ik.x2 = 0; }      // the first MinSeedLength bases do not occur: no seed here
			else {
				const bool hit = e1.x != 0;         // absent k-mer: the match is shorter than k
				if (hit) { ik.x0 = e0.x; ik.x1 = e0.y; ik.x2 = e1.x; pos = s + di.kmer_k; }
				mode = M_FM;                        // (no short table: walk it base by base from the first base)
				if (hit && ik.x2 == 1) { tp = (i64)(e1.y - 1) + di.kmer_k; mode = M_TEXT; }
				if (!hit && di.kmer_lo) { kid = kid & ((1u << (2 * di.kmer_lo_k)) - 1); mode = M_KLO; }      // (the start passed the N / length tests for kmer_k >= kmer_lo_k bases)
			}
		} else if (mode == M_KLO) {
			const bool hit = l1.x != 0;
			if (hit) { ik.x0 = l0.x; ik.x1 = l0.y; ik.x2 = l1.x; pos = s + di.kmer_lo_k; }
			mode = M_FM;
			if (hit && ik.x2 == 1) { tp = (i64)(l1.y - 1) + di.kmer_lo_k; mode = M_TEXT; }
		} else if (mode == M_LOC) {
			tp = (i64)sav + (pos - s); mode = M_TEXT;
		} else if (mode == M_TEXT) {
			int got = text_match32(w5.a, w5.b, w5.c, tp, (i64)di.seq_len, qp, qn, pos, clen);
			if (got == 32) got += text_match32(w5.c, w5.d, w5.e, tp + 32, (i64)di.seq_len, qp, qn, pos + 32, clen);
			pos += got; tp += got;
			ended = got < 64;
		} else if (mode == M_FM) {
			const bool can = pos < clen && !q_isn(qn, pos < clen ? pos : 0);
			const bool ok = can && fm_extend_loaded(di, ik, q_code(qp, pos < clen ? pos : 0), bk, bl, kk, ll, kn, ln, blk);
			ended = !ok;
			if (ok) { pos++; if (ik.x2 == 1) mode = M_LOC; }
		}
		if (ended) {
			const int len = pos - s;
			u32 rec = 0;
			if (len >= prm.MinSeedLength && ik.x2 <= GSA_MAX_SEED_FREQ) { rec = (u32)len | ((u32)ik.x2 << 16); x0o[s] = ik.x0; }
			lf[s] = rec;      // (the hop follows from the record: k_dense_resolve)
			all_blocks += blk;
			mode = M_ADV;
		}
		// ---- next start of this lane (starts that need no search are settled here, two per iteration) ----
		for (int step = 0; step < 2 && mode == M_ADV; step++) {
			if (nextp >= span1) { mode = M_DONE; break; }
			s = nextp; nextp += DENSE_TPB;
			const u32 nb = q_nbits32(qn, s);
			if ((nb & 1u) || s + prm.MinSeedLength > clen || (nb & (L == 32 ? ~0u : (1u << L) - 1)) != 0) { lf[s] = 0; continue; }      // ambiguous start, or MinSeedLength out of reach
			ik = fm_init(di, q_code(qp, s)); pos = s + 1; blk = 0; mode = M_FM;
			if (di.kmer_k > 1 && s + di.kmer_k <= clen && (nb & ((1u << di.kmer_k) - 1)) == 0) {
				const u64 qb = q_bits64(qp, s);
				kid = (u32)(qb & ((1ull << (2 * di.kmer_k)) - 1)); mode = M_KMER;
				{ const u32 b0 = di.pres_k ? pres4_bit(qb, di.pres_k, 0) : 0; pid = di.pres_k ? pres4_line(qb, di.pres_k) * 8 + (b0 >> 5) : 0; pext = b0 & 31u; }
			} else if (di.kmer_lo && s + di.kmer_lo_k <= clen && (nb & ((1u << di.kmer_lo_k) - 1)) == 0) {
				kid = (u32)(q_bits64(qp, s) & ((1ull << (2 * di.kmer_lo_k)) - 1)); mode = M_KLO;      // (too close to the chunk end or an N for the long table)
			}
		}
	}
	for (int o = 32; o; o >>= 1) all_blocks += __shfl_down(all_blocks, o);
	if ((j & 63) == 0 && all_blocks) atomicAdd((unsigned long long *)&cnt[CNT_OCCBLK_ALL], (unsigned long long)all_blocks);
}

// ---------------------------------------------------------------------------
// Sweep mode (round 3): next(s) for EVERY start of a chunk like k_dense_search, but not one search per start.  Two facts about
// L(s), the length of the longest match from s (what BWT_Search computes; its interval = all occurrences of q[s .. s+L(s))):
//   (1) L(s) <= L(s+1) + 1                    (q[s+1 .. s+L(s)) occurs)
//   (2) if q[s+1 .. e) is the longest match from s+1 and q[s .. e) occurs, then L(s) = e - s      (by (1))
// So a lane that owns SWEEP_SEG consecutive starts walks them RIGHT TO LEFT: one forward search from scratch for its last start
// (presence table -> k-mer table -> Occ steps -> dense SA -> 64-base text windows, as above), then for every start to the left
// it only asks "does the match extend by one base on the left?":
//   * the match is unique (one occurrence, at text position t): it extends iff text[t-1] equals the query base, and stays
//     unique -- 32 starts per comparison of packed words (M_BACK), no index access at all;
//   * the match has several occurrences (a repeat): ONE backward extension of the bi-interval (x0, x1, x2) -- the index is
//     symmetric (forward + reverse-complement text), so prepending base c is the forward step of the reference's BWT_Search
//     (bwt_search.cpp:152-165) with x0 and x1 swapped and the complementary base (M_BFM): one Occ step per start where the
//     reference and k_dense_search walk ~L(s) steps per start -- the `freq > MaxSeedFreq` reject-and-restart regime of
//     bwt_search.cpp:177-182 costs O(L) per copy of a repeat instead of O(L^2);
//   * it does not extend: L(s) < e - s, and the lane searches forward from s from scratch (exact by definition).
// Nothing is speculated and no lane depends on another: next(s) is a pure function of s.  Same outputs as k_dense_search
// (lf / x0 per start), so k_dense_resolve and everything downstream are unchanged; a unique match found by text
// comparison has no SA row at hand, so its x0 carries the text POSITION with bit 63 set (k_seed_select takes it as located).
// ---------------------------------------------------------------------------
#define SWEEP_POSFLAG (1ull << 63)
enum { M_BFM = 7, M_BACK = 8, M_LOC2 = 9 };
enum { HV_NONE = 0, HV_UNIQ = 1, HV_MULTI = 2 };
#ifdef SWEEP_NOSTORE      // (timing experiment: results are wrong)
#define SWEEP_ST(...)
#else
#define SWEEP_ST(...) __VA_ARGS__
#endif
#ifndef SWEEP_NCH
#define SWEEP_NCH 4            // chunks per workgroup (3.75 KB of LDS each)
#endif
#ifndef SWEEP_TPB
#define SWEEP_TPB 128
#endif
template <bool E16, int NCH, int TPB>
__global__ void __launch_bounds__(TPB) k_dense_sweep(DevIndex di, const uint8_t *__restrict__ q, i32 qlen, Params prm, const u32 *__restrict__ chunk_list, u32 n_slots,
                                                     u32 *dn_lf, u64 *dn_x0, u64 *cnt, int seg)
{
	// seg = starts per segment; long segments do the least work (one forward search per segment), short ones finish soonest.
	// Round 3 ran four waves per chunk with 40 starts per lane; round 4 first ONE wave per chunk with 160: inside a high-copy repeat a
	// segment costs ~150 Occ steps for its first (forward) search and one backward step per start, so 40 starts cost a lane 190
	// dependent steps and 160 cost it 310 -- a quarter of the forward searches.  But a lane in unique sequence is through with its
	// 160 starts after ~14 steps, and three lanes in four waited for the wave's repeat lanes while the kernel is bound by the
	// instructions its waves issue (tools/r4_adv_pmc.sh: 6.0 G VALU wave-instructions per 250 Mb, 318 iterations per wave).  So now
	// ONE WAVE takes NCH chunks and its lanes DRAW the segments (63 per chunk) from a counter in LDS: a lane that is through
	// takes the next one, whichever chunk it belongs to -- every lane carries its chunk (query words in LDS, record arrays) along.
	__shared__ u32 qp[NCH][QP_WORDS], qn[NCH][QN_WORDS];
	__shared__ u32 s_next;
	const int j = threadIdx.x;
	const u32 slot0 = blockIdx.x * (u32)NCH;
	const int nch = (int)(n_slots - slot0 < (u32)NCH ? n_slots - slot0 : (u32)NCH);
	const int spc = (GSA_CHUNK + seg - 1) / seg;                  // segments per chunk
	const u32 n_seg = (u32)(nch * spc);
	for (int ch = 0; ch < nch; ch++) {
		const u32 chunk = chunk_list ? chunk_list[slot0 + ch] : slot0 + ch;
		const i64 c0 = (i64)chunk * GSA_CHUNK;
		const int cl = (int)((i64)qlen - c0 < GSA_CHUNK ? (i64)qlen - c0 : GSA_CHUNK);
		for (int g = j; g < QN_WORDS; g += TPB) {
			u32 w0 = 0, w1 = 0, wn = 0;
			const int p0 = g << 5;
			if (p0 < cl) {
				stage32(q + c0 + p0, p0, cl, w0, w1, wn);
			} else wn = ~0u;
			if (2 * g < QP_WORDS) qp[ch][2 * g] = w0;
			if (2 * g + 1 < QP_WORDS) qp[ch][2 * g + 1] = w1;
			qn[ch][g] = wn;
		}
	}
	if (j == 0) s_next = 0;
	__syncthreads();
	// the lane's segment: chunk-relative starts [seg_a, cur] of the chunk whose words are qp_l / qn_l and whose records are lf / x0o
	const u32 *qp_l = qp[0], *qn_l = qn[0]; u32 *lf = dn_lf; u64 *x0o = dn_x0; int clen = 0;
	int seg_a = 0;
	int cur = -1;                                              // next start to settle; the lane draws a segment when cur < seg_a
	int s = 0, pos = 0, mode = M_ADV, have = HV_NONE, e_end = 0, prole = 0; u32 kid = 0, pid = 0, blk = 0, all_blocks = 0;
	u64 pqb = 0;
	FmIntv ik = {0, 0, 0}; i64 tp = 0, tps = 0;
	const int L = prm.MinSeedLength < 32 ? prm.MinSeedLength : 32;
	const u32 Lmask = L == 32 ? ~0u : (1u << L) - 1;
	// what is known about start S_ once its longest match [S_, S_ + LEN_) with X2_ occurrences is: the seed record (the hop follows from it)
#define SWEEP_SETTLE(S_, LEN_, X2_, X0_)                                                                                   \
	{                                                                                                                   \
		u32 rec_ = 0;                                                                                       \
		if ((LEN_) >= prm.MinSeedLength && (X2_) <= (u64)GSA_MAX_SEED_FREQ) { rec_ = (u32)(LEN_) | ((u32)(X2_) << 16); SWEEP_ST(x0o[S_] = (X0_);) } \
		SWEEP_ST(lf[S_] = rec_;)                                                                         \
	}
	while (!__all(mode == M_DONE)) {
		// ---- request phase: one pending request per lane, all lanes issue together ----
		u64 kk = 0, ll = 0; bool kn = true, ln = true;
		if (mode == M_FM || mode == M_BFM) {
			const u64 xr = mode == M_FM ? ik.x1 : ik.x0;      // forward extension counts on the reverse-strand interval, backward extension on the forward one
			const u64 k = xr - 1, l = xr - 1 + ik.x2;
			kn = (k == (u64)-1); ln = (l == (u64)-1);
			kk = kn ? 0 : k - (k >= di.primary); ll = ln ? 0 : l - (l >= di.primary);
		}
		const FmBlock bk = fm_load(di, kk >> 6), bl = fm_load(di, ll >> 6);
		struct __attribute__((packed, aligned(4))) W5 { u32 a, b, c, d, e; };
		// forward: 64 bases from tp on; backward (M_BACK): the three words that end with the base in front of the match
		const i64 bw_word = ((tps - 1) >> 4) - 2 > 0 ? ((tps - 1) >> 4) - 2 : 0;
		const W5 w5 = *(const W5 *)(di.ref2 + (mode == M_TEXT ? (tp >> 4) : (mode == M_BACK ? bw_word : 0)));
		ulonglong2 e0 = {0, 0}, e1 = {0, 0};
		if (E16) {
			const uint4 e = ((const uint4 *)(di.kmer ? di.kmer : (const u64 *)di.bwt))[mode == M_KMER ? kid : 0];
			e0.x = e.x; e0.y = e.y; e1.x = e.z; e1.y = e.w;
		} else {
			const ulonglong2 *pe = (const ulonglong2 *)((di.kmer ? di.kmer : (const u64 *)di.bwt) + (mode == M_KMER ? ((size_t)kid << 2) : 0));
			e0 = pe[0]; e1 = pe[1];
		}
		ulonglong2 l0 = {0, 0}, l1 = {0, 0};
		if (di.kmer_lo) {
			if (E16) { const uint4 e = ((const uint4 *)di.kmer_lo)[mode == M_KLO ? kid : 0]; l0.x = e.x; l0.y = e.y; l1.x = e.z; l1.y = e.w; }
			else { const ulonglong2 *pe = (const ulonglong2 *)(di.kmer_lo + (mode == M_KLO ? ((size_t)kid << 2) : 0)); l0 = pe[0]; l1 = pe[1]; }
		}
		// the line of the grouped presence table that answers for the start and up to three starts to its LEFT (pid = line, prole = role of s)
		uint4 pl0 = {~0u, ~0u, ~0u, ~0u}, pl1 = {~0u, ~0u, ~0u, ~0u};
		if (di.pres) { const uint4 *pp = (const uint4 *)di.pres + 2 * (size_t)(mode == M_KMER ? pid : 0); pl0 = pp[0]; pl1 = pp[1]; }
		const u64 sav = fm_locate(di, (mode == M_LOC || mode == M_LOC2) ? ik.x0 : 1);
		// ---- consume phase ----
		bool ended = false;
		if (mode == M_KMER) {
			// role r of the line: dword 2 r + (e >> 5), bit e & 31, e = pres4_bit(pqb, K, r) & 63 -- all four answers as a mask
			u32 pmask = 15u;
			if (di.pres) {
				const u32 e0_ = pres4_bit(pqb, di.pres_k, 0) & 63u, e1_ = pres4_bit(pqb, di.pres_k, 1) & 63u, e2_ = pres4_bit(pqb, di.pres_k, 2) & 63u, e3_ = pres4_bit(pqb, di.pres_k, 3) & 63u;
				pmask = ((((e0_ & 32u) ? pl0.y : pl0.x) >> (e0_ & 31u)) & 1u) | (((((e1_ & 32u) ? pl0.w : pl0.z) >> (e1_ & 31u)) & 1u) << 1)
				      | (((((e2_ & 32u) ? pl1.y : pl1.x) >> (e2_ & 31u)) & 1u) << 2) | (((((e3_ & 32u) ? pl1.w : pl1.z) >> (e3_ & 31u)) & 1u) << 3);
			}
			auto present = [&](int r) -> bool { return (pmask >> r) & 1u; };
			if (!present(prole)) {
				// the first MinSeedLength bases of s do not occur: no seed, nothing to extend; the same line answers for the starts to
				// the left while they pass the N / length tests (their own hop is 1 as well when they do not)
				SWEEP_ST(lf[s] = 0;) cur = s - 1; have = HV_NONE; mode = M_ADV;
				for (int r = prole - 1; r >= 0 && cur >= seg_a; r--) {
					const u32 nb = q_nbits32(qn_l, cur);
					if ((nb & 1u) || cur + prm.MinSeedLength > clen || (nb & Lmask) != 0) break;      // (the advance step settles those)
					if (present(r)) break;                                                            // occurs: needs its own search
					SWEEP_ST(lf[cur] = 0;) cur--;
				}
			} else {
				const bool hit = e1.x != 0;         // absent k-mer: the match is shorter than k
				if (hit) { ik.x0 = e0.x; ik.x1 = e0.y; ik.x2 = e1.x; pos = s + di.kmer_k; }
				mode = M_FM;
				if (hit && ik.x2 == 1) { tp = (i64)(e1.y - 1) + di.kmer_k; mode = M_TEXT; }
				if (!hit && di.kmer_lo) { kid = kid & ((1u << (2 * di.kmer_lo_k)) - 1); mode = M_KLO; }
			}
		} else if (mode == M_KLO) {
			const bool hit = l1.x != 0;
			if (hit) { ik.x0 = l0.x; ik.x1 = l0.y; ik.x2 = l1.x; pos = s + di.kmer_lo_k; }
			mode = M_FM;
			if (hit && ik.x2 == 1) { tp = (i64)(l1.y - 1) + di.kmer_lo_k; mode = M_TEXT; }
		} else if (mode == M_LOC) {
			tp = (i64)sav + (pos - s); mode = M_TEXT;
		} else if (mode == M_TEXT) {
			int got = text_match32(w5.a, w5.b, w5.c, tp, (i64)di.seq_len, qp_l, qn_l, pos, clen);
			if (got == 32) got += text_match32(w5.c, w5.d, w5.e, tp + 32, (i64)di.seq_len, qp_l, qn_l, pos + 32, clen);
			pos += got; tp += got;
			ended = got < 64;
		} else if (mode == M_FM || mode == M_BFM) {
			// one Occ step for both directions (a wave has lanes in either most of the time: the step is the heaviest block of the loop).
			// Forward: append q[pos] to [s, pos).  Backward: prepend q[cur] to the match [cur + 1, e_end) -- the reference's forward step
			// on the mirrored bi-interval with the complementary base
			const bool bw = mode == M_BFM;
			const int qi = bw ? cur : (pos < clen ? pos : 0);
			const bool can = bw || (pos < clen && !q_isn(qn_l, qi));
			const int code = q_code(qp_l, qi);
			FmIntv m = { bw ? ik.x1 : ik.x0, bw ? ik.x0 : ik.x1, ik.x2 };
			const bool ok = can && fm_extend_loaded(di, m, bw ? 3 - code : code, bk, bl, kk, ll, kn, ln, blk);
			if (ok) { ik.x0 = bw ? m.x1 : m.x0; ik.x1 = bw ? m.x0 : m.x1; ik.x2 = m.x2; }
			if (!bw) {
				ended = !ok;
				if (ok) { pos++; if (ik.x2 == 1) mode = M_LOC; }
			} else {
				all_blocks += blk; blk = 0;
				if (ok) {
					SWEEP_SETTLE(cur, e_end - cur, ik.x2, ik.x0)
					cur--;
					mode = ik.x2 == 1 ? M_LOC2 : M_ADV;       // one occurrence left: from here on the text itself answers
				} else { have = HV_NONE; mode = M_ADV; }      // (L(cur) < e_end - cur: search forward from cur)
			}
		} else if (mode == M_LOC2) {
			tps = (i64)sav; have = HV_UNIQ; mode = M_ADV;      // text position of start cur + 1
		} else if (mode == M_BACK) {
			// the match [cur + 1, e_end) sits once in the text, at tps: start cur - t extends it iff the t + 1 bases in front agree
			int n = cur - seg_a + 1; if (n > 32) n = 32; if ((i64)n > tps) n = (int)tps;
			int nbk = 0;
			if (n > 0) {
				const int off = (int)(tps - n - (bw_word << 4));                 // first compared text base inside the three words (0 .. 47)
				const u64 lo64 = (u64)w5.a | ((u64)w5.b << 32), hi64 = (u64)w5.c;
				const int sh = off * 2;
				const u64 T = sh == 0 ? lo64 : (sh < 64 ? (lo64 >> sh) | (hi64 << (64 - sh)) : (hi64 >> (sh - 64)));
				const u64 Q = q_bits64(qp_l, cur - n + 1);
				const u64 msk = n == 32 ? ~0ull : ((1ull << (2 * n)) - 1);
				const u64 d = (Q ^ T) & msk;
				const u64 dm = (d | (d >> 1)) & 0x5555555555555555ull;
				const u32 nm = q_nbits32(qn_l, cur - n + 1) & (n == 32 ? ~0u : ((1u << n) - 1));
				const int im = dm ? (63 - __clzll((long long)dm)) >> 1 : -1, in_ = nm ? 31 - __clz((int)nm) : -1;
				nbk = n - 1 - (im > in_ ? im : in_);
			}
			// the nbk starts [cur - nbk + 1, cur] are settled at once: start a + i matches [a + i, e_end) once, at text position tps - nbk + i.
			// Four starts per store (dword-aligned 16-byte stores: the records of a run are consecutive) -- a lane in unique sequence
			// settles 32 starts per iteration, and one scalar store per array and start was most of what the kernel asked of the L2
			{
				const int a = cur - nbk + 1;
				struct __attribute__((packed, aligned(4))) R4 { u32 v[4]; };
				struct __attribute__((packed, aligned(8))) X4 { u64 v[4]; };
				int i = 0;
				for (; i + 4 <= nbk; i += 4) {
					R4 r; X4 x;
#pragma unroll
					for (int e = 0; e < 4; e++) {
						const int len_ = e_end - (a + i + e);
						r.v[e] = len_ >= prm.MinSeedLength ? (u32)len_ | (1u << 16) : 0u;      // (SWEEP_SETTLE with one occurrence; -sen does not come here)
						x.v[e] = SWEEP_POSFLAG | (u64)(tps - nbk + i + e);
					}
					SWEEP_ST(*(R4 *)(lf + a + i) = r; *(X4 *)(x0o + a + i) = x;)
				}
				for (; i < nbk; i++) { const int st = a + i; SWEEP_SETTLE(st, e_end - st, 1ull, SWEEP_POSFLAG | (u64)(tps - nbk + i)) }
			}
			cur -= nbk; tps -= nbk;
			if (nbk < n || n <= 0) have = HV_NONE;                                  // stopped by a mismatch, an N or the start of the text
			mode = M_ADV;
		}
		if (ended) {
			// the forward search from s is through: [s, pos) with ik.x2 occurrences (ik.x0 = first row)
			const int len = pos - s;
			SWEEP_SETTLE(s, len, ik.x2, ik.x0)
			all_blocks += blk; blk = 0;
			cur = s - 1; e_end = pos;
			if (mode == M_TEXT) { have = HV_UNIQ; tps = tp - (i64)len; }            // (tp is the text position of pos)
			else have = (len >= 1 && ik.x2 >= 1) ? HV_MULTI : HV_NONE;
			mode = M_ADV;
		}
		// ---- what to do about start cur ----
		for (int step = 0; step < 3 && mode == M_ADV; step++) {
			if (cur < seg_a) {
				// through with the segment: the next one of the wave's chunks (LDS counter; M_DONE when they are all handed out)
				const u32 id = atomicAdd(&s_next, 1u);
				if (id >= n_seg) { mode = M_DONE; break; }
				const int ch = (int)id / spc, sg = (int)id - ch * spc;
				const u32 chunk = chunk_list ? chunk_list[slot0 + ch] : slot0 + ch;
				const i64 c0 = (i64)chunk * GSA_CHUNK;
				clen = (int)((i64)qlen - c0 < GSA_CHUNK ? (i64)qlen - c0 : GSA_CHUNK);
				qp_l = qp[ch]; qn_l = qn[ch]; lf = dn_lf + (size_t)(slot0 + ch) * GSA_CHUNK; x0o = dn_x0 + (size_t)(slot0 + ch) * GSA_CHUNK;
				seg_a = sg * seg; cur = (seg_a + seg < clen ? seg_a + seg : clen) - 1; have = HV_NONE;
				continue;
			}
			const u32 nb = q_nbits32(qn_l, cur);
			if (nb & 1u) { SWEEP_ST(lf[cur] = 0;) have = HV_NONE; cur--; continue; }      // ambiguous base: no search from here, nothing extends over it
			if (have == HV_UNIQ) { mode = M_BACK; break; }
			if (have == HV_MULTI) { mode = M_BFM; break; }
			if (cur + prm.MinSeedLength > clen || (nb & Lmask) != 0) { SWEEP_ST(lf[cur] = 0;) cur--; continue; }      // MinSeedLength out of reach
			s = cur; ik = fm_init(di, q_code(qp_l, s)); pos = s + 1; blk = 0; mode = M_FM;
			if (di.kmer_k > 1 && s + di.kmer_k <= clen && (nb & ((1u << di.kmer_k) - 1)) == 0) {
				kid = (u32)(q_bits64(qp_l, s) & ((1ull << (2 * di.kmer_k)) - 1)); mode = M_KMER;
				// presence: the group that starts three bases to the left (as far as the chunk goes), s in its last role
				prole = s >= 3 ? 3 : s;
				pqb = q_bits64(qp_l, s - prole);
				pid = di.pres_k ? pres4_line(pqb, di.pres_k) : 0;
			} else if (di.kmer_lo && s + di.kmer_lo_k <= clen && (nb & ((1u << di.kmer_lo_k) - 1)) == 0) {
				kid = (u32)(q_bits64(qp_l, s) & ((1ull << (2 * di.kmer_lo_k)) - 1)); mode = M_KLO;
			}
		}
	}
#undef SWEEP_SETTLE
	for (int o = 32; o; o >>= 1) all_blocks += __shfl_down(all_blocks, o);
	if ((j & 63) == 0 && all_blocks) atomicAdd((unsigned long long *)&cnt[CNT_OCCBLK_ALL], (unsigned long long)all_blocks);
}

// The chain of a dense chunk: orbit of 0 under p -> next(p) (from the start's record), then the accepted on-chain matches go to
// the chunk's candidate segment in the layout k_seed_wg leaves (so everything downstream is the same).  Round 3 marked the orbit by
// pointer doubling over all 10 000 positions (14 rounds: 2.8 ms per 250 Mb of dense chunks, a fifth of their seed stage).  next() only
// moves forward, so the chunk is cut into 256 blocks of 40 positions, one per thread:
//   1. every thread resolves its block from right to left: ex[p] = where the path through p LEAVES the block (40 dependent LDS reads);
//   2. one thread follows 0 -> ex[0] -> ex[ex[0]] ... (at most one step per block) and notes where the path enters each block;
//   3. every thread whose block the path enters walks it inside the block and marks the positions.
#define RS_B 40                 // positions per thread
#define RS_XS 42                // block stride of ex[] (u16; 21 dwords: odd, so the threads' blocks fall into different banks)
#define RS_HS 44                // block stride of hopb[] (u8; 11 dwords)
static_assert(256 * RS_B >= GSA_CHUNK, "one block per thread");
__global__ void __launch_bounds__(256) k_dense_resolve(const u32 *__restrict__ chunk_list, u32 n_total_chunks, i32 qlen, int sen, const u32 *__restrict__ dn_lf,
                                                        const u64 *__restrict__ dn_x0, u64 *cnt, i32 *cand_s, i32 *cand_len, u64 *cand_x0, i32 *cand_freq, u32 cand_cap, u32 *cand_cnt,
                                                        u32 *onpath, i32 *chunk_hits, u64 *hcnt, i32 *chunk_base)
{
	__shared__ uint16_t ex[256 * RS_XS];        // first position outside p's block on the path through p (0xffff: outside the chunk)
	__shared__ uint8_t hopb[256 * RS_HS];       // next(p) - p where that stays inside p's block, else 0
	__shared__ uint16_t s_entry[256];           // where the path from 0 enters the block (0xffff: it jumps over it)
	__shared__ u32 bits[PATH_WORDS];
	__shared__ u32 s_n, s_hits;
	__shared__ int s_last;
	const u32 slot = blockIdx.x, chunk = chunk_list ? chunk_list[slot] : slot;
	const int j = threadIdx.x;
	const i64 c0 = (i64)chunk * GSA_CHUNK;
	const int clen = (int)((i64)qlen - c0 < GSA_CHUNK ? (i64)qlen - c0 : GSA_CHUNK);
	const u32 *lf = dn_lf + (size_t)slot * GSA_CHUNK; const u64 *x0 = dn_x0 + (size_t)slot * GSA_CHUNK;
	// next(p): behind an accepted match (five bases on with -sen), else the next base -- the record says which (GSAlign.cpp:61-94)
	for (int p = j; p < clen; p += 256) {
		const u32 rec = lf[p];
		const int hop = rec ? (sen ? 5 : (int)(rec & 0xffffu) + 1) : 1, t = p + hop;
		const int b = p / RS_B, i = p - b * RS_B, bend = (b + 1) * RS_B < clen ? (b + 1) * RS_B : clen;
		ex[b * RS_XS + i] = (uint16_t)(t < clen ? t : 0xffff);
		hopb[b * RS_HS + i] = (uint8_t)(t < bend ? hop : 0);
	}
	for (int w = j; w < PATH_WORDS; w += 256) bits[w] = 0u;
	s_entry[j] = 0xffff;
	if (j == 0) { s_n = 0; s_hits = 0; }
	__syncthreads();
	{
		const int n = clen - j * RS_B < RS_B ? clen - j * RS_B : RS_B;      // (<= 0: the chunk ends in front of this block)
		for (int i = n - 1; i >= 0; i--) { const int h = hopb[j * RS_HS + i]; if (h) ex[j * RS_XS + i] = ex[j * RS_XS + i + h]; }
	}
	__syncthreads();
	if (j == 0) for (int e = 0; e != 0xffff;) { const int b = e / RS_B; s_entry[b] = (uint16_t)e; e = ex[b * RS_XS + (e - b * RS_B)]; }
	__syncthreads();
	if (s_entry[j] != 0xffff)
		for (int i = (int)s_entry[j] - j * RS_B;;) {
			const int p = j * RS_B + i;
			atomicOr(&bits[p >> 5], 1u << (p & 31));
			const int h = hopb[j * RS_HS + i];
			if (!h) break;
			i += h;
		}
	__syncthreads();
	const size_t cbase = (size_t)chunk * cand_cap;
	u32 h = 0;
	for (int p = j; p < clen; p += 256) {
		if (!((bits[p >> 5] >> (p & 31)) & 1u)) continue;
		const u32 rec = lf[p];
		if (!rec) continue;
		const u32 k = atomicAdd(&s_n, 1u);
		if (k < cand_cap) { cand_s[cbase + k] = (i32)(c0 + p); cand_len[cbase + k] = (i32)(rec & 0xffffu); cand_x0[cbase + k] = x0[p]; cand_freq[cbase + k] = (i32)(rec >> 16); h += rec >> 16; }
	}
	for (int o = 32; o; o >>= 1) h += __shfl_down(h, o);
	if ((j & 63) == 0 && h) atomicAdd(&s_hits, h);
	for (int w = j; w < PATH_WORDS; w += 256) onpath[(size_t)chunk * PATH_WORDS + w] = bits[w];
	__syncthreads();
	if (j == 0) {
		cand_cnt[chunk] = s_n < cand_cap ? s_n : cand_cap; lb_pub(&chunk_hits[chunk], (i32)s_hits); if (slot == 0) lb_pub(&chunk_hits[n_total_chunks], 0);
		atomicMax((unsigned long long *)&cnt[CNT_CAND], (unsigned long long)s_n);
		if (s_hits) atomicAdd((unsigned long long *)&cnt[CNT_HITS], (unsigned long long)s_hits);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (counters are device atomics; an agent-scope fence is an L2 write-back per workgroup here)
		s_last = atomicAdd((unsigned long long *)&cnt[CNT_DONE], 1ull) == (unsigned long long)gridDim.x - 1 ? 1 : 0;
	}
	__syncthreads();
	// the last workgroup puts the counters into pinned memory and leaves them at zero (as k_seed_wg does)
	if (s_last && j < 16) {
		hcnt[j] = j == CNT_DONE ? 0 : __hip_atomic_load(&cnt[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		cnt[j] = 0;
	}
	if (s_last) wg_exscan_hits<256>(chunk_hits, chunk_base, (int)n_total_chunks + 1);      // (every chunk's count: those of the speculative kernel too)
}

// ---------------------------------------------------------------------------
// Candidate -> seeds: keep the matches whose start lies on the true chain, locate
// every hit through the dense SA (a3: one read instead of ~31 dependent LF steps)
// and emit the 64-bit sort key ((PosDiff + qlen) << qbits) | qPos with the length.
// ---------------------------------------------------------------------------
// One workgroup per chunk.  Phase A: the on-chain candidates get their output ranges by a scan over the candidate index
// (deterministic order).  Phase B: one lane per HIT -- a seed with 100 hits is 100 lanes, not a 100-iteration loop of one
// lane -- which finds its candidate by binary search over the offsets (LDS), locates its row and ranks itself among the
// hits of its start by position: the tie-break of the (group, qPos) order, which the reference gets from a stable sort
// of the PosDiff order (the f rows of one start are re-read by f lanes: L1/L2 hits on the dense SA).
#define SEL_HASH 256
#ifndef SEL_BATCH
#define SEL_BATCH 1     // (experiment: > 1 = that many windows of 256 hits per round of k_seed_select's hit loop)
#endif
#ifndef SEL_TRIES
#define SEL_TRIES 4      // probes of the workgroup's LDS table of occupied PosDiff words before a hit goes to its word in HBM
#endif
// the coarse bitmap beside the PosDiff bitmap: bit (w >> 5) for bitmap word w (k_chain.hip, OpPdScan)
__device__ __forceinline__ void pd_coarse_set(u32 *pdcb, unsigned long long w)
{
	const unsigned long long blk = w >> 5; const u32 bit = 1u << (blk & 31);
	// (looked at first: the main diagonal of a whole contig sits in one block, and an unconditional atomic per workgroup queues on that word
	//  -- locate + order of a 250 Mb contig 0.31 -> 0.67 ms when tried; a kernel of its own with a thread per hit: 1.4 ms, the looks queue too)
#ifdef SEL_COARSE_PLAIN      // (experiment: a plain cached look instead of the agent-scope one)
	if (!(pdcb[blk >> 5] & bit)) atomicOr(&pdcb[blk >> 5], bit);
#else
	if (!(__hip_atomic_load(&pdcb[blk >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(&pdcb[blk >> 5], bit);
#endif
}
__global__ void __launch_bounds__(256) k_seed_select(DevIndex di, u32 cand_cap, const u32 *__restrict__ cand_cnt, const i32 *__restrict__ cand_s, const i32 *__restrict__ cand_len,
                                                      const u64 *__restrict__ cand_x0, const i32 *__restrict__ cand_freq, const u32 *__restrict__ onpath,
                                                      const i32 *__restrict__ hit_base, Bundle bnd, i32 s_off, int qbits, u64 *key, u32 *val, u32 *pdbm, u32 *pdcb, u32 lds_cand, uint8_t *pdby)
{
	// (bnd.lmax = length of the whole contig, s_off = contig position of the first chunk searched: not 0 when only a chunk range
	//  of the contig was seeded on this GPU, gsa_seed_chunks.  A bundle of contigs: the key's PosDiff is the true one of the
	//  chunk's contig plus that contig's stride, see Bundle)
	// Round 4: the hits leave this kernel in (qPos, rank) order -- chunk after chunk (the launch order), inside a chunk by the start position
	// of their candidate (the on-chain starts are marked in a bitmap over the chunk's positions: a candidate's place is the number of marked
	// starts below its own), inside a start by the rank of the hit.  Stage 2 then needs ONE stable sort by the group id alone (three 8-bit passes
	// instead of eight over the whole 57-bit key).  Off-chain candidates take no part.
	extern __shared__ u32 s_offs[];                    // [nc + 1] exclusive prefix of the hit counts of the on-chain candidates in start order | [nc] their candidate numbers
	u32 *s_ord = s_offs + lds_cand + 2;                // (lds_cand: the most candidates any chunk of THIS contig holds -- not the capacity of the segments)
	__shared__ unsigned long long s_w[SEL_HASH]; __shared__ u32 s_b[SEL_HASH];
	__shared__ u32 s_wsum[4], s_run, s_sb[GSA_CHUNK / 32 + 2], s_sbpre[GSA_CHUNK / 32 + 2];
	const u32 chunk = blockIdx.x, nc = cand_cnt[chunk];
	const size_t cbase = (size_t)chunk * cand_cap;
	const int j = threadIdx.x, lane = j & 63, wv = j >> 6;
	for (int t = j; t < SEL_HASH; t += 256) { s_w[t] = ~0ull; s_b[t] = 0; }
	for (int t = j; t < GSA_CHUNK / 32 + 2; t += 256) s_sb[t] = 0;
	if (j == 0) s_run = 0;
	__syncthreads();
	for (u32 i = j; i < nc; i += 256) {
		const i32 p = cand_s[cbase + i] - (i32)chunk * GSA_CHUNK;
		if ((onpath[(size_t)chunk * PATH_WORDS + (p >> 5)] >> (p & 31)) & 1u) atomicOr(&s_sb[p >> 5], 1u << (p & 31));
	}
	__syncthreads();
	if (wv == 0) {      // marked starts below each word: 313 words, five per lane
		constexpr int WPL = (GSA_CHUNK / 32 + 2 + 63) / 64;
		u32 c5 = 0;
		for (int k = 0; k < WPL; k++) { const int w = lane * WPL + k; if (w < GSA_CHUNK / 32 + 2) c5 += (u32)__popc(s_sb[w]); }
		u32 inc = c5;
		for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(inc, o); if (lane >= o) inc += t; }
		u32 run = inc - c5;
		for (int k = 0; k < WPL; k++) { const int w = lane * WPL + k; if (w < GSA_CHUNK / 32 + 2) { s_sbpre[w] = run; run += (u32)__popc(s_sb[w]); } }
		if (lane == 63) s_run = inc;      // on-chain candidates of the chunk
	}
	__syncthreads();
	const u32 n_on = s_run;
	for (u32 i = j; i < nc; i += 256) {
		const i32 p = cand_s[cbase + i] - (i32)chunk * GSA_CHUNK;
		if ((s_sb[p >> 5] >> (p & 31)) & 1u) {
			if ((onpath[(size_t)chunk * PATH_WORDS + (p >> 5)] >> (p & 31)) & 1u) {
				const u32 o = s_sbpre[p >> 5] + (u32)__popc(s_sb[p >> 5] & ((1u << (p & 31)) - 1u));
				s_ord[o] = i;
			}
		}
	}
	__syncthreads();
	if (j == 0) s_run = 0;
	__syncthreads();
	for (u32 i0 = 0; i0 < n_on; i0 += 256) {
		const u32 o = i0 + j;
		const u32 f = o < n_on ? (u32)cand_freq[cbase + s_ord[o]] : 0u;
		u32 inc = f;
		for (int oo = 1; oo < 64; oo <<= 1) { const u32 t = __shfl_up(inc, oo); if (lane >= oo) inc += t; }
		if (lane == 63) s_wsum[wv] = inc;
		__syncthreads();
		u32 wo = 0; for (int w = 0; w < wv; w++) wo += s_wsum[w];
		const u32 run = s_run;
		if (o < n_on) s_offs[o] = run + wo + inc - f;
		__syncthreads();
		if (j == 255) s_run = run + wo + inc;
		__syncthreads();
	}
	const u32 total = s_run;
	if (j == 0) s_offs[n_on] = total;
	__syncthreads();
	const u64 base = (u64)hit_base[chunk];
	i64 pd_base = bnd.lmax;                              // key = rPos - qPos + pd_base
	if (bnd.n) { const i32 ci = bnd.chunk_contig[chunk]; pd_base += (i64)bnd.off[ci] + (i64)ci * bnd.pds; }
#if SEL_BATCH > 1      // (experiment, round 5: SEL_BATCH windows of 256 hits per round -- their gathers and dense-SA reads in flight together, two barriers per round instead of per window)
	__shared__ unsigned long long s_rb[SEL_BATCH][256];
	for (u32 t0 = 0; t0 < total; t0 += 256 * SEL_BATCH) {
		u32 i_[SEL_BATCH], h_[SEL_BATCH], f_[SEL_BATCH], len_[SEL_BATCH]; i32 s_[SEL_BATCH]; u64 x0_[SEL_BATCH], r_[SEL_BATCH]; bool in_[SEL_BATCH];
#pragma unroll
		for (int u = 0; u < SEL_BATCH; u++) {
			const u32 t = t0 + (u32)u * 256 + j; in_[u] = t < total;
			const u32 tc = in_[u] ? t : total - 1;
			u32 lo = 0, hi = n_on;
			while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (s_offs[mid] <= tc) lo = mid; else hi = mid; }
			i_[u] = s_ord[lo]; h_[u] = tc - s_offs[lo];
		}
#pragma unroll
		for (int u = 0; u < SEL_BATCH; u++) { s_[u] = cand_s[cbase + i_[u]] + s_off; f_[u] = (u32)cand_freq[cbase + i_[u]]; len_[u] = (u32)cand_len[cbase + i_[u]]; x0_[u] = cand_x0[cbase + i_[u]]; }
#pragma unroll
		for (int u = 0; u < SEL_BATCH; u++) { const bool known = (x0_[u] >> 63) != 0; const u64 loc = fm_locate(di, known ? 0ull : x0_[u] + h_[u]); r_[u] = known ? (x0_[u] & ~(1ull << 63)) : loc; }
#pragma unroll
		for (int u = 0; u < SEL_BATCH; u++) s_rb[u][j] = r_[u];
		__syncthreads();
#pragma unroll
		for (int u = 0; u < SEL_BATCH; u++) if (in_[u]) {
			const u32 t = t0 + (u32)u * 256 + j, h = h_[u], f = f_[u]; const u64 r = r_[u], x0 = x0_[u]; const i32 s = s_[u];
			const i64 pd = (i64)r - s + pd_base;
			u32 rank = 0;
			if (f > 1) {
				const u32 first = t - h;
				for (u32 h2 = 0; h2 < f; h2++) {
					const u32 ts = first + h2;
					const u64 r2 = (ts >= t0 && ts < t0 + 256 * SEL_BATCH) ? s_rb[(ts - t0) >> 8][(ts - t0) & 255] : fm_locate(di, x0 + h2);
					rank += r2 < r ? 1u : 0u;
				}
			}
			const u64 at = base + (t - h) + rank;
			key[at] = ((u64)pd << qbits) | (u32)s;
			val[at] = len_[u] | (rank << 16);
			if (pdby) pdby[pd] = 1;
			else if (pdbm) {
				const unsigned long long w = (unsigned long long)(pd >> 5); const u32 bit = 1u << (pd & 31);
				int hh = (int)((w * 0x9E3779B1ull) >> 7) & (SEL_HASH - 1), tries = 0;
				for (; tries < SEL_TRIES; tries++, hh = (hh + 1) & (SEL_HASH - 1)) {
					const unsigned long long prev = atomicCAS(&s_w[hh], ~0ull, w);
					if (prev == ~0ull || prev == w) { atomicOr(&s_b[hh], bit); break; }
				}
				if (tries == SEL_TRIES) { atomicOr(&pdbm[w], bit); pd_coarse_set(pdcb, w); }
			}
		}
		__syncthreads();
	}
#else
	__shared__ unsigned long long s_r[256];                  // located positions of the 256 hits in flight: a hit ranks itself among its siblings from here
	for (u32 t0 = 0; t0 < total; t0 += 256) {
		const u32 t = t0 + j;
		u32 i = 0, h = 0, f = 0, len = 0; i32 s = 0; u64 x0 = 0, r = 0;
		if (t < total) {
			// the candidate whose range holds hit t: the last i with s_offs[i] <= t (empty ranges share their start with the next one)
			u32 lo = 0, hi = n_on;
			while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (s_offs[mid] <= t) lo = mid; else hi = mid; }
			i = s_ord[lo]; h = t - s_offs[lo];
			s = cand_s[cbase + i] + s_off; f = (u32)cand_freq[cbase + i]; len = (u32)cand_len[cbase + i]; x0 = cand_x0[cbase + i];
			r = (x0 >> 63) ? (x0 & ~(1ull << 63)) : fm_locate(di, x0 + h);      // (bit 63: a unique match whose text position the sweep already knows)
		}
		s_r[j] = r;
		__syncthreads();
		if (t < total) {
		const i64 pd = (i64)r - s + pd_base;
		u32 rank = 0;
		if (f > 1) {
			const u32 first = t - h;                            // hit index of sibling 0
			for (u32 h2 = 0; h2 < f; h2++) {
				const u32 ts = first + h2;
				const u64 r2 = (ts >= t0 && ts < t0 + 256) ? s_r[ts - t0] : fm_locate(di, x0 + h2);      // (siblings in another window of 256: the dense SA again)
				rank += r2 < r ? 1u : 0u;
			}
		}
		const u64 at = base + (t - h) + rank;                 // (the hits of a start in rank order: distinct text positions, so the ranks are a permutation)
		key[at] = ((u64)pd << qbits) | (u32)s;
		val[at] = len | (rank << 16);
		// occupied PosDiff values: groups without sorting by PosDiff (k_chain.hip).  Collected per workgroup in LDS, one
		// global OR per touched word at the end: a chunk's hits sit in two or three words and the whole contig's main
		// diagonal in one cache line -- an atomic (or even a look) per hit queues 75 k operations on that line.  Hits of
		// repeats scatter over the genome: after a few probes they go straight to their own (uncontended) word.
		// (round 5) where the hits scatter -- -sen: a chunk holds thousands of chance hits on as many words, and a device-scope atomic each is what `locate` then
		// costs (9.7 M of them in a 60 Mb bundle, 2.8 ms) -- a byte per PosDiff value takes a plain store; k_pd_pack makes the bitmap of it
		if (pdby) pdby[pd] = 1;
		else if (pdbm) {
			const unsigned long long w = (unsigned long long)(pd >> 5); const u32 bit = 1u << (pd & 31);
			int hh = (int)((w * 0x9E3779B1ull) >> 7) & (SEL_HASH - 1), tries = 0;
			for (; tries < SEL_TRIES; tries++, hh = (hh + 1) & (SEL_HASH - 1)) {
				const unsigned long long prev = atomicCAS(&s_w[hh], ~0ull, w);
				if (prev == ~0ull || prev == w) { atomicOr(&s_b[hh], bit); break; }
			}
			if (tries == SEL_TRIES) { atomicOr(&pdbm[w], bit); pd_coarse_set(pdcb, w); }
		}
		}
		__syncthreads();
	}
#endif
	if (pdbm && !pdby) {
		__syncthreads();
		for (int t = j; t < SEL_HASH; t += 256) if (s_w[t] != ~0ull) { atomicOr(&pdbm[s_w[t]], s_b[t]); pd_coarse_set(pdcb, s_w[t]); }
	}
}

// The byte map of occupied PosDiff values -> the bitmap and its coarse bitmap, and the bytes back to zero.  A workgroup per 1 024 bitmap words (32 KB of
// bytes, one coarse word): a thread reads the 32 bytes of a word, gathers their low bits (the bytes are 0 or 1: one multiplication per four), stores the
// word if it holds a hit and clears its bytes; the coarse bits are the waves' ballots.  The bitmap is all zero when the pass starts (the invariant of
// stage 2: k_pd_gather clears what a contig set), so plain stores do.
__device__ __forceinline__ u32 pd_nib(u32 x) { return ((x * 0x01020408u) >> 24) & 15u; }      // bytes b0..b3 in {0, 1} -> b0 | b1 << 1 | b2 << 2 | b3 << 3 (no two partial products share a bit)
__global__ void __launch_bounds__(256) k_pd_pack(uint8_t *pdby, i64 nw, u32 *pdbm, u32 *pdcb)
{
	__shared__ u32 s_c[4];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const i64 n_tiles = (nw + 1023) >> 10;
	for (i64 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		u32 cb = 0;
		uint4 a[4], b[4];
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const i64 w = (tile << 10) + k * 256 + tid;
			a[k] = make_uint4(0, 0, 0, 0); b[k] = a[k];
			if (w < nw) { a[k] = ((const uint4 *)pdby)[2 * w]; b[k] = ((const uint4 *)pdby)[2 * w + 1]; }
		}
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const i64 w = (tile << 10) + k * 256 + tid;
			const bool any = (a[k].x | a[k].y | a[k].z | a[k].w | b[k].x | b[k].y | b[k].z | b[k].w) != 0;
			if (any) {
				const u32 word = pd_nib(a[k].x) | (pd_nib(a[k].y) << 4) | (pd_nib(a[k].z) << 8) | (pd_nib(a[k].w) << 12) | (pd_nib(b[k].x) << 16) | (pd_nib(b[k].y) << 20) | (pd_nib(b[k].z) << 24) | (pd_nib(b[k].w) << 28);
				pdbm[w] = word;
				((uint4 *)pdby)[2 * w] = make_uint4(0, 0, 0, 0); ((uint4 *)pdby)[2 * w + 1] = make_uint4(0, 0, 0, 0);
			}
			const unsigned long long m = __ballot(any);      // 64 words = two blocks of 32
			cb |= ((m & 0xffffffffull) ? 1u : 0u) << (k * 8 + wv * 2);
			cb |= ((m >> 32) ? 1u : 0u) << (k * 8 + wv * 2 + 1);
		}
		if (lane == 0) s_c[wv] = cb;
		__syncthreads();
		if (tid == 0) { const u32 v = s_c[0] | s_c[1] | s_c[2] | s_c[3]; if (v) pdcb[tile] = v; }
		__syncthreads();
	}
}

// PosDiff bitmap bits of hits that arrived from another GPU (gsa_import_hits)
__global__ void __launch_bounds__(256) k_pd_from_keys(i64 n, const u64 *__restrict__ key, int qbits, u32 *pdbm, u32 *pdcb)
{
	const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const u64 pd = key[i] >> qbits;
	atomicOr(&pdbm[pd >> 5], 1u << (pd & 31)); pd_coarse_set(pdcb, (unsigned long long)(pd >> 5));
}

// Accounting build only: the LF steps bwt_sa would walk for every located hit (the row is sampled every 32 ROWS, so the
// walk ends at the first row divisible by 32: bwt_search.cpp:129-139).  The hot path reads the dense SA instead; this
// is the algorithmic figure of SURVEY.md section 8(d).
__global__ void __launch_bounds__(256) k_count_lf(DevIndex di, u32 cand_cap, const u32 *__restrict__ cand_cnt, const i32 *__restrict__ cand_s, const u64 *__restrict__ cand_x0,
                                                   const i32 *__restrict__ cand_freq, const u32 *__restrict__ onpath, unsigned long long *out)
{
	const u32 chunk = blockIdx.x, nc = cand_cnt[chunk];
	const size_t cbase = (size_t)chunk * cand_cap;
	unsigned long long steps = 0;
	for (u32 i = threadIdx.x; i < nc; i += blockDim.x) {
		const i32 p = cand_s[cbase + i] - (i32)chunk * GSA_CHUNK;
		if (!((onpath[(size_t)chunk * PATH_WORDS + (p >> 5)] >> (p & 31)) & 1u)) continue;
		const u32 f = (u32)cand_freq[cbase + i];
		for (u32 h = 0; h < f; h++) { u32 st = 0; (void)fm_locate_walk(di, cand_x0[cbase + i] + h, st); steps += st; }
	}
	for (int o = 32; o; o >>= 1) steps += __shfl_down(steps, o);
	if ((threadIdx.x & 63) == 0 && steps) atomicAdd(out, steps);
}

// sorted keys -> SoA seeds + group ids (SeedGrouping, a6): one fused pass (gsa_scan.h); a new group
// starts where PosDiff jumps by more than MaxIndelSize
struct OpDecodeGroup {
	i64 n; const u64 *key; const u32 *val; Bundle bnd; int qbits; i32 max_indel;
	i32 *s_q, *s_len; i64 *s_r; i32 *s_gid, *g_beg, *mail;
	__device__ i32 value(i64 i, int) const
	{
		if (i == 0) return 1;
		const i64 pd = (i64)(key[i] >> qbits), pd0 = (i64)(key[i - 1] >> qbits);      // (a bundle: the stride between contigs exceeds max_indel)
		return (pd - pd0 > max_indel) ? 1 : 0;
	}
	__device__ void emit(i64 i, const i32 *v, const i32 *ex) const
	{
		const u64 k = key[i];
		const i32 qp = (i32)(k & ((1ull << qbits) - 1)); i64 pd = (i64)(k >> qbits) - bnd.lmax;
		if (bnd.n) { const i32 ci = bnd.chunk_contig[qp / GSA_CHUNK]; pd -= (i64)bnd.off[ci] + (i64)ci * bnd.pds; }      // rPos - qp
		s_q[i] = qp; s_len[i] = (i32)(val[i] & 0xffffu); s_r[i] = pd + qp;
		const i32 g = ex[0] + v[0] - 1;
		s_gid[i] = g;
		if (v[0]) g_beg[g] = (i32)i;
	}
	__device__ void done(const i32 *t) const { g_beg[t[0]] = (i32)n; mail[M_NG] = t[0]; }
};

// ---------------------------------------------------------------------------
// Dense SA (index upload time).  The on-disk SA keeps every 32nd ROW; a walk from
// sampled row k (SA = p) visits rows with SA p-1, p-2, ... and stops at the next
// sampled row, so the walks started at all sampled rows together touch every row
// exactly once: 2G LF steps in total, one lane per sampled row.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_densify_sa(DevIndex di, u64 n_sa, u32 *d32, u64 *d64)
{
	const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_sa) return;
	u64 k = i << 5;
	u64 p = i == 0 ? di.seq_len : di.sa[i];
	if (d32) d32[k] = (u32)(i == 0 ? 0xFFFFFFFFu : p); else d64[k] = i == 0 ? (u64)-1 : p;
	for (;;) {
		k = fm_lf(di, k); p -= 1;
		if ((k & 31) == 0) break;
		if (d32) d32[k] = (u32)p; else d64[k] = p;
	}
}

// k-mer jump table: entry id = the interval BWT_Search holds after matching the k bases of id
// (base t in bits 2t..2t+1); x2 = 0 when the walk dies earlier (then the stepwise walk is used).
// Needs the dense SA (unique k-mers carry their text position).
__global__ void __launch_bounds__(256) k_build_kmer(DevIndex di, int k, u64 *tab, int e16)
{
	const u64 id = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (id >= (1ull << (2 * k))) return;
	FmIntv ik = fm_init(di, (int)(id & 3));                  // base t of the k-mer = bits 2t..2t+1 (same packing as the query in LDS)
	u32 blk = 0; bool alive = true;
	for (int t = 1; t < k && alive; t++) alive = fm_extend(di, ik, (int)((id >> (2 * t)) & 3), blk);
	const u64 loc1 = (alive && ik.x2 == 1) ? fm_locate(di, ik.x0) + 1 : 0;      // unique k-mer: where it is in the text (+1; saves the SA read)
	if (e16) { ((uint4 *)tab)[id] = make_uint4((u32)ik.x0, (u32)ik.x1, alive ? (u32)ik.x2 : 0u, (u32)loc1); return; }
	u64 *e = tab + ((size_t)id << 2);
	e[0] = ik.x0; e[1] = ik.x1; e[2] = alive ? ik.x2 : 0; e[3] = loc1;
}

// RefSequence from the .pac bytes (RestoreReferenceInfo, bwt_index.cpp:229-264; packing: bntseq.c _get_pac -- base f in byte f >> 2, bits ((~f & 3) << 1)): the forward
// strand, then its reverse complement.  A thread takes one pac byte = four bases: one 4-byte store forward, one 4-byte store (when G is a multiple of four; else bytes) backward
__global__ void __launch_bounds__(256) k_unpack_pac(const uint8_t *__restrict__ pac, i64 G, uint8_t *ref)
{
	const i64 G2 = 2 * G;
	for (i64 b = (i64)blockIdx.x * blockDim.x + threadIdx.x; b * 4 < G; b += (i64)gridDim.x * blockDim.x) {
		const u32 v = pac[b];
		const i64 f0 = b * 4;
#pragma unroll
		for (int t = 0; t < 4; t++) {
			const i64 f = f0 + t;
			if (f < G) { const u32 code = (v >> ((~(u32)t & 3u) << 1)) & 3u; ref[f] = (uint8_t)"ACGT"[code]; ref[G2 - 1 - f] = (uint8_t)"TGCA"[code]; }
		}
	}
}
int unpack_pac(gsa_ctx *c, const uint8_t *d_pac, i64 G, uint8_t *d_ref)
{
	const u64 n = ((u64)G + 3) / 4;
	hipLaunchKernelGGL(k_unpack_pac, dim3(grid_for(std::min<u64>(n, 1ull << 28), 256)), dim3(256), 0, c->stream, d_pac, G, d_ref);
	GSA_CHECK(c, hipGetLastError());
	return GSA_OK;
}

// 2-bit packed copy of RefSequence (16 bases per word, LSB first) for the unique-interval text comparison
__global__ void __launch_bounds__(256) k_pack_ref(const uint8_t *__restrict__ ref, u64 n, u32 *out, u64 words)
{
	const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= words) return;
	u32 v = 0;
	for (int t = 0; t < 16; t++) { const u64 p = w * 16 + t; if (p < n) v |= ((u32)gsa_nt4(ref[p]) & 3) << (2 * t); }
	out[w] = v;
}

// (layout of the grouped presence table: comment at pres4_line, top of this file)
// (grid-stride: a launch holds at most 2^32 - 1 work-items -- the dispatch packet's grid size is 32 bits wide -- and a human index has 6.2 G text
//  positions.  Until round 5 this kernel was launched with one work-item per position: the runtime took the count modulo 2^32, only the first 1.86 G
//  positions of a 3.08 Gbp index were entered, and a 15-mer whose occurrences all lie behind them -- one in five -- was reported ABSENT: the search from
//  such a start ended without a seed.  Found by the first oracle comparison on the native index, tests/human_scale_check.py.)
__global__ void __launch_bounds__(256) k_build_pres(const u32 *__restrict__ ref2, u64 seq_len, int k, u32 *bm)
{
	for (u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x; p + (u64)k <= seq_len; p += (u64)gridDim.x * blockDim.x) {
	const u64 w = p >> 4;
	const u64 X = funnel64(ref2[w], ref2[w + 1], ref2[w + 2], (int)(p & 15) << 1) & ((1ull << (2 * k)) - 1);      // the k-mer at p, base t at bits 2t
#pragma unroll
	for (int i = 0; i < 4; i++) {
		// as member i of the group that starts at p - i: its core is X[3-i .. k-i), the rest are X's first 3-i and last i bases
		const u32 line = (u32)((X >> (2 * (3 - i))) & ((1ull << (2 * (k - 3))) - 1));
		const u32 head = (u32)X & ((1u << (2 * (3 - i))) - 1), tail = (u32)(X >> (2 * (k - i))) & ((1u << (2 * i)) - 1);
		const u32 bit = (u32)i * 64u + (head | (tail << (2 * (3 - i))));
		atomicOr(&bm[(size_t)line * 8 + (bit >> 5)], 1u << (bit & 31));
	}
	}
}

// The same table from the k-mer jump table, when that holds k-mers of exactly this length (the default: -slen 15 against a text of more than 4^13
// rows): a k-mer occurs iff its entry's interval is not empty, so 4^k entries are read in order instead of 2G text positions (a human index: 1.07 G
// entries against 6.2 G positions; the scan's four atomics per position were 0.9 s of gsa_create there).
__global__ void __launch_bounds__(256) k_pres_from_kmer(const u64 *__restrict__ tab, int e16, int k, u32 *bm)
{
	const u64 n = 1ull << (2 * k);
	for (u64 X = (u64)blockIdx.x * blockDim.x + threadIdx.x; X < n; X += (u64)gridDim.x * blockDim.x) {
		const bool occurs = e16 ? ((const uint4 *)tab)[X].z != 0u : tab[(X << 2) + 2] != 0ull;
		if (!occurs) continue;
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const u32 line = (u32)((X >> (2 * (3 - i))) & ((1ull << (2 * (k - 3))) - 1));
			const u32 head = (u32)X & ((1u << (2 * (3 - i))) - 1), tail = (u32)(X >> (2 * (k - i))) & ((1u << (2 * i)) - 1);
			const u32 bit = (u32)i * 64u + (head | (tail << (2 * (3 - i))));
			atomicOr(&bm[(size_t)line * 8 + (bit >> 5)], 1u << (bit & 31));
		}
	}
}

// the short companion of the k-mer table (DevIndex::kmer_lo): MinSeedLength bases, when that is less than kmer_k
static int build_kmer_lo(gsa_ctx *c)
{
	const int k = c->prm.MinSeedLength;
	if (c->di.kmer_lo && c->di.kmer_lo_k == k) return GSA_OK;
	c->di.kmer_lo = nullptr; c->di.kmer_lo_k = 0;
	if (!c->di.kmer || k < 8 || k >= c->di.kmer_k || k > 13) return GSA_OK;
	const size_t n = (size_t)1 << (2 * k);
	if (!dev_ensure<u64>(c, c->d_kmer_lo, c->di.kmer_e16 ? n * 2 : n * 4, true)) return GSA_ERR_NOMEM;
	hipLaunchKernelGGL(k_build_kmer, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, c->di, k, c->d_kmer_lo.as<u64>(), c->di.kmer_e16);
	GSA_CHECK(c, hipGetLastError());
	GSA_CHECK(c, hipStreamSynchronize(c->stream));
	c->di.kmer_lo = c->d_kmer_lo.as<u64>(); c->di.kmer_lo_k = k;
	return GSA_OK;
}

int build_presence(gsa_ctx *c)
{
	if (!c->di.ref2) return GSA_OK;                       // (gsa_create sets the parameters after the index is up)
	if (int rcl = build_kmer_lo(c)) return rcl;
	int k = c->prm.MinSeedLength < 16 ? c->prm.MinSeedLength : 16;
	if (k == c->di.pres_k && c->di.pres) return GSA_OK;
	c->di.pres = nullptr; c->di.pres_k = 0;
	if (k < 8) return GSA_OK;                             // short seeds: nearly every k-mer present, nothing to gain
	const size_t words = ((size_t)1 << (2 * (k - 3))) * 8;      // 4^(k-3) lines of 32 bytes
	if (!dev_ensure<u32>(c, c->d_pres, words, true)) return GSA_ERR_NOMEM;
	GSA_CHECK(c, hipMemsetAsync(c->d_pres.p, 0, words * 4, c->stream));
	if (c->di.kmer && c->di.kmer_k == k && c->opt.pres_from_kmer)
		hipLaunchKernelGGL(k_pres_from_kmer, dim3(grid_for(std::min<u64>(1ull << (2 * k), 1ull << 28), 256)), dim3(256), 0, c->stream, c->di.kmer, c->di.kmer_e16, k, c->d_pres.as<u32>());
	else
		hipLaunchKernelGGL(k_build_pres, dim3(grid_for(std::min<u64>(c->di.seq_len, 1ull << 30), 256)), dim3(256), 0, c->stream, c->di.ref2, c->di.seq_len, k, c->d_pres.as<u32>());
	GSA_CHECK(c, hipGetLastError());
	GSA_CHECK(c, hipStreamSynchronize(c->stream));
	c->di.pres = c->d_pres.as<u32>(); c->di.pres_k = k;
	return GSA_OK;
}

// ---- Occ blocks: the reference's interleaved layout (128 rows per 64-byte block: four u64 counts + 128 symbols,
// bwt_search.cpp:69-119) regrouped into 64 rows per 32-byte block (FmBlock, gsa_fm.h) ----
__global__ void __launch_bounds__(256) k_occ_base(const uint4 *__restrict__ src, u64 n_super, int shift, u64 *base)
{
	const u64 sb = (u64)blockIdx.x * 256 + threadIdx.x;
	if (sb >= n_super) return;
	const uint4 *p = src + (((sb << shift) >> 1) << 2);              // (a super-block starts on an even block: a header of the reference)
	const uint4 c0 = p[0], c1 = p[1];
	base[4 * sb] = ((u64)c0.y << 32) | c0.x; base[4 * sb + 1] = ((u64)c0.w << 32) | c0.z; base[4 * sb + 2] = ((u64)c1.y << 32) | c1.x; base[4 * sb + 3] = ((u64)c1.w << 32) | c1.z;
}
__global__ void __launch_bounds__(256) k_occ_relayout(const uint4 *__restrict__ src, u64 n_blocks, const u64 *__restrict__ base, int shift, uint4 *dst)
{
	const u64 b = (u64)blockIdx.x * 256 + threadIdx.x;
	if (b >= n_blocks) return;
	const uint4 *p = src + ((b >> 1) << 2);
	const uint4 c0 = p[0], c1 = p[1], w = p[2 + (b & 1)];
	u64 ca = ((u64)c0.y << 32) | c0.x, cc = ((u64)c0.w << 32) | c0.z, cg = ((u64)c1.y << 32) | c1.x, ct = ((u64)c1.w << 32) | c1.z;
	if (b & 1) {                                                      // the second half of a reference block: its header + its first 64 symbols
		const uint4 w0 = p[2];
		const u64 M = 0x5555555555555555ull;
		u32 n1 = 0, n2 = 0, n3 = 0;
		for (int J = 0; J < 2; J++) {
			const u64 W = J ? (((u64)w0.z << 32) | w0.w) : (((u64)w0.x << 32) | w0.y);
			const u64 lo = W & M, hi = (W >> 1) & M;
			n3 += __popcll(hi & lo); n2 += __popcll(hi & ~lo & M); n1 += __popcll(~hi & lo & M);
		}
		ca += 64 - n1 - n2 - n3; cc += n1; cg += n2; ct += n3;
	}
	if (base) { const u64 *sb = base + ((b >> shift) << 2); ca -= sb[0]; cc -= sb[1]; cg -= sb[2]; ct -= sb[3]; }
	dst[2 * b] = make_uint4((u32)ca, (u32)cc, (u32)cg, (u32)ct);
	dst[2 * b + 1] = w;
}

// `ref_layout` = the index file's bwt words on the device, whole 64-byte blocks, zero-padded
int build_occ(gsa_ctx *c, const void *ref_layout, u64 n_blocks128)
{
	const u64 n_blocks = 2 * n_blocks128;
	const bool wide = c->force_wide || c->di.seq_len >= 0xFFFFFF00ull;
	// super-blocks of 2^31 rows where the counts need them; the forced-wide layout of the test-suite uses 2^16 rows so that small
	// texts have several super-blocks and their relative counts really are relative
	const int shift = c->di.seq_len >= 0xFFFFFF00ull ? 25 : 10;
	if (!dev_ensure<uint4>(c, c->d_bwt, 2 * n_blocks + 4, true)) return GSA_ERR_NOMEM;
	GSA_CHECK(c, hipMemsetAsync(c->d_bwt.p, 0, (2 * n_blocks + 4) * sizeof(uint4), c->stream));
	u64 *base = nullptr;
	if (wide) {
		const u64 n_super = (n_blocks >> shift) + 1;
		if (!dev_ensure<u64>(c, c->d_occ_base, 4 * n_super, true)) return GSA_ERR_NOMEM;
		base = c->d_occ_base.as<u64>();
		hipLaunchKernelGGL(k_occ_base, dim3(grid_for(n_super, 256)), dim3(256), 0, c->stream, (const uint4 *)ref_layout, n_super, shift, base);
		GSA_CHECK(c, hipGetLastError());
	}
	hipLaunchKernelGGL(k_occ_relayout, dim3(grid_for(n_blocks, 256)), dim3(256), 0, c->stream, (const uint4 *)ref_layout, n_blocks, (const u64 *)base, shift, c->d_bwt.as<uint4>());
	GSA_CHECK(c, hipGetLastError());
	GSA_CHECK(c, hipStreamSynchronize(c->stream));
	c->di.bwt = c->d_bwt.as<uint4>(); c->di.occ_base = base; c->di.occ_shift = shift;
	return GSA_OK;
}

int build_dense_sa(gsa_ctx *c, u64 n_sa)
{
	{
		const u64 words = c->di.seq_len / 16 + 8;      // (the 64-base text window reads five words from any base)
		if (!dev_ensure<u32>(c, c->d_ref2, words, true)) return GSA_ERR_NOMEM;
		hipLaunchKernelGGL(k_pack_ref, dim3(grid_for(words, 256)), dim3(256), 0, c->stream, c->di.ref, c->di.seq_len, c->d_ref2.as<u32>(), words);
		GSA_CHECK(c, hipGetLastError());
		c->di.ref2 = c->d_ref2.as<u32>();
	}
	const u64 rows = c->di.seq_len + 1;
	const bool use32 = c->di.seq_len < 0xFFFFFFF0ull && !c->force_wide;
	if (use32) { if (!dev_ensure<u32>(c, c->d_sa_dense, rows + 32, true)) return GSA_ERR_NOMEM; c->di.sa32 = c->d_sa_dense.as<u32>(); c->di.sa64 = nullptr; }
	else { if (!dev_ensure<u64>(c, c->d_sa_dense, rows + 32, true)) return GSA_ERR_NOMEM; c->di.sa64 = c->d_sa_dense.as<u64>(); c->di.sa32 = nullptr; }
	hipLaunchKernelGGL(k_densify_sa, dim3(grid_for(n_sa, 256)), dim3(256), 0, c->stream, c->di, n_sa, (u32 *)c->di.sa32, (u64 *)c->di.sa64);
	GSA_CHECK(c, hipGetLastError());
	GSA_CHECK(c, hipStreamSynchronize(c->stream));
	{
		// k = ceil(log4(2G)) + 2: nearly all k-mers that occur are unique then (a 10 Mb text: 96 % at k = 14, 86 % at k = 13), so a
		// search is table -> text comparison with no stepwise Occ walk in between -- each Occ step is a round trip AND the
		// heaviest block of the search loop.  Capped at 15 and at a quarter of the free device memory (4^15 x 16 B = 16 GiB
		// of the 288: a human-chromosome-sized text of 5 x 10^8 rows has 34 % unique k-mers at k = 14, 78 % at 15).
		int k = 0; while ((1ull << (2 * k)) < c->di.seq_len) k++;
		k += 2; if (k > 15) k = 15;      // (not beyond the default MinSeedLength: a start whose first 15 bases occur -- presence bitmap -- must find its entry, else it walks base by base)
		{
			size_t fr = 0, tot = 0;
			if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); fr = 8ull << 30; }
			const size_t esz = (c->di.seq_len < 0xFFFFFFF0ull && !c->force_wide) ? 16 : 32;
			while (k > 2 && ((size_t)esz << (2 * k)) > fr / 4) k--;
			if (c->opt.kmer_k) { const int kk = c->opt.kmer_k; if (kk >= 2 && kk <= 15 && ((size_t)esz << (2 * kk)) <= fr / 2) k = kk; }      // (GSA_CREATE_KMER_K; tests: a long table on a short text)
		}
		if (k >= 2) {
			const size_t n = (size_t)1 << (2 * k);
			const int e16 = (c->di.seq_len < 0xFFFFFFF0ull && !c->force_wide) ? 1 : 0;
			{	// (exactly this size: dev_ensure's growth margin would be 16 GiB on the longest table)
				const size_t bytes = (e16 ? n * 2 : n * 4) * sizeof(u64);
				if (c->d_kmer.cap < bytes) {
					if (c->d_kmer.p) { hipFree(c->d_kmer.p); c->d_kmer.p = nullptr; c->d_kmer.cap = 0; }
					size_t got = 0;
					if (void *r = dev_take_reserved(c->device, bytes, &got)) { c->d_kmer.p = r; c->d_kmer.cap = got; }
					else {
					if (hipMalloc(&c->d_kmer.p, bytes) != hipSuccess) { (void)hipGetLastError(); return gsa_fail(c, GSA_ERR_NOMEM, "hipMalloc (k-mer table)"); }
					c->d_kmer.cap = bytes;
					}
				}
			}
			hipLaunchKernelGGL(k_build_kmer, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, c->di, k, c->d_kmer.as<u64>(), e16);
			c->di.kmer_e16 = e16;
			GSA_CHECK(c, hipGetLastError());
			GSA_CHECK(c, hipStreamSynchronize(c->stream));
			c->di.kmer = c->d_kmer.as<u64>(); c->di.kmer_k = k;
		}
	}
	return GSA_OK;
}

// grow a device buffer keeping its first `keep` elements
template <class T> static T *dev_grow_keep(gsa_ctx *c, DevBuf &b, size_t n, size_t keep)
{
	if ((n ? n : 1) * sizeof(T) <= b.cap) return (T *)b.p;
	DevBuf nb;
	if (!dev_ensure<T>(c, nb, n + n / 2)) return nullptr;
	if (keep && b.p) { if (hipMemcpyAsync(nb.p, b.p, keep * sizeof(T), hipMemcpyDeviceToDevice, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { hipFree(nb.p); gsa_fail(c, GSA_ERR_HIP, "hipMemcpyAsync"); return nullptr; } }
	if (b.p) { ctx_quiesce(c); hipFree(b.p); }
	b = nb;
	return (T *)b.p;
}

// Groups without the PosDiff sort: decide whether the bitmap of occupied PosDiff values is kept for this contig and clear it
static int prepare_pd_bitmap(gsa_ctx *c, i64 n_hits, i64 n_chunks = 0)
{
	const u64 pd_words = ((u64)c->pd_span >> 5) + 2;
	// (a chunk range: the hit count of the whole contig is not known here; the bitmap is kept whenever MaxIndelSize allows it)
	// (since the scan only visits occupied blocks -- OpPdScan, round 4 -- the bitmap pays whatever the hit count: a human reference is 776 MB per
	//  contig; beyond 2 GB -- bundles against a large reference -- only with enough hits to justify the memory)
	c->pd_path = (n_hits > 0 || c->split) && c->prm.MaxIndelSize >= 0 && c->prm.MaxIndelSize <= 31 && (c->split || pd_words <= (512ull << 20) || pd_words <= 64ull * (u64)n_hits + 65536) && c->opt.pd_bitmap;
	c->seed_view_ready = false;
	if (c->pd_path) {
		const size_t cap0 = c->d_pdbm.cap;
		const size_t ccap0 = c->d_pdcb.cap;
		if (!dev_ensure<u32>(c, c->d_pdbm, (size_t)pd_words + 66) || !dev_ensure<u32>(c, c->d_pdcb, (size_t)(pd_words >> 10) + 4)) {
			// no room for the bitmap (up to 2 GB per context): this contig's groups come from the PosDiff sort instead (seed_view_sort), as
			// for MaxIndelSize > 31 -- slower, same result
			(void)hipGetLastError(); c->err.clear(); c->pd_path = false; c->pdbm_dirty = true;
			return GSA_OK;
		}
		if (c->d_pdbm.cap != cap0 || c->pdbm_dirty) GSA_CHECK(c, hipMemsetAsync(c->d_pdbm.p, 0, c->d_pdbm.cap, c->stream));
		if (c->d_pdcb.cap != ccap0 || c->pdbm_dirty) GSA_CHECK(c, hipMemsetAsync(c->d_pdcb.p, 0, c->d_pdcb.cap, c->stream));
		c->pdbm_dirty = true; c->pd_words = (i64)pd_words;
		// the byte map (Options::pd_bytes): where a chunk's hits overflow the workgroup's table of words (256) and the pass over the bytes costs less than their atomics would
		c->pd_bytes = false;
		if (!c->split && n_chunks > 0 && (c->opt.pd_bytes == 2 || (c->opt.pd_bytes == 1 && n_hits >= 512 * n_chunks && pd_words * 32 <= 256ull * (u64)n_hits && pd_words <= (128ull << 20)))) {      // (at most 4 GB of bytes per context)
			const size_t bcap0 = c->d_pdby.cap;
			if (dev_ensure<uint8_t>(c, c->d_pdby, ((size_t)pd_words + 66) * 32)) {
				if (c->d_pdby.cap != bcap0) GSA_CHECK(c, hipMemsetAsync(c->d_pdby.p, 0, c->d_pdby.cap, c->stream));
				c->pd_bytes = true;
			} else { (void)hipGetLastError(); c->err.clear(); }      // (no room: the atomics do it)
		}
	}
	return GSA_OK;
}

// Stage 1 for the whole contig, or -- c->split -- for the chunk range [rng_beg, rng_end) of it: chunks are searched
// independently (GSAlign.cpp:61-94: a thread takes 10 000-bp chunks off a counter; seeds never cross a chunk edge), so a
// long contig can be seeded by several GPUs and the hits sent to the GPU that chains it.  The
// kernels see the range as a contig of its own (pointer + length); only the select kernel needs the absolute position.
// At most GSA_SEED_SLOTS contexts of one GPU inside their seed-search kernels at a time (0 = no limit).  Contexts that are
// handed contigs of one size start together and stay in step -- four speculative kernels compete for the one resource that bounds
// them (random 32-byte sectors of HBM), then four chaining stages, which are chains of short dependent passes, leave the chip
// idle together (kernel timeline: profiles/archive/r03_timeline_multi_human.txt).  With a gate in front of the seed kernels the contexts
// fall out of step: one searches while the others chain and extend.
static std::mutex g_seed_mu; static std::condition_variable g_seed_cv; static int g_seed_busy[64];
struct SeedGate {
	int dev, slots; bool held;
	SeedGate(int d, int n) : dev(d & 63), slots(n), held(false) { if (slots > 0) { std::unique_lock<std::mutex> lk(g_seed_mu); g_seed_cv.wait(lk, [&] { return g_seed_busy[dev] < slots; }); g_seed_busy[dev]++; held = true; } }
	void release() { if (held) { { std::lock_guard<std::mutex> lk(g_seed_mu); g_seed_busy[dev]--; } g_seed_cv.notify_all(); held = false; } }
	~SeedGate() { release(); }
};

int stage1_seed(gsa_ctx *c)
{
	// GSA_SEED_CUS=n (experiment): the seed-search kernels run on a stream that may only use n of the CUs, spread evenly
#ifdef GSA_EXPERIMENTS
	static const int seed_cus = [] { const char *e = getenv("GSA_SEED_CUS"); return e ? atoi(e) : 0; }();
#else
	const int seed_cus = 0;
#endif
	if (seed_cus > 0 && !c->stream_seed) {
		hipDeviceProp_t pr; GSA_CHECK(c, hipGetDeviceProperties(&pr, c->device));
		const int ncu = pr.multiProcessorCount;
		std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
		for (int i = 0; i < ncu; i++) if ((long long)(i + 1) * seed_cus / ncu != (long long)i * seed_cus / ncu) mask[(size_t)i >> 5] |= 1u << (i & 31);
		GSA_CHECK(c, hipExtStreamCreateWithCUMask(&c->stream_seed, (uint32_t)mask.size(), mask.data()));
		GSA_CHECK(c, hipEventCreateWithFlags(&c->ev_seed_fork, hipEventDisableTiming));
	}
	hipStream_t st = c->stream;
	if (c->stream_seed) {      // (everything in front of the search on the main stream -- the contig's upload -- first; the host waits for the search itself)
		GSA_CHECK(c, hipEventRecord(c->ev_seed_fork, c->stream)); GSA_CHECK(c, hipStreamWaitEvent(c->stream_seed, c->ev_seed_fork, 0));
		st = c->stream_seed;
	}
	const bool split = c->split;
	const i64 all_chunks = ((i64)c->qlen + GSA_CHUNK - 1) / GSA_CHUNK;
	const i64 cb = split ? c->rng_beg : 0, ce = split ? (c->rng_end < all_chunks ? c->rng_end : all_chunks) : all_chunks;
	const i32 s_off = (i32)(cb * GSA_CHUNK);
	const i32 qlen_full = c->qlen;
	const i32 qlen = ce > cb ? (i32)(((i64)ce * GSA_CHUNK < (i64)qlen_full ? (i64)ce * GSA_CHUNK : (i64)qlen_full) - s_off) : 0;
	const uint8_t *d_q = c->q_dev + s_off;
	c->n_seeds = 0; c->n_groups = 0;
	if (qlen <= 0) return split ? prepare_pd_bitmap(c, 0) : GSA_OK;
	const i64 n_chunks = ((i64)qlen + GSA_CHUNK - 1) / GSA_CHUNK;
	size_t ccap = c->cand_cap_per_chunk;                // candidate slots per chunk (grows on overflow)
	if (!dev_ensure<u32>(c, c->d_onpath, (size_t)n_chunks * PATH_WORDS) || !dev_ensure<i32>(c, c->d_chunk_hits, (size_t)n_chunks + 1) || !dev_ensure<i32>(c, c->d_chunk_base, (size_t)n_chunks + 1)) return GSA_ERR_NOMEM;
	i64 n_hits = 0;
	u64 *cnt = c->d_cnt.as<u64>();
	// Dense mode (one search per start position, k_dense_search): every chunk under -sen, otherwise only the chunks the
	// speculative kernel gives up on.  The accounting build walks everything the reference's way and takes neither path.
	// GSA_SEED_MODE: "spec" (default) = the speculative kernel, and the right-to-left sweep (k_dense_sweep) for the chunks it gives up
	// on and for every chunk under -sen; "sweep" = every chunk through the sweep; "search" = round 2's dense kernel (one search per
	// start) in place of the sweep
	const int seed_mode = c->opt.seed_mode;
	// Which kernel for what (measured, profiles/archive/r03_seed_modes.txt): the speculative kernel wins where matches are unique (the bench
	// workload: 4.4 ms per 250 Mb against 19 for the sweep over every chunk); the sweep wins where repeats with thousands of copies make
	// most chunks exceed the speculative budget (adversarial 250 Mb: 31 ms against 62 for round 2's path); one search per start wins
	// under -sen (4.2 against 16 ms per 12 Mb: matches are short, a lane's chain of 40 starts is the longer road) and for a handful
	// of handed-over chunks.  A contig whose predecessor handed more than 40 % of its chunks over skips the speculative attempt.
	const bool sweep_all = !c->prm.bSensitive && !c->count_blocks && (seed_mode == 0 || (seed_mode == 1 && c->seed_sweep_next));
	const bool dense_all = (c->prm.bSensitive || sweep_all) && !c->count_blocks;
	const u32 budget = c->count_blocks ? 0u : c->seed_budget;
	if (dense_all && ccap < GSA_CHUNK / 5 + 64) { ccap = GSA_CHUNK / 5 + 64; c->cand_cap_per_chunk = ccap; }      // one accepted start in five at most
	u64 occ_all = 0;
#ifdef GSA_EXPERIMENTS
	static const int seed_slots = [] { const char *e = getenv("GSA_SEED_SLOTS"); return e ? atoi(e) : 0; }();
#else
	const int seed_slots = 0;
#endif
	SeedGate gate(c->device, (c->profiling || c->count_blocks) ? 0 : seed_slots);
	u64 contig_maxcand = 0;      // most candidates in one chunk of this contig
	for (int attempt = 0;; attempt++) {
		if (attempt == 8) return gsa_fail(c, GSA_ERR_LIMIT, "seed buffers keep overflowing");
		const size_t ctot = ccap * (size_t)n_chunks;
		if (!dev_ensure<i32>(c, c->d_cand_s, ctot) || !dev_ensure<i32>(c, c->d_cand_len, ctot) || !dev_ensure<u64>(c, c->d_cand_x0, ctot) || !dev_ensure<i32>(c, c->d_cand_freq, ctot) || !dev_ensure<u32>(c, c->d_cand_cnt, (size_t)n_chunks)) return GSA_ERR_NOMEM;
		if (!dense_all && !dev_ensure<u32>(c, c->d_heavy, (size_t)n_chunks)) return GSA_ERR_NOMEM;
		// (the counters were left at zero by the previous contig's last workgroup; chunk_hits[n_chunks] = 0 is written by chunk 0)
		if (c->profiling || c->prof_seed) hipEventRecord(c->ev[0], st);
		u64 hits = 0, maxcand = 0, n_heavy = dense_all ? (u64)n_chunks : 0;
		occ_all = 0;
		if (!dense_all) {
#define GSA_SEED_ARGS c->di, d_q, qlen, c->prm, cnt, c->d_cand_s.as<i32>(), c->d_cand_len.as<i32>(), c->d_cand_x0.as<u64>(), c->d_cand_freq.as<i32>(), (u32)ccap, \
			c->d_cand_cnt.as<u32>(), c->d_onpath.as<u32>(), c->d_chunk_hits.as<i32>(), c->h_cnt, budget, c->d_heavy.as<u32>(), c->d_chunk_base.as<i32>()
			// (persistent launch: ONE workgroup of SEED_WPW independent waves per CU -- its LDS is more than half a CU's, so the dispatcher cannot stack two -- every wave
			//  draws chunk after chunk from the ticket counter.  A short contig: as many workgroups as its chunks fill.  Until round 5: twelve one-wave workgroups per CU;
			//  the dispatcher fills a CU before it moves on, so a shorter grid of those meant FEWER CUs, not thinner ones: 3.6 / 4.8 / 7.1 ms at 8 / 5 / 3 per CU against 3.2 at 12)
			const i64 n_units = c->count_blocks ? n_chunks : (n_chunks + SEED_NCH - 1) / SEED_NCH;      // (a wave of the production kernel owns SEED_NCH chunks)
			const int wpw = c->count_blocks ? 1 : SEED_WPW;
			unsigned grid = (unsigned)((n_units + wpw - 1) / wpw);
			if (!c->count_blocks) {
				if (c->n_cus <= 0) { hipDeviceProp_t pr; GSA_CHECK(c, hipGetDeviceProperties(&pr, c->device)); c->n_cus = pr.multiProcessorCount; }
				const i64 cap = (i64)c->n_cus * (SEED_WPW > 1 ? SEED_WGS_PER_CU : SEED_PERSIST);      // (SEED_WPW = 1: round 5's twelve one-wave workgroups per CU, kept for A/B builds)
				if (cap < (i64)grid) grid = (unsigned)cap;
			}
			const u64 tk_base = c->seed_ticket; c->seed_ticket += (u64)n_units + (u64)grid * (u64)wpw;      // (every wave's last draw is the one that fails)
			const size_t dl_count = sizeof(SeedLds<true, 1>);
			size_t dl = sizeof(SeedLds<false, SEED_NCH>) * SEED_WPW; if (SEED_WPW > 1 && dl < SEED_WG_LDS_MIN) dl = SEED_WG_LDS_MIN;      // (more than half a CU's LDS: two fat workgroups never share a CU)
			if (c->count_blocks) hipLaunchKernelGGL((k_seed_wg<true, false>), dim3(grid), dim3(SEED_WG), dl_count, st, GSA_SEED_ARGS, tk_base, (u32)n_chunks);
			else if (c->di.kmer_e16) hipLaunchKernelGGL((k_seed_wg<false, true>), dim3(grid), dim3(SEED_WG * SEED_WPW), dl, st, GSA_SEED_ARGS, tk_base, (u32)n_chunks);
			else hipLaunchKernelGGL((k_seed_wg<false, false>), dim3(grid), dim3(SEED_WG * SEED_WPW), dl, st, GSA_SEED_ARGS, tk_base, (u32)n_chunks);
#undef GSA_SEED_ARGS
			if (c->profiling || c->prof_seed) hipEventRecord(c->ev[1], st);
			// (the counters are in pinned memory when the seed kernel is done; the host waits for that, not for the scan of the
			//  per-chunk hit counts behind it)
			GSA_CHECK(c, hipEventRecord(c->ev[21], st));      // (the exclusive prefix of the per-chunk hit counts is left by the kernel's last workgroup)
			GSA_CHECK(c, hipEventSynchronize(c->ev[21]));
			hits = c->h_cnt[CNT_HITS]; maxcand = c->h_cnt[CNT_CAND]; n_heavy = c->h_cnt[CNT_HEAVY]; occ_all = c->h_cnt[CNT_OCCBLK_ALL];
			c->dbg[0] = c->h_cnt[11]; c->dbg[1] = n_heavy; c->dbg[2] = c->h_cnt[13]; c->dbg[3] = c->h_cnt[14]; c->dbg[4] = c->h_cnt[15]; c->dbg[5] = c->h_cnt[7];
			c->counters[0] = c->h_cnt[CNT_OCCBLK];
			if (seed_mode == 1) {
				// re-decided by every contig that goes through the speculative kernel.  A look that only confirms the sweep doubles the distance to
				// the next one (8, 16, 32, 64 contigs: the speculative attempt costs a repeat-rich 250 Mb contig 4.9 ms on top of its 12)
				c->seed_sweep_next = n_heavy * 5 > (u64)n_chunks * 2;
				c->seed_sweep_period = (c->seed_sweep_next && c->seed_sweep_probe) ? (c->seed_sweep_period < 64 ? c->seed_sweep_period * 2 : 64) : 8;
				c->seed_sweep_probe = false;
			}
		}
		if (n_heavy > 0) {
			const size_t nd = (size_t)n_heavy * GSA_CHUNK;
			if (!dev_ensure<u32>(c, c->dn_lf, nd) || !dev_ensure<u64>(c, c->dn_x0, nd)) return GSA_ERR_NOMEM;
			const u32 *list = dense_all ? (const u32 *)nullptr : c->d_heavy.as<u32>();
#define GSA_DENSE_ARGS(SPAN) dim3((unsigned)(n_heavy * DENSE_WGS(SPAN))), dim3(DENSE_TPB), 0, st, c->di, d_q, qlen, c->prm, list, c->dn_lf.as<u32>(), c->dn_x0.as<u64>(), cnt
#ifdef GSA_EXPERIMENTS
			static const u64 sweep_min = [] { const char *e = getenv("GSA_SWEEP_MIN"); return e ? (u64)atoll(e) : 1024ull; }();
#else
			const u64 sweep_min = 1024;
#endif
			const bool use_sweep = seed_mode != 2 && ((seed_mode == 0 && dense_all) || n_heavy >= sweep_min);      // (-sen: every chunk is dense -- a bundle's worth of them is swept, a short contig's few are searched start by start: one round trip chain of ~5 per start beats a segment's chain of 60-250 when the chip is empty)
			if (sweep_all && seed_mode == 1 && ++c->seed_sweep_run >= c->seed_sweep_period) { c->seed_sweep_next = false; c->seed_sweep_probe = true; c->seed_sweep_run = 0; }      // look again now and then
			if (use_sweep) {
#ifdef GSA_EXPERIMENTS
				static const int seg_env = [] { const char *e = getenv("GSA_SWEEP_SEG"); return e ? atoi(e) : 0; }();
#else
				const int seg_env = 0;
#endif
				const int shape_env = c->opt.sweep_shape;
				// few dense chunks (a bundle of short contigs): one chunk per workgroup of four waves and 40 starts per segment, so that the
				// chip has waves to run; many: four chunks per workgroup of two waves, 160 starts per segment
				const bool small = shape_env >= 0 ? shape_env == 1 : n_heavy < 4096;      // (a 60 Mb -sen bundle, 6 000 chunks: 3.9 ms with four chunks per workgroup, 4.8 with one)
				const int seg = seg_env > 0 ? seg_env : (small ? 40 : 160);
#define GSA_SWEEP_ARGS(NCH_, TPB_) dim3((unsigned)((n_heavy + (NCH_) - 1) / (NCH_))), dim3(TPB_), 0, st, c->di, d_q, qlen, c->prm, list, (u32)n_heavy, c->dn_lf.as<u32>(), c->dn_x0.as<u64>(), cnt, seg
				if (small) { if (c->di.kmer_e16) hipLaunchKernelGGL((k_dense_sweep<true, 1, 256>), GSA_SWEEP_ARGS(1, 256)); else hipLaunchKernelGGL((k_dense_sweep<false, 1, 256>), GSA_SWEEP_ARGS(1, 256)); }
				else { if (c->di.kmer_e16) hipLaunchKernelGGL((k_dense_sweep<true, SWEEP_NCH, SWEEP_TPB>), GSA_SWEEP_ARGS(SWEEP_NCH, SWEEP_TPB)); else hipLaunchKernelGGL((k_dense_sweep<false, SWEEP_NCH, SWEEP_TPB>), GSA_SWEEP_ARGS(SWEEP_NCH, SWEEP_TPB)); }
#undef GSA_SWEEP_ARGS
			}
			else if (dense_all) { if (c->di.kmer_e16) hipLaunchKernelGGL((k_dense_search<true, 512>), GSA_DENSE_ARGS(512)); else hipLaunchKernelGGL((k_dense_search<false, 512>), GSA_DENSE_ARGS(512)); }
			else { if (c->di.kmer_e16) hipLaunchKernelGGL((k_dense_search<true, 256>), GSA_DENSE_ARGS(256)); else hipLaunchKernelGGL((k_dense_search<false, 256>), GSA_DENSE_ARGS(256)); }
#undef GSA_DENSE_ARGS
			hipLaunchKernelGGL(k_dense_resolve, dim3((unsigned)n_heavy), dim3(256), 0, st, list, (u32)n_chunks, qlen, (int)c->prm.bSensitive, c->dn_lf.as<u32>(), c->dn_x0.as<u64>(), cnt,
			                   c->d_cand_s.as<i32>(), c->d_cand_len.as<i32>(), c->d_cand_x0.as<u64>(), c->d_cand_freq.as<i32>(), (u32)ccap, c->d_cand_cnt.as<u32>(), c->d_onpath.as<u32>(),
			                   c->d_chunk_hits.as<i32>(), c->h_cnt, c->d_chunk_base.as<i32>());
			GSA_CHECK(c, hipGetLastError());
			if (c->profiling || c->prof_seed) hipEventRecord(c->ev[1], st);
			GSA_CHECK(c, hipEventRecord(c->ev[21], st));      // (the exclusive prefix of the per-chunk hit counts is left by the kernel's last workgroup)
			GSA_CHECK(c, hipEventSynchronize(c->ev[21]));
			hits += c->h_cnt[CNT_HITS]; if (c->h_cnt[CNT_CAND] > maxcand) maxcand = c->h_cnt[CNT_CAND]; occ_all += c->h_cnt[CNT_OCCBLK_ALL];
			if (dense_all) { c->dbg[0] = 0; c->dbg[1] = n_heavy; c->dbg[2] = c->dbg[3] = c->dbg[4] = c->dbg[5] = 0; c->counters[0] = 0; }
		}
		if (hits >= (1ull << 31) - 2) return gsa_fail(c, GSA_ERR_LIMIT, "more than 2^31 seeds in one contig");
		if (maxcand > ccap) { ccap = (size_t)maxcand + 256; c->cand_cap_per_chunk = ccap; continue; }
		n_hits = (i64)hits; contig_maxcand = maxcand;
		break;
	}
	gate.release();
	st = c->stream;      // (the host has waited for the search kernels: what follows is ordered behind them)
	const size_t hcap = (size_t)n_hits + 64;
	// Groups: a new group starts where the sorted PosDiff values jump by more than MaxIndelSize.  With a bitmap of the
	// occupied PosDiff values that needs no sort: group id = number of group starts at or below a hit's PosDiff (a scan
	// over the bitmap, stage 2).  The PosDiff-sorted view of the seeds (stage-1 view of the C ABI) is then built on demand.
	if (int rcp = prepare_pd_bitmap(c, n_hits, n_chunks)) return rcp;
	if (n_hits > 0) {
		if (!dev_ensure<u64>(c, c->d_key_a, hcap) || !dev_ensure<u32>(c, c->d_val_a, hcap)) return GSA_ERR_NOMEM;
		// (LDS by the contig's own maximum: the segments' capacity only grows -- one contig with a crowded chunk, or the counting pass of the
		//  bench, and every later launch would run one workgroup per CU)
		const size_t sel_cand = contig_maxcand < ccap ? (((size_t)contig_maxcand + 64) & ~(size_t)63) : ccap;
		hipLaunchKernelGGL(k_seed_select, dim3((unsigned)n_chunks), dim3(256), 2 * (sel_cand + 2) * sizeof(u32), st, c->di, (u32)ccap, c->d_cand_cnt.as<u32>(), c->d_cand_s.as<i32>(), c->d_cand_len.as<i32>(),
		                   c->d_cand_x0.as<u64>(), c->d_cand_freq.as<i32>(), c->d_onpath.as<u32>(), c->d_chunk_base.as<i32>(), c->bnd, s_off, c->qbits, c->d_key_a.as<u64>(), c->d_val_a.as<u32>(), c->pd_path ? c->d_pdbm.as<u32>() : (u32 *)nullptr, c->d_pdcb.as<u32>(), (u32)sel_cand, (c->pd_path && c->pd_bytes) ? c->d_pdby.as<uint8_t>() : (uint8_t *)nullptr);
		if (c->pd_path && c->pd_bytes) {
			const i64 tiles = (c->pd_words + 1023) >> 10;
			hipLaunchKernelGGL(k_pd_pack, dim3((unsigned)(tiles < 4096 ? tiles : 4096)), dim3(256), 0, st, c->d_pdby.as<uint8_t>(), c->pd_words, c->d_pdbm.as<u32>(), c->d_pdcb.as<u32>());
		}
	}
	if (c->profiling) hipEventRecord(c->ev[2], st);
	u64 lf_steps = 0;
	if (c->count_blocks && n_hits > 0) {
		unsigned long long *d_lf = (unsigned long long *)(c->d_mail.as<i32>() + M_LFSTEPS);
		GSA_CHECK(c, hipMemsetAsync(d_lf, 0, 8, st));
		hipLaunchKernelGGL(k_count_lf, dim3((unsigned)n_chunks), dim3(256), 0, st, c->di, (u32)ccap, c->d_cand_cnt.as<u32>(), c->d_cand_s.as<i32>(), c->d_cand_x0.as<u64>(),
		                   c->d_cand_freq.as<i32>(), c->d_onpath.as<u32>(), d_lf);
		GSA_CHECK(c, hipMemcpyAsync(&c->h_cnt[CNT_DONE], d_lf, 8, hipMemcpyDeviceToHost, st));      // (h_cnt[CNT_DONE] is always 0 after the seed kernel: a free pinned slot)
		GSA_CHECK(c, hipStreamSynchronize(st));
		lf_steps = c->h_cnt[CNT_DONE];
	}
	c->counters[1] = lf_steps; c->counters[2] = (u64)n_hits; c->counters[3] = (u64)n_hits; c->counters[7] = occ_all;
	c->n_seeds = n_hits; c->hits_sorted = !split;
	if (split) return GSA_OK;                 // (the tail of stage 1 runs in gsa_finish_contig, on the hits of all ranges)
	if (n_hits == 0) { if (c->profiling) { GSA_CHECK(c, hipStreamSynchronize(st)); float ms; hipEventElapsedTime(&ms, c->ev[0], c->ev[1]); c->kernel_ms[0] = ms; } return GSA_OK; }
	if (n_hits >= (1ll << 31) - 2) return gsa_fail(c, GSA_ERR_LIMIT, "more than 2^31 seeds in one contig");
	if (c->pd_path) {
		if (c->profiling) hipEventRecord(c->ev[3], st);
		c->n_groups = -1; c->ev_pending |= 1;
		return GSA_OK;
	}
	return seed_view_sort(c);
}

// Seeds in PosDiff order with their group ids (CompByPosDiff + SeedGrouping, a5/a6): always for the stage-1 view of
// the C ABI, and as the front of stage 2 when the PosDiff bitmap does not apply.
int seed_view_sort(gsa_ctx *c)
{
	if (c->seed_view_ready || c->n_seeds == 0) return GSA_OK;
	hipStream_t st = c->stream;
	const size_t n = (size_t)c->n_seeds, hcap = n + 64;
	if (!dev_ensure<u64>(c, c->d_key_b, hcap) || !dev_ensure<u32>(c, c->d_val_b, hcap)) return GSA_ERR_NOMEM;
	int rc = gsa_sort_pairs_u64_u32(c, c->d_key_a.as<u64>(), c->d_key_b.as<u64>(), c->d_val_a.as<u32>(), c->d_val_b.as<u32>(), n, 0, c->qbits + c->pdbits);
	if (rc) return rc;
	if (!dev_ensure<i32>(c, c->s_q, n) || !dev_ensure<i32>(c, c->s_len, n) || !dev_ensure<i64>(c, c->s_r, n) || !dev_ensure<i32>(c, c->s_gid, n) ||
	    !dev_ensure<i32>(c, c->d_flag, n + 1) || !dev_ensure<i32>(c, c->d_scan, n + 1) || !dev_ensure<i32>(c, c->g_beg, n + 1)) return GSA_ERR_NOMEM;
	{
		OpDecodeGroup op = { (i64)n, c->d_key_b.as<u64>(), c->d_val_b.as<u32>(), c->bnd, c->qbits, c->prm.MaxIndelSize,
		                     c->s_q.as<i32>(), c->s_len.as<i32>(), c->s_r.as<i64>(), c->s_gid.as<i32>(), c->g_beg.as<i32>(), c->d_mail.as<i32>() };
		rc = lb_launch<1>(c, (i64)n, op);
		if (rc) return rc;
	}
	if (c->profiling && !c->pd_path) hipEventRecord(c->ev[3], st);
	// the group count stays on the device (mailbox); nothing downstream needs it on the host
	c->n_groups = -1;
	if (!c->pd_path) c->ev_pending |= 1;
	c->seed_view_ready = true;
	return GSA_OK;
}

// hits of another GPU's chunk range behind this context's own ones (keys / vals: host or device memory)
int stage1_import_hits(gsa_ctx *c, const u64 *keys, const u32 *vals, i64 n)
{
	if (n <= 0) return GSA_OK;
	if (c->n_seeds + n >= (1ll << 31) - 2) return gsa_fail(c, GSA_ERR_LIMIT, "more than 2^31 seeds in one contig");
	const size_t have = (size_t)c->n_seeds, want = have + (size_t)n + 64;
	if (!dev_grow_keep<u64>(c, c->d_key_a, want, have) || !dev_grow_keep<u32>(c, c->d_val_a, want, have)) return GSA_ERR_NOMEM;
	// (host memory, memory of this GPU, or of another GPU of the node -- then the copy goes peer to peer over xGMI)
	int src_dev = -1;
	{ hipPointerAttribute_t at; if (hipPointerGetAttributes(&at, keys) == hipSuccess && at.type == hipMemoryTypeDevice) src_dev = at.device; else (void)hipGetLastError(); }
	if (src_dev >= 0 && src_dev != c->device) {
		int can = 0; (void)hipDeviceCanAccessPeer(&can, c->device, src_dev);
		if (can) { hipError_t e = hipDeviceEnablePeerAccess(src_dev, 0); if (e != hipSuccess) (void)hipGetLastError(); }      // (already enabled is fine)
		GSA_CHECK(c, hipMemcpyPeerAsync(c->d_key_a.as<u64>() + have, c->device, keys, src_dev, (size_t)n * 8, c->stream));
		GSA_CHECK(c, hipMemcpyPeerAsync(c->d_val_a.as<u32>() + have, c->device, vals, src_dev, (size_t)n * 4, c->stream));
	} else {
		GSA_CHECK(c, hipMemcpyAsync(c->d_key_a.as<u64>() + have, keys, (size_t)n * 8, hipMemcpyDefault, c->stream));
		GSA_CHECK(c, hipMemcpyAsync(c->d_val_a.as<u32>() + have, vals, (size_t)n * 4, hipMemcpyDefault, c->stream));
	}
	if (c->pd_path) hipLaunchKernelGGL(k_pd_from_keys, dim3(grid_for((size_t)n, 256)), dim3(256), 0, c->stream, n, c->d_key_a.as<u64>() + have, c->qbits, c->d_pdbm.as<u32>(), c->d_pdcb.as<u32>());
	GSA_CHECK(c, hipGetLastError());
	GSA_CHECK(c, hipStreamSynchronize(c->stream));      // (the caller's buffers are free again)
	c->n_seeds += n;
	return GSA_OK;
}

// the tail of stage 1 once the hits of every chunk range are here
int stage1_finish_split(gsa_ctx *c)
{
	c->counters[2] = c->counters[3] = (u64)c->n_seeds;
	c->seed_view_ready = false;
	if (c->n_seeds == 0) return GSA_OK;
	if (c->pd_path) { c->n_groups = -1; return GSA_OK; }
	return seed_view_sort(c);
}

// Stage 2 consumed the PosDiff bitmap (k_pd_gather clears the words it read); a second stage 2 on the same hits needs it back.
int stage1_restore_pdbm(gsa_ctx *c)
{
	if (!c->pd_path || c->n_seeds == 0) return GSA_OK;
	if (c->pdbm_dirty) { GSA_CHECK(c, hipMemsetAsync(c->d_pdbm.p, 0, c->d_pdbm.cap, c->stream)); GSA_CHECK(c, hipMemsetAsync(c->d_pdcb.p, 0, c->d_pdcb.cap, c->stream)); }
	hipLaunchKernelGGL(k_pd_from_keys, dim3(grid_for((size_t)c->n_seeds, 256)), dim3(256), 0, c->stream, c->n_seeds, c->d_key_a.as<u64>(), c->qbits, c->d_pdbm.as<u32>(), c->d_pdcb.as<u32>());
	GSA_CHECK(c, hipGetLastError());
	c->pdbm_dirty = true;
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
		// even hits: the reference's LF walk on the sampled SA; odd hits: the dense SA (both must agree with bwt_sa)
		for (int h = 0; h < f; h++) out_loc[(i64)i * GSA_MAX_SEED_FREQ + h] = (h & 1) ? (i64)fm_locate(di, ik.x0 + h) : (i64)fm_locate_walk(di, ik.x0 + h, steps);
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
	i32 *d_start = dev_ensure<i32>(c, c->leaf[0], (size_t)n), *d_stop = dev_ensure<i32>(c, c->leaf[1], (size_t)n), *d_len = dev_ensure<i32>(c, c->leaf[2], (size_t)n), *d_freq = dev_ensure<i32>(c, c->leaf[3], (size_t)n);
	i64 *d_loc = dev_ensure<i64>(c, c->leaf[4], (size_t)n * GSA_MAX_SEED_FREQ);
	if (!d_start || !d_stop || !d_len || !d_freq || !d_loc) return GSA_ERR_NOMEM;
	GSA_CHECK(c, hipMemcpyAsync(d_start, start, n * 4, hipMemcpyHostToDevice, st));
	GSA_CHECK(c, hipMemcpyAsync(d_stop, stop, n * 4, hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(k_search_batch, dim3(grid_for(n, 64)), dim3(64), 0, st, c->di, c->q_dev, c->prm, n, d_start, d_stop, d_len, d_freq, d_loc);
	GSA_CHECK(c, hipMemcpyAsync(out_len, d_len, n * 4, hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipMemcpyAsync(out_freq, d_freq, n * 4, hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipMemcpyAsync(out_loc, d_loc, (size_t)n * GSA_MAX_SEED_FREQ * 8, hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipStreamSynchronize(st));
	return GSA_OK;
}
