// gsalign_amd/csrc/k_refine.hip -- stages 3-5, device part: RemoveOverlaps (S3),
// large-gap cuts with the gap-similarity test (S4), reference-chromosome cuts
// (S5), and the leaf table the host list logic works on (a8).
//
// Replaces RemoveOverlaps / RemoveBadSeeds, CheckGapsBetweenSeeds,
// CheckAlnBlockSpanMultipleRefChrs, CalAlnBlockScore
// (reference src/ProcessCandidateAlignment.cpp:26-36,63-70,81-156,189-239).
//
// RemoveOverlaps looks sequential but a pass only ever modifies the LEFT seed of
// a pair using the (immutable) start coordinates of the right one, so all pairs
// of a pass are independent: one lane per seed, repeat while any seed died.
// A pass that kills nothing is idempotent, so running the passes globally over
// all blocks gives every block the same fixed point as the per-block loop.
#include "gsa_ctx.h"
#include "gsa_scan.h"
#include "gsa_gap.h"

#define TPB 256
#define EARLY_CHUNK 4096     // large gaps copied together with the leaf table (more -> a second copy)
#define GID(n) i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; if (i >= (n)) return
#define LAUNCH(k, n, ...) hipLaunchKernelGGL(k, dim3(grid_for((size_t)(n), TPB)), dim3(TPB), 0, st, __VA_ARGS__)
#define ENS(T, buf, n) do { if (!dev_ensure<T>(c, c->buf, (size_t)(n))) return GSA_ERR_NOMEM; } while (0)
#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// (each struct below is one fused pass: value -> exclusive scan -> emit, see gsa_scan.h; `ub`, the
//  host-known upper bound of every count, is the seed count -- the live counts sit in the mailbox)
struct OpTakeInBlock {      // seeds that belong to a kept S2 block
	const i32 *c_q, *c_len; const i64 *c_r; const i32 *c_bid;
	i32 *r_q, *r_len; i64 *r_r; i32 *r_bid, *r_orig, *mail;      // r_orig: index in the stage-2 seed arrays (links stage-2's early DP launches to the final gaps)
	static constexpr bool clamped = true;      // (gsa_scan.h: loads of every element unconditional; what lies beyond the live count is stale but allocated)
	struct Item { i32 q, len, bid; i64 r; };      // (bid < 0: not taken)
	__device__ Item load(i64 i) const { Item it; it.bid = c_bid[i]; it.q = c_q[i]; it.len = c_len[i]; it.r = c_r[i]; return it; }
	__device__ void prep(Item &it, i64 i) const { if (i >= mail[M_NC]) it.bid = -1; }
	__device__ i32 value(const Item &it, i64, int) const { return it.bid >= 0 ? 1 : 0; }
	__device__ void emit(const Item &it, i64 i, const i32 *v, const i32 *ex) const
	{
		if (!v[0]) return;
		const i32 p = ex[0];
		r_q[p] = it.q; r_len[p] = it.len; r_r[p] = it.r; r_bid[p] = it.bid; r_orig[p] = (i32)i;
	}
	__device__ void done(const i32 *t) const { mail[M_NR] = t[0]; for (int k = 0; k < 32; k++) mail[M_ANY + k] = 0; }      // (+ the "a seed died" flags of the overlap rounds: no fill operation in front of them)
};

// one RemoveOverlaps pass (ProcessCandidateAlignment.cpp:197-226) + its compaction; the seed count
// is read from mail[nin] and written to mail[nout] (two slots: other tiles still read the old one)
struct OpOverlapPass {
	const i32 *q, *len; const i64 *r; const i32 *bid, *orig;
	i32 *oq, *olen; i64 *orr; i32 *obid, *oorig, *mail; int nin, nout, anyslot;
	static constexpr bool clamped = true;
	i64 ub;      // (elements of the launch = length of the arrays; set behind the aggregate)
	struct Item { i32 in, keep, l, q, bid, orig; i64 r;      // in: i < the live count; keep / l: the pass's verdict and the trimmed length
	              i32 qj, bj; i64 rj; };                        // raw: the next seed
	__device__ Item load(i64 i) const
	{
		const i64 ip = i + 1 < ub ? i + 1 : i;
		Item it; it.in = 0; it.keep = 0;
		it.l = len[i]; it.r = r[i]; it.q = q[i]; it.bid = bid[i]; it.orig = orig[i];
		it.bj = bid[ip]; it.rj = r[ip]; it.qj = q[ip];
		return it;
	}
	__device__ void prep(Item &it, i64 i) const
	{
		const i64 n = mail[nin];
		if (i >= n) return;
		it.in = 1; it.keep = 1;
		i32 l = it.l;
		if (i + 1 < n && it.bj == it.bid) {
			if (it.rj <= it.r) it.keep = 0;
			else {
				i32 ov = (i32)(it.r + l - it.rj);
				if (ov > 0) { l -= ov; if (l <= 0) it.keep = 0; }
				if (it.keep) { ov = it.q + l - it.qj; if (ov > 0) { l -= ov; if (l <= 0) it.keep = 0; } }
			}
		}
		it.l = l;
	}
	__device__ i32 value(const Item &it, i64, int) const { return it.in ? it.keep : 0; }
	__device__ void emit(const Item &it, i64, const i32 *v, const i32 *ex) const
	{
		if (!it.in) return;
		if (!v[0]) { mail[anyslot] = 1; return; }
		const i32 p = ex[0];
		oq[p] = it.q; olen[p] = it.l; orr[p] = it.r; obid[p] = it.bid; oorig[p] = it.orig;
	}
	__device__ void done(const i32 *t) const { mail[nout] = t[0]; }
};

// S4 (CheckGapsBetweenSeeds, :120-156): cut4[i] = 1 cut before i; gaps between the two limits become
// CalGapSimilarity jobs
struct OpGapCuts {
	int nin; const i32 *q, *len; const i64 *r; const i32 *bid;
	i32 *cut4, *jq1, *jq2; i64 *jr1, *jr2; i32 *jseed, *mail;
	static constexpr bool clamped = true;
	struct Item { i32 cut, jb, q1, q2; i64 r1, r2; i32 b0, b1, l0; };
	__device__ Item load(i64 i) const
	{
		const i64 im = i > 0 ? i - 1 : 0;
		Item it; it.cut = 0; it.jb = 0;
		it.b0 = bid[im]; it.b1 = bid[i]; it.l0 = len[im]; it.q1 = q[im]; it.q2 = q[i]; it.r1 = r[im]; it.r2 = r[i];
		return it;
	}
	__device__ void prep(Item &it, i64 i) const
	{
		if (i < mail[nin] && i > 0 && it.b0 == it.b1) {
			it.q1 += it.l0; it.r1 += it.l0;
			const i32 qGap = it.q2 - it.q1;
			const i32 rGap = (i32)(it.r2 - it.r1);
			if (qGap > GSA_GAP_CHECK || rGap > GSA_GAP_CHECK) {
				if (qGap > GSA_MAX_SEED_GAP || rGap > GSA_MAX_SEED_GAP) it.cut = 1; else it.jb = 1;
			}
		} else { it.q1 = it.q2 = 0; it.r1 = it.r2 = 0; }
	}
	__device__ i32 value(const Item &it, i64, int) const { return it.jb; }
	__device__ void emit(const Item &it, i64 i, const i32 *v, const i32 *ex) const
	{
		cut4[i] = it.cut;
		if (!v[0]) return;
		const i32 p = ex[0];
		jq1[p] = it.q1; jq2[p] = it.q2; jr1[p] = it.r1; jr2[p] = it.r2; jseed[p] = (i32)i;
	}
	__device__ void done(const i32 *t) const { mail[M_NJ] = t[0]; }
};

// S5 (CheckAlnBlockSpanMultipleRefChrs, :81-118) + leaf heads.  Within a block
// rPos is strictly increasing after S3, so "first seed past the end of the copy
// holding the piece's first seed" == "copy index changes".  Second component: prefix sums of the
// trimmed lengths, 32-bit wrapping -- only differences over a leaf are ever used.
// STEPS > 0 (round 5): the reference has fewer than 2^STEPS sequence ends, and both lower bounds of EVERY element are found by exactly STEPS probes without a branch --
// the probes of a thread's elements go out together (gsa_scan.h, clamped Ops).  The loop form (STEPS = 0: any number of ends) ran its two searches of ~6 dependent
// loads each for one element after the other: ~100 dependent loads per thread and tile.
template <int STEPS>
struct OpChrCutsT {
	static constexpr bool clamped = STEPS > 0;
	i64 ub; const i32 *d_n; DevIndex di; const i64 *r; const i32 *bid, *cut4, *len;
	i32 *cut5, *lstart, *head; u32 *ps; i32 *mail;
	__device__ void cuts(i64 i, i32 &c5, i32 &h) const
	{
		c5 = 0; h = 1;
		if (i > 0 && bid[i - 1] == bid[i]) {
			h = cut4[i];
			if (!cut4[i]) {
				// lower_bound on the sorted last coordinates (ChrLocMap)
				int lo0 = 0, hi0 = di.n_ends; const i64 a = r[i - 1]; while (lo0 < hi0) { int m = (lo0 + hi0) >> 1; if (di.chr_end[m] < a) lo0 = m + 1; else hi0 = m; }
				int lo1 = 0, hi1 = di.n_ends; const i64 b = r[i];     while (lo1 < hi1) { int m = (lo1 + hi1) >> 1; if (di.chr_end[m] < b) lo1 = m + 1; else hi1 = m; }
				if (lo0 != lo1) { c5 = 1; h = 1; }
			}
		}
	}
	__device__ int lower_fixed(i64 a) const
	{
		int lo = 0, hi = di.n_ends;
#pragma unroll
		for (int s = 0; s < STEPS; s++) {
			const int m = (lo + hi) >> 1;
			const i64 v = di.chr_end[m < di.n_ends ? m : di.n_ends - 1];
			const bool act = lo < hi, less = v < a;
			lo = (act && less) ? m + 1 : lo; hi = (act && !less) ? m : hi;
		}
		return lo;
	}
	struct Item { i32 in, c5, h, len, b0, b1, c4, lo0, lo1; };
	__device__ Item load(i64 i) const
	{
		Item it; it.in = 0; it.c5 = 0; it.h = 0; it.len = 0;
		if constexpr (STEPS > 0) {
			const i64 im = i > 0 ? i - 1 : 0;
			it.len = len[i]; it.b0 = bid[im]; it.b1 = bid[i]; it.c4 = cut4[i];
			it.lo0 = lower_fixed(r[im]); it.lo1 = lower_fixed(r[i]);
		} else {
			if (i >= *d_n) return it;
			it.in = 1; it.len = len[i]; cuts(i, it.c5, it.h);
		}
		return it;
	}
	__device__ void prep(Item &it, i64 i) const
	{
		if constexpr (STEPS > 0) {
			if (i >= *d_n) { it.len = 0; return; }
			it.in = 1; it.c5 = 0; it.h = 1;
			if (i > 0 && it.b0 == it.b1) {
				it.h = it.c4;
				if (!it.c4 && it.lo0 != it.lo1) { it.c5 = 1; it.h = 1; }
			}
		}
	}
	__device__ i32 value(const Item &it, i64, int c) const { return c == 1 ? it.len : it.h; }
	__device__ void emit(const Item &it, i64 i, const i32 *v, const i32 *ex) const
	{
		ps[i] = (u32)ex[1];                                    // (behind the last seed: the total)
		if (!it.in) { head[i] = 1; return; }
		cut5[i] = it.c5; head[i] = v[0];
		if (v[0]) lstart[ex[0]] = (i32)i;
	}
	__device__ void done(const i32 *t) const { mail[M_NL] = t[0]; ps[ub] = (u32)t[1]; head[ub] = 1; }
};

__global__ void k_leaf_emit(i64 ub, const i32 *__restrict__ d_n, const i32 *__restrict__ mail, const i32 *__restrict__ lstart, const i32 *__restrict__ q,
                            const i32 *__restrict__ len, const i64 *__restrict__ r, const i32 *__restrict__ bid, const i32 *__restrict__ cut4,
                            const i32 *__restrict__ cut5, const u32 *__restrict__ ps, const i32 *__restrict__ blk_score, Leaf *leaf,
                            i32 *hmail, Leaf *hleaf, i32 hleaf_cap)
{
	// last kernel in front of the host's look: the mailbox and the first leaves are written straight into pinned memory
	// (a copy operation behind the kernel costs ~15 us of stream latency each, these stores ride along)
	if (blockIdx.x == 0 && threadIdx.x < MAIL_N) hmail[threadIdx.x] = mail[threadIdx.x];
	GID(ub);
	const i32 nl = mail[M_NL];
	if (i >= nl) return;
	const i32 s = lstart[i], e = (i + 1 < nl) ? lstart[i + 1] : *d_n;
	Leaf L;
	L.beg = s; L.end = e; L.sumlen = (i32)(ps[e] - ps[s]);
	L.q_first = q[s]; L.q_last_end = q[e - 1] + len[e - 1]; L.r_first = r[s]; L.r_last_end = r[e - 1] + len[e - 1];
	L.blk = bid[s]; L.cut4 = cut4[s]; L.cut5 = cut5[s]; L.blk_score = blk_score[L.blk];
	leaf[i] = L;
	if (i < hleaf_cap) hleaf[i] = L;
}

int stage345_refine(gsa_ctx *c)
{
	hipStream_t st = c->stream;
	c->n_r = 0; c->h_leaf.clear(); c->blocks.clear(); c->have_host_seeds = false;
	const i64 ub = c->n_seeds;                    // every count below is <= the seed count
	if (ub == 0) { c->n_blocks2 = 0; c->n_c = 0; c->n_b = 0; return GSA_OK; }
	if (c->profiling) hipEventRecord(c->ev[6], st);
	i32 *mail = c->d_mail.as<i32>();
	ENS(i32, r_q, ub + 1); ENS(i32, r_len, ub + 1); ENS(i64, r_r, ub + 1); ENS(i32, r_bid, ub + 1);
	ENS(i32, r_tmp_q, ub + 1); ENS(i32, r_tmp_len, ub + 1); ENS(i64, r_tmp_r, ub + 1); ENS(i32, r_tmp_bid, ub + 1);
	ENS(i32, r_cut4, ub + 1); ENS(i32, r_cut5, ub + 1); ENS(i32, r_simres, ub + 1);
	// job arrays carved from the (free) stage-2 scratch
	ENS(i32, a_uniq, ub + 1); ENS(i32, a_cu, ub + 1); ENS(i64, w_best, ub + 1); ENS(i64, w_sum, ub + 1); ENS(i32, a_brk, ub + 1);
	i32 *cut4 = c->r_cut4.as<i32>(), *cut5 = c->r_cut5.as<i32>();
	i32 *jq1 = c->a_uniq.as<i32>(), *jq2 = c->a_cu.as<i32>(), *jseed = c->a_brk.as<i32>(); i64 *jr1 = c->w_best.as<i64>(), *jr2 = c->w_sum.as<i64>();
	ENS(i32, r_orig, ub + 1); ENS(i32, r_tmp_orig, ub + 1);
	{ OpTakeInBlock op = { c->c_q.as<i32>(), c->c_len.as<i32>(), c->c_r.as<i64>(), c->c_bid.as<i32>(),
	                       c->r_q.as<i32>(), c->r_len.as<i32>(), c->r_r.as<i64>(), c->r_bid.as<i32>(), c->r_orig.as<i32>(), mail }; RC((lb_launch<1>(c, ub, op))); }
	// S3: passes until nothing dies.  A pass that kills nothing is the identity, so two passes are issued
	// and then EVERYTHING behind them -- S4 gap scan + similarity jobs, S5 cuts, leaf table, the list of
	// large DP gaps -- with the live seed count read on the device; the host looks once, at the end, and
	// only if the second pass still killed something the tail is redone after two more passes.
	ENS(i32, a_next, ub + 1); ENS(u32, d_flag, ub + 2); ENS(i32, r_head, ub + 2); ENS(Leaf, d_leaf, ub + 1);
	i32 *lstart = c->a_next.as<i32>(); u32 *ps = c->d_flag.as<u32>();
	if (!pin_ensure<Leaf>(c, c->p_leaf, (size_t)LEAF_CHUNK)) return GSA_ERR_NOMEM;
	const size_t first = (size_t)std::min<i64>(ub, LEAF_CHUNK);
	int round = 0, cur = M_NR, oth = M_NR2;      // (mail[M_ANY ..] was cleared by the pass above)
	for (;;) {
		for (int k = 0; k < 2; k++, round++) {
			OpOverlapPass op = { c->r_q.as<i32>(), c->r_len.as<i32>(), c->r_r.as<i64>(), c->r_bid.as<i32>(), c->r_orig.as<i32>(),
			                     c->r_tmp_q.as<i32>(), c->r_tmp_len.as<i32>(), c->r_tmp_r.as<i64>(), c->r_tmp_bid.as<i32>(), c->r_tmp_orig.as<i32>(), mail, cur, oth, M_ANY + (round & 31) }; op.ub = ub;
			RC((lb_launch<1>(c, ub, op)));
			std::swap(c->r_q, c->r_tmp_q); std::swap(c->r_len, c->r_tmp_len); std::swap(c->r_r, c->r_tmp_r); std::swap(c->r_bid, c->r_tmp_bid); std::swap(c->r_orig, c->r_tmp_orig);
			std::swap(cur, oth);
		}
		// S4 cuts + similarity jobs
		{ OpGapCuts op = { cur, c->r_q.as<i32>(), c->r_len.as<i32>(), c->r_r.as<i64>(), c->r_bid.as<i32>(), cut4, jq1, jq2, jr1, jr2, jseed, mail }; RC((lb_launch<1>(c, ub, op))); }
		RC(run_gapsim_jobs(c, (i32)ub, mail + M_NJ, jq1, jq2, jr1, jr2, c->r_simres.as<i32>(), jseed, cut4));      // (sets cut4 of the dissimilar gaps)
		// S5 cuts + leaf table + large DP gaps of the leaves
		if (c->di.n_ends > 0 && c->di.n_ends < 256) { OpChrCutsT<8> op = { ub, mail + cur, c->di, c->r_r.as<i64>(), c->r_bid.as<i32>(), cut4, c->r_len.as<i32>(), cut5, lstart, c->r_head.as<i32>(), ps, mail }; RC((lb_launch<2>(c, ub, op))); }
		else { OpChrCutsT<0> op = { ub, mail + cur, c->di, c->r_r.as<i64>(), c->r_bid.as<i32>(), cut4, c->r_len.as<i32>(), cut5, lstart, c->r_head.as<i32>(), ps, mail }; RC((lb_launch<2>(c, ub, op))); }
		LAUNCH(k_leaf_emit, ub, ub, mail + cur, mail, lstart, c->r_q.as<i32>(), c->r_len.as<i32>(), c->r_r.as<i64>(), c->r_bid.as<i32>(), cut4, cut5, ps,
		       c->blk_score.as<i32>(), c->d_leaf.as<Leaf>(), c->h_mail, c->p_leaf.as<Leaf>(), (i32)first);
		// (the mailbox and the first LEAF_CHUNK leaves are in pinned memory when this kernel is done)
		if (c->profiling) hipEventRecord(c->ev[7], st);
		// everything of S3-S5 is enqueued: now start the striped DP for the large gaps stage 2 listed (its own stream)
		RC(launch_early_dp(c));
		GSA_CHECK(c, hipStreamSynchronize(st));
		if (c->h_mail[M_LBERR]) return gsa_fail(c, GSA_ERR_STATE, "internal: look-back scan timed out");
		if (!c->h_mail[M_ANY + ((round - 1) & 31)]) break;
		if (round >= 2000) return gsa_fail(c, GSA_ERR_STATE, "internal: RemoveOverlaps does not converge");
		if ((round & 31) == 0) GSA_CHECK(c, hipMemsetAsync(mail + M_ANY, 0, 32 * sizeof(i32), st));
	}
	collect_events(c);
	c->n_b = c->h_mail[M_NB]; c->n_c = c->h_mail[M_NC]; c->n_blocks2 = c->h_mail[M_NBLK];
	const i64 nr = c->h_mail[cur];
	c->n_r = nr;
	if (c->n_blocks2 == 0 || nr == 0) { c->n_r = 0; return GSA_OK; }
	if (c->h_mail[M_LBERR]) return gsa_fail(c, GSA_ERR_STATE, "internal: look-back scan timed out");
	const i32 nl = c->h_mail[M_NL];
	if ((size_t)nl > first) {
		if (!pin_ensure<Leaf>(c, c->p_leaf, (size_t)nl)) return GSA_ERR_NOMEM;
		GSA_CHECK(c, hipMemcpyAsync(c->p_leaf.p, c->d_leaf.p, (size_t)nl * sizeof(Leaf), hipMemcpyDeviceToHost, st));
		GSA_CHECK(c, hipStreamSynchronize(st));
	}
	c->h_leaf.assign(c->p_leaf.as<Leaf>(), c->p_leaf.as<Leaf>() + nl);
	if (c->profiling) { float ms; hipEventElapsedTime(&ms, c->ev[6], c->ev[7]); c->kernel_ms[4] = ms; }
	// host list after S3 = the S2 blocks, in S2 order, with their S2 scores
	c->blocks.resize(c->n_blocks2);
	{
		size_t l = 0;
		for (i32 b = 0; b < c->n_blocks2; b++) {
			HostBlock &hb = c->blocks[b];
			hb.leaf_beg = (i32)l;
			while (l < c->h_leaf.size() && c->h_leaf[l].blk == b) l++;
			hb.leaf_end = (i32)l; hb.bdup = 0; hb.aln_len = 0; hb.bdir = 0; hb.gpos = 0; hb.chr = 0;
			if (hb.leaf_end == hb.leaf_beg) return gsa_fail(c, GSA_ERR_STATE, "internal: S2 block without seeds after S3");
			hb.score = c->h_leaf[hb.leaf_beg].blk_score;
		}
	}
	return GSA_OK;
}
