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

#define TPB 256
#define GID(n) i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; if (i >= (n)) return
#define LAUNCH(k, n, ...) hipLaunchKernelGGL(k, dim3(grid_for((size_t)(n), TPB)), dim3(TPB), 0, st, __VA_ARGS__)
#define ENS(T, buf, n) do { if (!dev_ensure<T>(c, c->buf, (size_t)(n))) return GSA_ERR_NOMEM; } while (0)
#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// `ub` is the host-known upper bound (the seed count); the live count sits in the mailbox.
__global__ void k_inblock_flag(i64 ub, const i32 *__restrict__ d_n, const i32 *__restrict__ bid, i32 *flag)
{
	i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > ub) return;
	flag[i] = (i < *d_n && bid[i] >= 0) ? 1 : 0;
}

__global__ void k_take(i64 n, const i32 *__restrict__ keep, const i32 *__restrict__ ex, const i32 *__restrict__ q, const i32 *__restrict__ len,
                       const i64 *__restrict__ r, const i32 *__restrict__ bid, i32 *oq, i32 *olen, i64 *orr, i32 *obid, i32 *d_nout)
{
	GID(n);
	if (i == 0) *d_nout = ex[n];
	if (!keep[i]) return;
	const i32 p = ex[i];
	oq[p] = q[i]; olen[p] = len[i]; orr[p] = r[i]; obid[p] = bid[i];
}

// one RemoveOverlaps pass (ProcessCandidateAlignment.cpp:197-226)
__global__ void k_overlap_pass(i64 ub, const i32 *__restrict__ d_n, const i32 *__restrict__ q, i32 *len, const i64 *__restrict__ r, const i32 *__restrict__ bid, i32 *keep, i32 *anykill)
{
	i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > ub) return;
	const i64 n = *d_n;
	if (i >= n) { keep[i] = 0; return; }
	i32 k = 1;
	if (i + 1 < n && bid[i + 1] == bid[i]) {
		const i64 ri = r[i], rj = r[i + 1]; const i32 qi = q[i], qj = q[i + 1];
		i32 l = len[i];
		if (rj <= ri) k = 0;
		else {
			i32 ov = (i32)(ri + l - rj);
			if (ov > 0) { l -= ov; if (l <= 0) k = 0; }
			if (k) { ov = qi + l - qj; if (ov > 0) { l -= ov; if (l <= 0) k = 0; } }
			len[i] = l;      // a killed seed's length is never read again
		}
	}
	keep[i] = k;
	if (!k) *anykill = 1;
}

// S4 (CheckGapsBetweenSeeds, :120-156): cut4[i] = 1 cut before i; job[i] = needs CalGapSimilarity
__global__ void k_gap_cuts(i64 ub, const i32 *__restrict__ d_n, const i32 *__restrict__ q, const i32 *__restrict__ len, const i64 *__restrict__ r, const i32 *__restrict__ bid,
                           i32 *cut4, i32 *job)
{
	i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > ub) return;
	const i64 n = *d_n;
	i32 cut = 0, jb = 0;
	if (i < n && i > 0 && bid[i - 1] == bid[i]) {
		const i32 qGap = q[i] - q[i - 1] - len[i - 1];
		const i32 rGap = (i32)(r[i] - r[i - 1] - len[i - 1]);
		if (qGap > GSA_GAP_CHECK || rGap > GSA_GAP_CHECK) {
			if (qGap > GSA_MAX_SEED_GAP || rGap > GSA_MAX_SEED_GAP) cut = 1; else jb = 1;
		}
	}
	cut4[i] = cut; job[i] = jb;
}

__global__ void k_gap_jobs(i64 n, const i32 *__restrict__ job, const i32 *__restrict__ jobEx, const i32 *__restrict__ q, const i32 *__restrict__ len,
                           const i64 *__restrict__ r, i32 *jq1, i32 *jq2, i64 *jr1, i64 *jr2, i32 *jseed)
{
	GID(n);
	if (!job[i]) return;
	const i32 p = jobEx[i];
	jq1[p] = q[i - 1] + len[i - 1]; jq2[p] = q[i]; jr1[p] = r[i - 1] + len[i - 1]; jr2[p] = r[i]; jseed[p] = (i32)i;
}

__global__ void k_gap_apply(i32 nj, const i32 *__restrict__ jseed, const i32 *__restrict__ res, i32 *cut4)
{
	GID(nj);
	if (!res[i]) cut4[jseed[i]] = 1;
}

// S5 (CheckAlnBlockSpanMultipleRefChrs, :81-118) + leaf heads.  Within a block
// rPos is strictly increasing after S3, so "first seed past the end of the copy
// holding the piece's first seed" == "copy index changes".
__global__ void k_chr_cuts(i64 n, DevIndex di, const i64 *__restrict__ r, const i32 *__restrict__ bid, const i32 *__restrict__ cut4, i32 *cut5, i32 *head)
{
	i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > n) return;
	if (i == n) { head[i] = 0; return; }
	i32 c5 = 0, h = 1;
	if (i > 0 && bid[i - 1] == bid[i]) {
		h = cut4[i];
		if (!cut4[i]) {
			// lower_bound on the sorted last coordinates (ChrLocMap)
			int lo0 = 0, hi0 = di.n_ends; const i64 a = r[i - 1]; while (lo0 < hi0) { int m = (lo0 + hi0) >> 1; if (di.chr_end[m] < a) lo0 = m + 1; else hi0 = m; }
			int lo1 = 0, hi1 = di.n_ends; const i64 b = r[i];     while (lo1 < hi1) { int m = (lo1 + hi1) >> 1; if (di.chr_end[m] < b) lo1 = m + 1; else hi1 = m; }
			if (lo0 != lo1) { c5 = 1; h = 1; }
		}
	}
	cut5[i] = c5; head[i] = h;
}

__global__ void k_leaf_emit(i64 n, const i32 *__restrict__ head, const i32 *__restrict__ headEx, const i32 *__restrict__ lstart, const i32 *__restrict__ q,
                            const i32 *__restrict__ len, const i64 *__restrict__ r, const i32 *__restrict__ bid, const i32 *__restrict__ cut4,
                            const i32 *__restrict__ cut5, const i64 *__restrict__ ps, const i32 *__restrict__ blk_score, Leaf *leaf, i32 *d_nl)
{
	GID(n);
	const i32 nl = headEx[n];
	if (i == 0) *d_nl = nl;
	if (i >= nl) return;
	const i32 s = lstart[i], e = (i + 1 < nl) ? lstart[i + 1] : (i32)n;
	Leaf L;
	L.beg = s; L.end = e; L.sumlen = (i32)(ps[e] - ps[s]);
	L.q_first = q[s]; L.q_last_end = q[e - 1] + len[e - 1]; L.r_first = r[s]; L.r_last_end = r[e - 1] + len[e - 1];
	L.blk = bid[s]; L.cut4 = cut4[s]; L.cut5 = cut5[s]; L.blk_score = blk_score[L.blk];
	leaf[i] = L;
}

__global__ void k_scatter_idx2(i64 n, const i32 *__restrict__ flag, const i32 *__restrict__ ex, i32 *list)
{
	GID(n);
	if (flag[i]) list[ex[i]] = (i32)i;
}

int stage345_refine(gsa_ctx *c)
{
	hipStream_t st = c->stream;
	c->n_r = 0; c->h_leaf.clear(); c->blocks.clear(); c->have_host_seeds = false;
	const i64 ub = c->n_seeds;                    // every count below is <= the seed count
	if (ub == 0) { c->n_blocks2 = 0; c->n_c = 0; c->n_b = 0; return GSA_OK; }
	if (c->profiling) hipEventRecord(c->ev[6], st);
	i32 *mail = c->d_mail.as<i32>();
	// seeds that belong to a kept S2 block
	ENS(i32, d_flag, ub + 1); ENS(i32, d_scan, ub + 1);
	i32 *flag = c->d_flag.as<i32>(), *ex = c->d_scan.as<i32>();
	ENS(i32, r_q, ub + 1); ENS(i32, r_len, ub + 1); ENS(i64, r_r, ub + 1); ENS(i32, r_bid, ub + 1);
	ENS(i32, r_tmp_q, ub + 1); ENS(i32, r_tmp_len, ub + 1); ENS(i64, r_tmp_r, ub + 1); ENS(i32, r_tmp_bid, ub + 1);
	ENS(i32, r_cut4, ub + 1); ENS(i32, r_cut5, ub + 1); ENS(i32, r_simjob, ub + 1); ENS(i32, r_simres, ub + 1);
	i32 *cut4 = c->r_cut4.as<i32>(), *cut5 = c->r_cut5.as<i32>(), *job = c->r_simjob.as<i32>();
	LAUNCH(k_inblock_flag, ub + 1, ub, mail + M_NC, c->c_bid.as<i32>(), flag);
	RC(prim_exscan_i32(c, flag, ex, (size_t)ub + 1));
	LAUNCH(k_take, ub, ub, flag, ex, c->c_q.as<i32>(), c->c_len.as<i32>(), c->c_r.as<i64>(), c->c_bid.as<i32>(),
	       c->r_q.as<i32>(), c->r_len.as<i32>(), c->r_r.as<i64>(), c->r_bid.as<i32>(), mail + M_NR);
	// S3: passes until nothing dies.  A pass that kills nothing followed by its compaction is the
	// identity, so the passes are issued two at a time with the S4 gap scan behind them and the
	// host looks at the kill flags once per batch.
	GSA_CHECK(c, hipMemsetAsync(mail + M_ANY, 0, 32 * sizeof(i32), st));
	int round = 0;
	for (;;) {
		for (int k = 0; k < 2; k++, round++) {
			LAUNCH(k_overlap_pass, ub + 1, ub, mail + M_NR, c->r_q.as<i32>(), c->r_len.as<i32>(), c->r_r.as<i64>(), c->r_bid.as<i32>(), flag, mail + M_ANY + (round & 31));
			RC(prim_exscan_i32(c, flag, ex, (size_t)ub + 1));
			LAUNCH(k_take, ub, ub, flag, ex, c->r_q.as<i32>(), c->r_len.as<i32>(), c->r_r.as<i64>(), c->r_bid.as<i32>(),
			       c->r_tmp_q.as<i32>(), c->r_tmp_len.as<i32>(), c->r_tmp_r.as<i64>(), c->r_tmp_bid.as<i32>(), mail + M_NR);
			std::swap(c->r_q, c->r_tmp_q); std::swap(c->r_len, c->r_tmp_len); std::swap(c->r_r, c->r_tmp_r); std::swap(c->r_bid, c->r_tmp_bid);
		}
		// S4 cuts (speculative: valid if the last pass killed nothing)
		LAUNCH(k_gap_cuts, ub + 1, ub, mail + M_NR, c->r_q.as<i32>(), c->r_len.as<i32>(), c->r_r.as<i64>(), c->r_bid.as<i32>(), cut4, job);
		RC(prim_exscan_i32(c, job, ex, (size_t)ub + 1));
		GSA_CHECK(c, hipMemcpyAsync(mail + M_NJ, ex + ub, 4, hipMemcpyDeviceToDevice, st));
		GSA_CHECK(c, hipMemcpyAsync(c->h_mail, mail, MAIL_N * sizeof(i32), hipMemcpyDeviceToHost, st));
		GSA_CHECK(c, hipStreamSynchronize(st));
		if (!c->h_mail[M_ANY + ((round - 1) & 31)]) break;
		if (round >= 2000) return gsa_fail(c, GSA_ERR_STATE, "internal: RemoveOverlaps does not converge");
		if ((round & 31) == 0) GSA_CHECK(c, hipMemsetAsync(mail + M_ANY, 0, 32 * sizeof(i32), st));
	}
	collect_events(c);
	c->n_b = c->h_mail[M_NB]; c->n_c = c->h_mail[M_NC]; c->n_blocks2 = c->h_mail[M_NBLK];
	const i64 nr = c->h_mail[M_NR]; const i32 nj = c->h_mail[M_NJ];
	c->n_r = nr;
	if (c->n_blocks2 == 0 || nr == 0) { c->n_r = 0; return GSA_OK; }
	if (nj > 0) {
		// job arrays carved from the (free) stage-2 scratch
		ENS(i32, a_uniq, nj); ENS(i32, a_cu, nj); ENS(i64, w_best, nj); ENS(i64, w_sum, nj); ENS(i32, a_brk, nj);
		i32 *jq1 = c->a_uniq.as<i32>(), *jq2 = c->a_cu.as<i32>(), *jseed = c->a_brk.as<i32>(); i64 *jr1 = c->w_best.as<i64>(), *jr2 = c->w_sum.as<i64>();
		LAUNCH(k_gap_jobs, nr, nr, job, ex, c->r_q.as<i32>(), c->r_len.as<i32>(), c->r_r.as<i64>(), jq1, jq2, jr1, jr2, jseed);
		RC(run_gapsim_jobs(c, nj, jq1, jq2, jr1, jr2, c->r_simres.as<i32>()));
		LAUNCH(k_gap_apply, nj, nj, jseed, c->r_simres.as<i32>(), cut4);
	}
	// S5 cuts + leaf table
	LAUNCH(k_chr_cuts, nr + 1, nr, c->di, c->r_r.as<i64>(), c->r_bid.as<i32>(), cut4, cut5, flag);
	RC(prim_exscan_i32(c, flag, ex, (size_t)nr + 1));
	ENS(i32, a_next, nr + 1); ENS(i64, d_i64a, nr + 2);
	i32 *lstart = c->a_next.as<i32>();
	LAUNCH(k_scatter_idx2, nr, nr, flag, ex, lstart);
	// prefix sums of the trimmed lengths (zero tail)
	GSA_CHECK(c, hipMemsetAsync(c->r_len.as<i32>() + nr, 0, 4, st));
	RC(prim_exscan_i32_i64(c, c->r_len.as<i32>(), c->d_i64a.as<i64>(), (size_t)nr + 1));
	ENS(Leaf, d_leaf, nr + 1);
	LAUNCH(k_leaf_emit, nr, nr, flag, ex, lstart, c->r_q.as<i32>(), c->r_len.as<i32>(), c->r_r.as<i64>(), c->r_bid.as<i32>(), cut4, cut5, c->d_i64a.as<i64>(),
	       c->blk_score.as<i32>(), c->d_leaf.as<Leaf>(), mail + M_NL);
	// the leaf count and the first LEAF_CHUNK leaves come back together
	const size_t first = (size_t)std::min<i64>(nr, LEAF_CHUNK);
	if (!pin_ensure<Leaf>(c, c->p_leaf, (size_t)LEAF_CHUNK)) return GSA_ERR_NOMEM;
	GSA_CHECK(c, hipMemcpyAsync(c->h_mail + M_NL, mail + M_NL, sizeof(i32), hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipMemcpyAsync(c->p_leaf.p, c->d_leaf.p, first * sizeof(Leaf), hipMemcpyDeviceToHost, st));
	if (c->profiling) hipEventRecord(c->ev[7], st);
	GSA_CHECK(c, hipStreamSynchronize(st));
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
