// gsalign_amd/csrc/gsa_blocks.cpp -- host-side bookkeeping of the alignment-block
// LIST (a8 tail, a10, a15): which blocks exist, in which order, with which score.
//
// The per-seed work of these stages runs on the GPU (k_refine.hip, k_extend.hip);
// what stays here touches O(#blocks) records: splitting a block into the pieces
// the device cut, RemoveBadAlnBlocks, EstChromosomeSimilarity,
// RemoveRedundantAlnBlocks, the identity filter and GenCoordinateInfo
// (reference src/ProcessCandidateAlignment.cpp:72-79,101-117,140-155;
// src/GSAlign.cpp:393-471,529-540; src/tools.cpp:120-140,305-312).
// It is host code on purpose: the reference orders blocks with std::sort on an
// incomplete key (score only, App. B #10), so the order of equal-score blocks is
// whatever libstdc++'s introsort does to the incoming permutation -- the same
// std::sort on the same permutation is the way to be bit-identical.
#include <algorithm>
#include <chrono>
#include <cstring>
#include "gsa_ctx.h"

int stage7_fill(gsa_ctx *c);

namespace {

struct ByScore { bool operator()(const HostBlock &a, const HostBlock &b) const { return a.score > b.score; } };

// RemoveBadAlnBlocks (ProcessCandidateAlignment.cpp:72-79)
void remove_bad(std::vector<HostBlock> &v)
{
	std::sort(v.begin(), v.end(), ByScore());
	size_t n = v.size(); while (n > 0 && v[n - 1].score == 0) n--;
	v.resize(n);
}

// CalAlnBlockScore (:26-36) of the leaf range [lb, le)
int piece_score(const gsa_ctx *c, int lb, int le)
{
	if (le <= lb) return 0;
	if (c->h_leaf[le - 1].q_last_end - c->h_leaf[lb].q_first < c->prm.MinAlnLength) return 0;
	int s = 0; for (int l = lb; l < le; l++) s += c->h_leaf[l].sumlen;
	return s;
}

// split every block of the current list at the leaves flagged by `which` (4 or 5);
// pieces go to the END of the list in parent order, the parent is zeroed
void split_blocks(gsa_ctx *c, int which)
{
	const size_t nb = c->blocks.size();
	for (size_t b = 0; b < nb; b++) {
		const int lb = c->blocks[b].leaf_beg, le = c->blocks[b].leaf_end;
		bool any = false;
		for (int l = lb + 1; l < le && !any; l++) any = which == 4 ? c->h_leaf[l].cut4 != 0 : c->h_leaf[l].cut5 != 0;
		if (!any) continue;
		c->blocks[b].score = 0;
		int s = lb;
		for (int l = lb + 1; l <= le; l++) {
			const bool cut = (l == le) || (which == 4 ? c->h_leaf[l].cut4 != 0 : c->h_leaf[l].cut5 != 0);
			if (!cut) continue;
			HostBlock p; p.leaf_beg = s; p.leaf_end = l; p.bdup = 0; p.aln_len = 0; p.bdir = 0; p.gpos = 0; p.chr = 0;
			p.score = piece_score(c, s, l);
			if (p.score > c->prm.MinAlnBlockScore) c->blocks.push_back(p);       // strict '>' (App. B #5)
			s = l;
		}
	}
	remove_bad(c->blocks);
}

inline int chr_of(const gsa_ctx *c, i64 rpos, i64 *end_key = nullptr)
{
	size_t k = std::lower_bound(c->h_chr_end.begin(), c->h_chr_end.end(), rpos) - c->h_chr_end.begin();
	if (end_key) *end_key = c->h_chr_end[k];
	return c->h_chr_of_end[k];
}

struct ByQ {
	const gsa_ctx *c;
	bool operator()(const HostBlock &a, const HostBlock &b) const {
		const i32 qa = c->h_leaf[a.leaf_beg].q_first, qb = c->h_leaf[b.leaf_beg].q_first;
		return qa == qb ? a.score > b.score : qa < qb;                          // CompByAlnBlockQueryPos, GSAlign.cpp:17-21
	}
};
struct ByR {
	const gsa_ctx *c;
	bool operator()(const HostBlock &a, const HostBlock &b) const {
		const i64 ra = c->h_leaf[a.leaf_beg].r_first, rb = c->h_leaf[b.leaf_beg].r_first;
		return ra == rb ? a.score > b.score : ra < rb;                          // CompByAlnBlockRefPos, :23-27
	}
};

// RemoveRedundantAlnBlocks (GSAlign.cpp:415-471)
void remove_redundant(gsa_ctx *c, int type, const std::vector<i64> &chr_score)
{
	std::vector<HostBlock> &B = c->blocks;
	const int nb = (int)B.size();
	if (type == 1) { ByQ cmp = { c }; std::sort(B.begin(), B.end(), cmp); } else { ByR cmp = { c }; std::sort(B.begin(), B.end(), cmp); }
	const i64 G = c->G, G2 = 2 * c->G;
	for (int i = 0; i < nb; i++) {
		if (B[i].score == 0) continue;
		const Leaf &f1 = c->h_leaf[B[i].leaf_beg], &l1 = c->h_leaf[B[i].leaf_end - 1];
		i64 h1 = type == 1 ? f1.q_first : f1.r_first, t1 = type == 1 ? (i64)l1.q_last_end - 1 : l1.r_last_end - 1;
		const int c1 = chr_of(c, f1.r_first);
		if (type == 2 && h1 >= G) { const i64 t = h1; h1 = G2 - 1 - t1; t1 = G2 - 1 - t; }      // ReverseRefCoordinate, after sorting (App. B #11)
		for (int j = i + 1; j < nb; j++) {
			if (B[j].score == 0) continue;
			const Leaf &f2 = c->h_leaf[B[j].leaf_beg], &l2 = c->h_leaf[B[j].leaf_end - 1];
			i64 h2 = type == 1 ? f2.q_first : f2.r_first, t2 = type == 1 ? (i64)l2.q_last_end - 1 : l2.r_last_end - 1;
			if (type == 1 && h1 == h2 && t1 == t2) { B[i].bdup = 1; B[j].score = 0; continue; }
			const int c2 = chr_of(c, f2.r_first);
			if (type == 2 && h2 >= G) { const i64 t = h2; h2 = G2 - 1 - t2; t2 = G2 - 1 - t; }
			if (h2 < t1) {
				const i64 ov = t2 > t1 ? t1 - h2 : t2 - h2;
				const float f1r = 1. * ov / (t1 - h1), f2r = 1. * ov / (t2 - h2);                 // float holding a double quotient (App. B #9)
				const int s1 = (int)chr_score[c1], s2 = (int)chr_score[c2];                      // CheckDuplicatedChrScore takes int
				if ((f1r > f2r && f1r >= 0.9) || (c->prm.OneOnOne && (s2 > s1 && s2 >= s1 * 2))) { B[i].score = 0; break; }
				if ((f2r > f1r && f2r >= 0.9) || (c->prm.OneOnOne && (s1 > s2 && s1 >= s2 * 2))) B[j].score = 0;
			} else break;
		}
	}
	remove_bad(B);
}

} // namespace

// ---- a bundle of contigs (Bundle, gsa_internal.h): the reference keeps ONE AlnBlockVec per query sequence and clears it between
// sequences (GSAlign.cpp:483-490), and its std::sort calls see that sequence's blocks only -- so the list logic runs per contig,
// on that contig's blocks in the order the device left them (S2 order: groups ascend with the contig's PosDiff stride).
static inline int contig_of_q(const gsa_ctx *c, i32 q)
{
	const i32 *o = c->b_off.data() + 1;            // contig k = [b_off[k], b_off[k + 1]); empty contigs share their start with the next one
	return (int)(std::upper_bound(o, o + c->bnd.n, q) - o);
}
void bundle_split_lists(gsa_ctx *c)
{
	c->b_lists.assign((size_t)c->bnd.n, std::vector<HostBlock>());
	for (const HostBlock &hb : c->blocks) c->b_lists[(size_t)contig_of_q(c, c->h_leaf[hb.leaf_beg].q_first)].push_back(hb);
	c->blocks.clear();
}
void bundle_join_lists(gsa_ctx *c)
{
	c->blocks.clear(); c->b_blk0.assign((size_t)c->bnd.n + 1, 0);
	for (int k = 0; k < c->bnd.n; k++) {
		c->b_blk0[(size_t)k] = (i32)c->blocks.size();
		c->blocks.insert(c->blocks.end(), c->b_lists[(size_t)k].begin(), c->b_lists[(size_t)k].end());
	}
	c->b_blk0[(size_t)c->bnd.n] = (i32)c->blocks.size();
	c->b_lists.clear();
}

static int host_stage4_5_6_one(gsa_ctx *c, int stage);
int host_stage4_5_6(gsa_ctx *c, int stage)
{
	if (!c->bnd.n) return host_stage4_5_6_one(c, stage);
	int rc = GSA_OK;
	for (std::vector<HostBlock> &L : c->b_lists) { c->blocks.swap(L); const int r1 = host_stage4_5_6_one(c, stage); c->blocks.swap(L); if (r1 != GSA_OK && rc == GSA_OK) rc = r1; }
	return rc;
}

static int host_stage4_5_6_one(gsa_ctx *c, int stage)
{
	auto t0 = std::chrono::steady_clock::now();
	if (stage == 4) split_blocks(c, 4);
	else if (stage == 5) split_blocks(c, 5);
	else {
		for (size_t b = 0; b < c->blocks.size(); b++) c->blocks[b].bdup = 0;                      // GSAlign.cpp:510
		std::vector<i64> chr_score(c->h_chr_len.size(), 0);                                      // EstChromosomeSimilarity :393-407
		for (size_t b = 0; b < c->blocks.size(); b++) chr_score[chr_of(c, c->h_leaf[c->blocks[b].leaf_beg].r_first)] += c->blocks[b].score;
		remove_redundant(c, 1, chr_score); remove_redundant(c, 2, chr_score);
	}
	c->kernel_ms[7] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
	return GSA_OK;
}

// identity filter + GenCoordinateInfo + final RemoveBadAlnBlocks (GSAlign.cpp:529-540)
int host_stage8_finish(gsa_ctx *c)
{
	const size_t nfb = c->blocks.size();
	c->h_blocks.clear(); c->h_frags.clear(); c->h_aln1.clear(); c->h_aln2.clear(); c->result_pinned = false;
	if (nfb == 0) return GSA_OK;
	// stage78_extend left everything in flight: per-block sums, records, patch list, string pools, mailbox.
	// The records came early, without the string lengths of their DP gaps: those of the small jobs are patched in while
	// the striped kernel is still running (the large ones below, from the patch list)
	GSA_CHECK(c, hipEventSynchronize(c->ev[15]));
	{
		gsa_rec *fr = c->p_frags.as<gsa_rec>();
		const i32 *rec = c->p_jpatch.as<i32>(), *len = rec + c->n_jobs;
		for (i32 j = 0; j < c->n_jobs; j++) fr[rec[j]].gap.aln_len = len[j];
	}
	GSA_CHECK(c, hipStreamSynchronize(c->stream));
	if (c->profiling) { float ms; if (hipEventElapsedTime(&ms, c->ev[8], c->ev[9]) == hipSuccess) c->kernel_ms[5] = ms; (void)hipGetLastError(); }
	i32 *bl_len = c->p_blk.as<i32>(), *bl_score = bl_len + nfb, *fragbase = bl_score + nfb;
	const i32 *hm = c->h_tmail;
	if (hm[M_LBERR]) return gsa_fail(c, GSA_ERR_STATE, "internal: look-back scan timed out");
	if (hm[M_DPERR2]) { c->dp_dirty = c->dp_timeout = true; return gsa_fail(c, GSA_ERR_STATE, "internal: DP stripe hand-off timed out"); }
	if (c->dp_fake_timeout > 0 && !c->dp_safe) { c->dp_fake_timeout--; c->dp_dirty = c->dp_timeout = true; return gsa_fail(c, GSA_ERR_STATE, "internal: DP stripe hand-off timed out (test hook)"); }
	if (c->profiling) { const unsigned long long *cc = (const unsigned long long *)(hm + M_CELLS); c->counters[4] += cc[0]; c->counters[6] += cc[1]; }
	if (c->n_early > 0 && hm[M_DPERR3]) { c->dp_dirty = c->dp_timeout = true; return gsa_fail(c, GSA_ERR_STATE, "internal: DP stripe hand-off timed out (early launch)"); }
	// the large DP jobs finished after the records left: their (aln_len, score) arrive as a patch list
	// (first the ones that only turned up in the job list, then the ones launched from the leaf table;
	//  record -1 = an early job whose leaf the list logic dropped)
	{
		gsa_rec *fr = c->p_frags.as<gsa_rec>();
		const i32 *pt = c->h_tpatch;
		const i32 np = c->n_large + c->n_early;
		for (i32 g = 0; g < np; g++) {
			const i32 rec = pt[3 * g], L = pt[3 * g + 1], sc = pt[3 * g + 2];
			if (rec < 0) continue;
			fr[rec].gap.aln_len = L;
			const size_t k = (size_t)(std::upper_bound(fragbase, fragbase + nfb, rec) - fragbase) - 1;      // block of the record
			bl_len[k] += L; bl_score[k] += sc;
			if (c->profiling && g >= c->n_large) {
				const i32 *e = &c->h_early[3 * (size_t)(g - c->n_large)];
				c->counters[4] += (u64)e[1] * (u64)e[2]; c->counters[5] += 1; c->counters[6] += (u64)(e[1] + e[2]);
			}
		}
	}
	auto t0 = std::chrono::steady_clock::now();
	std::vector<HostBlock> &B = c->blocks;
	// keep (frag_off, n_frag) beside each block while the list is re-sorted
	struct Ext { HostBlock b; i64 frag_off; i32 n_frag; };
	std::vector<Ext> E(nfb);
	for (size_t k = 0; k < nfb; k++) {
		E[k].b = B[k]; E[k].b.aln_len = bl_len[k]; E[k].b.score = bl_score[k];
		E[k].frag_off = fragbase[k];
		E[k].n_frag = (i32)((k + 1 < nfb ? (i64)fragbase[k + 1] : c->n_frags) - fragbase[k]);
	}
	for (size_t k = 0; k < nfb; k++) {
		HostBlock &hb = E[k].b;
		if ((int)(100 * (1.0 * hb.score / hb.aln_len)) < c->prm.MinSeqIdy) hb.score = 0;
		else {
			const i64 rpos = c->h_leaf[hb.leaf_beg].r_first;
			i64 key; hb.chr = chr_of(c, rpos, &key);
			if (rpos < c->G) { hb.bdir = 1; hb.gpos = (i32)(rpos + 1 - c->h_chr_fwd[hb.chr]); }
			else { hb.bdir = 0; hb.gpos = (i32)(key - rpos + 1); }
		}
	}
	struct ByScoreE { bool operator()(const Ext &a, const Ext &b) const { return a.b.score > b.b.score; } };
	// final RemoveBadAlnBlocks (GSAlign.cpp:540) -- per query sequence: a bundle sorts every contig's segment of the list by itself
	// and hands out record offsets relative to the contig's first record
	const int nseg = c->bnd.n ? c->bnd.n : 1;
	if (c->bnd.n) { c->b_nblk.assign((size_t)nseg, 0); c->b_frag0.assign((size_t)nseg + 1, c->n_frags); }
	B.clear(); c->h_blocks.clear();
	for (int sgi = 0; sgi < nseg; sgi++) {
		const size_t sb = c->bnd.n ? (size_t)c->b_blk0[(size_t)sgi] : 0, se = c->bnd.n ? (size_t)c->b_blk0[(size_t)sgi + 1] : nfb;
		const i64 f0 = c->bnd.n ? (sb < nfb ? (i64)fragbase[sb] : c->n_frags) : 0;
		if (c->bnd.n) c->b_frag0[(size_t)sgi] = f0;
		std::sort(E.begin() + sb, E.begin() + se, ByScoreE());
		size_t m = se; while (m > sb && E[m - 1].b.score == 0) m--;
		for (size_t k = sb; k < m; k++) {
			B.push_back(E[k].b);
			gsa_block o;
			o.score = E[k].b.score; o.aln_len = E[k].b.aln_len; o.bdup = E[k].b.bdup; o.n_frag = E[k].n_frag; o.frag_off = E[k].frag_off - f0;
			o.bdir = E[k].b.bdir; o.gpos = E[k].b.gpos; o.chr = E[k].b.chr; o._pad = 0;
			c->h_blocks.push_back(o);
		}
		if (c->bnd.n) c->b_nblk[(size_t)sgi] = (i32)(m - sb);
	}
	c->kernel_ms[7] += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
	c->result_pinned = true;
	c->frags_stage = 8;
	return GSA_OK;
}

// materialise the current AlnBlockVec for the getters (stages 2..7; stage 8 is built above)
int build_block_view(gsa_ctx *c)
{
	c->h_blocks.clear(); c->h_frags.clear(); c->h_aln1.clear(); c->h_aln2.clear(); c->result_pinned = false;
	if (c->stage == 8) { c->frags_stage = 8; return GSA_OK; }     // nothing survived
	if (c->stage == 2) {
		if (int rc = stage2_fetch_host(c)) return rc;
		const size_t nc = (size_t)c->n_c;
		std::vector<i32> q(nc), l(nc); std::vector<i64> r(nc);
		if (nc) {
			GSA_CHECK(c, hipMemcpy(q.data(), c->c_q.p, nc * 4, hipMemcpyDeviceToHost));
			GSA_CHECK(c, hipMemcpy(l.data(), c->c_len.p, nc * 4, hipMemcpyDeviceToHost));
			GSA_CHECK(c, hipMemcpy(r.data(), c->c_r.p, nc * 8, hipMemcpyDeviceToHost));
		}
		for (i32 b = 0; b < c->n_blocks2; b++) {
			gsa_block o; memset(&o, 0, sizeof(o));
			o.score = c->h_blk_score[b]; o.frag_off = (i64)c->h_frags.size(); o.n_frag = c->h_blk_end[b] - c->h_blk_beg[b];
			for (i32 s = c->h_blk_beg[b]; s < c->h_blk_end[b]; s++) {
				gsa_rec f; f.seed.qpos = q[s]; f.seed.len = l[s]; f.seed.rpos = r[s];
				c->h_frags.push_back(f);
			}
			c->h_blocks.push_back(o);
		}
		c->frags_stage = 2;
		return GSA_OK;
	}
	if (c->stage == 7) {
		const size_t nfb = c->blocks.size();
		std::vector<i32> fragbase(nfb);
		if (nfb) GSA_CHECK(c, hipMemcpy(fragbase.data(), c->bl_alnlen.as<i32>() + 2 * (size_t)nfb, nfb * 4, hipMemcpyDeviceToHost));      // (bl_alnlen | bl_score | fragbase: one buffer)
		frags_count(c);
		// (stage 7 view: the records exist in the device's working layout only -- gsa_frag -- and are packed here)
		std::vector<gsa_frag> wide((size_t)c->n_frags);
		if (c->n_frags) GSA_CHECK(c, hipMemcpy(wide.data(), c->f_rec.p, (size_t)c->n_frags * sizeof(gsa_frag), hipMemcpyDeviceToHost));
		c->h_frags.resize((size_t)c->n_frags);
		for (size_t i = 0; i < wide.size(); i++) {
			const gsa_frag &w = wide[i]; gsa_rec &o = c->h_frags[i];
			if (w.bseed) { o.seed.qpos = w.qpos; o.seed.len = w.qlen; o.seed.rpos = w.rpos; }
			else { o.gap.nqlen = -1 - w.qlen; o.gap.rlen = w.rlen; o.gap.aln_len = w.aln_len; o.gap.aln_off = (uint32_t)w.aln_off; }
		}
		for (size_t k = 0; k < nfb; k++) {
			gsa_block o; memset(&o, 0, sizeof(o));
			o.score = c->blocks[k].score; o.bdup = c->blocks[k].bdup; o.frag_off = fragbase[k];
			o.n_frag = (i32)((k + 1 < nfb ? (i64)fragbase[k + 1] : c->n_frags) - fragbase[k]);
			c->h_blocks.push_back(o);
		}
		c->frags_stage = 7;
		return GSA_OK;
	}
	// stages 3..6: seeds of the refined arrays, blocks from the host list
	if (!c->have_host_seeds) {
		const size_t nr = (size_t)c->n_r;
		c->h_r_q.resize(nr); c->h_r_len.resize(nr); c->h_r_r.resize(nr);
		if (nr) {
			GSA_CHECK(c, hipMemcpy(c->h_r_q.data(), c->r_q.p, nr * 4, hipMemcpyDeviceToHost));
			GSA_CHECK(c, hipMemcpy(c->h_r_len.data(), c->r_len.p, nr * 4, hipMemcpyDeviceToHost));
			GSA_CHECK(c, hipMemcpy(c->h_r_r.data(), c->r_r.p, nr * 8, hipMemcpyDeviceToHost));
		}
		c->have_host_seeds = true;
	}
	for (size_t k = 0; k < c->blocks.size(); k++) {
		const HostBlock &hb = c->blocks[k];
		gsa_block o; memset(&o, 0, sizeof(o));
		o.score = hb.score; o.bdup = hb.bdup; o.frag_off = (i64)c->h_frags.size();
		const i32 sb = c->h_leaf[hb.leaf_beg].beg, se = c->h_leaf[hb.leaf_end - 1].end;
		o.n_frag = se - sb;
		for (i32 s = sb; s < se; s++) {
			gsa_rec f; f.seed.qpos = c->h_r_q[s]; f.seed.len = c->h_r_len[s]; f.seed.rpos = c->h_r_r[s];
			c->h_frags.push_back(f);
		}
		c->h_blocks.push_back(o);
	}
	c->frags_stage = c->stage;
	return GSA_OK;
}
