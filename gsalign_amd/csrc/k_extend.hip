// gsalign_amd/csrc/k_extend.hip -- stages 7 and 8, device part: gap records
// between consecutive seeds (a11), gap classification and the DP launch (a12),
// materialisation of the gapped strings and per-block (aln_len, score).
//
// Replaces IdentifyNormalPairs / FillAlnBlockGaps, GenerateFragAlignment,
// CheckFragPairMismatch, CountIdenticalPairs and the string surgery of
// ksw2_alignment (reference src/ProcessCandidateAlignment.cpp:38-61,241-276,
// 290-351; src/ksw2_alignment.cpp:264-272).
#include "gsa_ctx.h"
#include "gsa_fm.h"
#include "gsa_scan.h"
#include "gsa_gap.h"

#define TPB 256
#define GID(n) i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; if (i >= (n)) return
#define LAUNCH(k, n, ...) hipLaunchKernelGGL(k, dim3(grid_for((size_t)(n), TPB)), dim3(TPB), 0, st, __VA_ARGS__)
#define ENS(T, buf, n) do { if (!dev_ensure<T>(c, c->buf, (size_t)(n))) return GSA_ERR_NOMEM; } while (0)
#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)


__device__ __forceinline__ i32 find_block(const i32 *__restrict__ base, i32 nfb, i64 slot)
{
	i32 lo = 0, hi = nfb;                      // last k with base[k] <= slot
	while (hi - lo > 1) { i32 m = (lo + hi) >> 1; if (base[m] <= slot) lo = m; else hi = m; }
	return lo;
}

// (each struct below is one fused pass: value -> exclusive scan -> emit, see gsa_scan.h)
// per seed slot: the seed record and, if a gap follows (IdentifyNormalPairs :241-265), the gap record,
// classified (GenerateFragAlignment :311-342)
// STEPS > 0 (round 5): fewer than 2^STEPS blocks -- the slot's block is found by exactly STEPS probes without a branch and every load of load() is unconditional, so
// the loads of a thread's elements go out together (gsa_scan.h, clamped Ops); STEPS = 0: any number of blocks, the loop form.
template <int STEPS>
struct OpSlotsT {
	static constexpr bool clamped = STEPS > 0;
	i32 nfb; const i32 *seedbase, *sbeg, *q, *len; const i64 *r; const i32 *e_id, *r_orig;
	gsa_frag *frag; i32 *ftype, *fmism, *fragbase, *fearly, *mail;
	__device__ void slot(i64 i, i32 &k, i32 &s, i32 &n) const
	{
		k = find_block(seedbase, nfb, i);
		s = sbeg[k] + (i32)(i - seedbase[k]);
		n = 1;
		if (i + 1 != seedbase[k + 1]) {
			const i32 qg = q[s + 1] - (q[s] + len[s]); const i64 rg = r[s + 1] - (r[s] + len[s]);
			if (qg > 0 || rg > 0) n = 2;
		}
	}
	struct Item { i32 k, first, n, q, len, qn, early; i64 r, rn; i32 last; };      // first: the slot opens block k; n = 2: a gap record follows; qn / rn: the next seed's start; early: its early DP job
	__device__ Item load(i64 i) const
	{
		Item it; i32 s;
		if constexpr (STEPS > 0) {
			i32 lo = 0, hi = nfb;                      // last k with seedbase[k] <= i
#pragma unroll
			for (int st = 0; st < STEPS; st++) { const i32 m = (lo + hi) >> 1; const bool act = hi - lo > 1, le = seedbase[m] <= i; lo = (act && le) ? m : lo; hi = (act && !le) ? m : hi; }
			it.k = lo;
			const i32 sb = seedbase[lo], sb1 = seedbase[lo + 1];
			s = sbeg[lo] + (i32)(i - sb);
			it.first = i == sb ? 1 : 0; it.last = i + 1 == sb1 ? 1 : 0;
			const i32 sp = it.last ? s : s + 1;      // (the block's last seed has no successor in the block: its own values are read and not used)
			it.q = q[s]; it.len = len[s]; it.r = r[s]; it.qn = q[sp]; it.rn = r[sp]; it.early = e_id[r_orig[s]]; it.n = 1;
		} else {
			slot(i, it.k, s, it.n);
			it.first = i == seedbase[it.k] ? 1 : 0; it.last = 0;
			it.q = q[s]; it.len = len[s]; it.r = r[s]; it.qn = 0; it.rn = 0; it.early = -1;
			if (it.n == 2) { it.qn = q[s + 1]; it.rn = r[s + 1]; it.early = e_id[r_orig[s]]; }
		}
		return it;
	}
	__device__ void prep(Item &it, i64) const
	{
		if constexpr (STEPS > 0) {
			it.n = 1;
			if (!it.last) { const i32 qg = it.qn - (it.q + it.len); const i64 rg = it.rn - (it.r + it.len); if (qg > 0 || rg > 0) it.n = 2; }
			if (it.n != 2) { it.qn = 0; it.rn = 0; it.early = -1; }
		}
	}
	__device__ i32 value(const Item &it, i64, int) const { return it.n; }
	__device__ void emit(const Item &it, i64, const i32 *v, const i32 *ex) const
	{
		const i32 k = it.k, p = ex[0];
		if (it.first) { fragbase[k] = p; fragbase[k - 2 * (i64)nfb] = 0; fragbase[k - (i64)nfb] = 0; }      // (+ the block's aln_len / score sums start at 0: bl_alnlen[nfb] | bl_score[nfb] | fragbase[] is one buffer)
		gsa_frag f; f.bseed = 1; f.qpos = it.q; f.qlen = it.len; f.rlen = it.len; f.rpos = it.r; f.aln_off = 0; f.aln_len = 0; f._pad = 0;
		frag[p] = f; ftype[p] = FT_SEED; fmism[p] = 0; fearly[p] = -1;
		if (v[0] == 2) {
			i32 qg = it.qn - (it.q + it.len); if (qg < 0) qg = 0;
			i64 rg64 = it.rn - (it.r + it.len); i32 rg = rg64 < 0 ? 0 : (i32)rg64;
			gsa_frag g; g.bseed = 0; g.qpos = it.q + it.len; g.rpos = it.r + it.len; g.qlen = qg; g.rlen = rg; g.aln_off = 0; g.aln_len = 0; g._pad = 0;
			// (class, mismatch count and the link to an early DP launch: k_gap_class, one thread per record)
			frag[p + 1] = g; ftype[p + 1] = FT_DP; fmism[p + 1] = 0;
			fearly[p + 1] = it.early;
		}
	}
	__device__ void done(const i32 *t) const { mail[M_NF] = t[0]; }
};

// class of every gap record (GenerateFragAlignment :311-342) -- a kernel of its own, one thread per record: the
// mismatch count of an equal-length gap is a serial loop over its bases, too heavy for a thread of a fused pass
// -- and the link to a large DP gap stage 2 launched early: the result counts only if this is exactly the gap
// it listed.
__global__ void k_gap_class(i64 ub, const i32 *__restrict__ mail, const gsa_frag *__restrict__ frag, const uint8_t *__restrict__ query, const uint8_t *__restrict__ ref,
                            const i32 *__restrict__ e_list, const i64 *__restrict__ e_off1, const i64 *__restrict__ e_off2, i32 *ftype, i32 *fmism, i32 *fearly, i32 *e_rec)
{
	GID(ub);
	if (i >= mail[M_NF] || ftype[i] == FT_SEED) return;
	const gsa_frag g = frag[i];
	i32 mism; const i32 t = classify_gap(query, ref, g.qpos, g.rpos, g.qlen, g.rlen, mism);
	ftype[i] = t; fmism[i] = mism;
	i32 e = fearly[i];
	if (e >= 0 && !(t == FT_DP && e_off2[e] == g.qpos && e_off1[e] == g.rpos && e_list[3 * e + 1] == g.rlen && e_list[3 * e + 2] == g.qlen)) e = -1;
	fearly[i] = e; if (e >= 0) e_rec[e] = (i32)i;
}

// DP job list and string offsets in one pass.  Component 0 counts the DP jobs.  Component 1 gives every
// gap the room it can need AT MOST (a DP gap m+n, the others their exact length): the offsets of the
// gapped strings, and of a DP job's op string, then do not depend on the DP results, and everything
// that is not a large DP job can be written while the striped kernel still runs.
struct OpDpJobs {
	const i32 *ftype, *fearly; gsa_frag *frag; gsa_rec *rec16;
	i32 *jfrag; i64 *off1; i32 *len1; i64 *off2; i32 *len2; i64 *opsoff; i32 *fjob, *alen; i64 *aoff; i32 *mail;
	// (not a clamped Op: twelve 48-byte Items per thread are 156 VGPRs as it is; with the loads of four elements pinned together the kernel needs 324)
	struct Item { gsa_frag f; i32 t, early; };      // t = -1: behind the last record
	__device__ Item load(i64 i) const
	{
		Item it; it.t = -1; it.early = -1;
		if (i >= mail[M_NF]) return it;
		it.t = ftype[i]; it.f = frag[i]; it.early = fearly[i];
		return it;
	}
	__device__ i32 value(const Item &it, i64, int c) const
	{
		if (it.t < 0) return 0;
		if (c == 0) return (it.t == FT_DP && it.early < 0) ? 1 : 0;      // (early jobs are already running)
		if (it.t == FT_DEL) return it.f.rlen;
		if (it.t == FT_INS || it.t == FT_EQ) return it.f.qlen;
		if (it.t == FT_DP) return it.f.rlen + it.f.qlen;
		return 0;
	}
	__device__ void emit(const Item &it, i64 i, const i32 *v, const i32 *ex) const
	{
		if (it.t < 0) return;
		alen[i] = v[1]; aoff[i] = ex[1];                      // (alen of a DP gap: replaced by the op count)
		// the record's own string fields: final here for everything but a DP gap (its length comes from the DP kernel
		// of its size class, or from the host's patch list for the striped ones), so the records can leave early
		const i32 al = it.t == FT_DP ? 0 : v[1];
		if (it.t != FT_SEED) { frag[i].aln_off = ex[1]; frag[i].aln_len = al; }
		{	// the 16-byte record that travels (gsa_rec, gsa_hip.h): a seed as it is, a gap without its positions
			gsa_rec r;
			if (it.t == FT_SEED) { r.seed.qpos = it.f.qpos; r.seed.len = it.f.qlen; r.seed.rpos = it.f.rpos; }
			else { r.gap.nqlen = -1 - it.f.qlen; r.gap.rlen = it.f.rlen; r.gap.aln_len = al; r.gap.aln_off = (u32)ex[1]; }
			rec16[i] = r;
		}
		if (!v[0]) { fjob[i] = (it.t == FT_DP) ? -2 - it.early : -1; return; }      // <= -2: early job -2 - fjob
		const i32 j = ex[0];
		jfrag[j] = (i32)i; off1[j] = it.f.rpos; len1[j] = it.f.rlen; off2[j] = it.f.qpos; len2[j] = it.f.qlen; opsoff[j] = ex[1];
		fjob[i] = j;
	}
	__device__ void done(const i32 *t) const
	{
		mail[M_NJOB] = t[0]; mail[M_NALN] = t[1];
		for (int k = 0; k < 8; k++) mail[M_DPERR + k] = 0;      // the DP kernels' error words and cell counters start clean (saves a fill operation in front of them)
	}
};

// one DP record written by a whole 256-thread workgroup: 256 positions per pass, prefix counts of the
// consumed bases across the workgroup.  Returns the score (identical pairs) in every thread.
__device__ i32 write_dp_record_wg(i64 i, i32 L, const uint8_t *__restrict__ op, const i64 *__restrict__ aoff,
                                  const uint8_t *__restrict__ query, const uint8_t *__restrict__ ref, gsa_frag *frag, uint8_t *aln1, uint8_t *aln2,
                                  int *s_w1, int *s_w2, int *s_sc)
{
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const gsa_frag f = frag[i];
	const i64 o = aoff[i];
	const uint8_t *qs = query + f.qpos, *rs = ref + f.rpos;
	i32 i1 = 0, i2 = 0, score = 0;
	if (tid == 0) *s_sc = 0;
	for (i32 base = 0; base < L; base += 256) {
		const i32 p = base + tid;
		const uint8_t ch = p < L ? op[p] : 0;
		const int c1 = (ch == 'M' || ch == 'I') ? 1 : 0, c2 = (ch == 'M' || ch == 'D') ? 1 : 0;
		int s1 = c1, s2 = c2;
		for (int d = 1; d < 64; d <<= 1) { int a = __shfl_up(s1, d), b = __shfl_up(s2, d); if (lane >= d) { s1 += a; s2 += b; } }
		__syncthreads();
		if (lane == 63) { s_w1[wv] = s1; s_w2[wv] = s2; }
		__syncthreads();
		int w1 = 0, w2 = 0, t1 = 0, t2 = 0;
		for (int w = 0; w < 4; w++) { if (w < wv) { w1 += s_w1[w]; w2 += s_w2[w]; } t1 += s_w1[w]; t2 += s_w2[w]; }
		if (p < L) {
			// ops are forward M/D/I; 'D' puts '-' into aln1, 'I' into aln2 (ksw2_alignment.cpp:264-272)
			const uint8_t a1 = c1 ? rs[i1 + w1 + s1 - 1] : '-', a2 = c2 ? qs[i2 + w2 + s2 - 1] : '-';
			aln1[o + p] = a1; aln2[o + p] = a2;
			score += (gsa_nt4(a1) == gsa_nt4(a2));             // CountIdenticalPairs (:38-47)
		}
		i1 += t1; i2 += t2;
	}
	for (int d = 32; d; d >>= 1) score += __shfl_xor(score, d);
	__syncthreads();
	if (lane == 0 && score) atomicAdd(s_sc, score);
	__syncthreads();
	const i32 total = *s_sc;
	if (tid == 0) { frag[i].aln_off = o; frag[i].aln_len = L; }
	__syncthreads();
	return total;
}

// Gapped strings and the per-block sums of the records' (aln_len, score) contributions.  MAT_WGS workgroups, each over one
// contiguous range of the records in tiles of 256: a thread settles its own record when it is a seed or a short gap (the bulk:
// median gap 11 bases); longer gaps are queued in LDS and written by whole wavefronts.  The sums stay in registers while the
// tiles lie in one block and are added with one atomic pair when the block changes (a contig has a handful of blocks); a tile
// that straddles a block edge adds per record.  (Round 2 stored the contributions per record and summed them in a second
// kernel, k_block_reduce: 8 bytes written and read per record and 0.8 ms on the tail of every human-sized contig.)
#define MAT_SERIAL 32
#define MAT_WGS 2048
__global__ void __launch_bounds__(256) k_materialize(i32 nfb, const i32 *__restrict__ nf_ptr, const i32 *__restrict__ fragbase, const i32 *__restrict__ ftype, const i32 *__restrict__ fmism, const i32 *__restrict__ fjob,
                                                      const i32 *__restrict__ jlarge, const i32 *__restrict__ nops, const i32 *__restrict__ alen, const i64 *__restrict__ aoff, const uint8_t *__restrict__ ops,
                                                      const i64 *__restrict__ opsoff, const uint8_t *__restrict__ query, const uint8_t *__restrict__ ref,
                                                      gsa_frag *frag, uint8_t *aln1, uint8_t *aln2, i32 *bl_len, i32 *bl_score)
{
	__shared__ i32 s_list[256];
	__shared__ int s_n;
	__shared__ i32 s_len[4], s_sc[4];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const i64 nf = nf_ptr[0];
	const i64 tiles = (nf + 255) / 256, per = (tiles + gridDim.x - 1) / gridDim.x;
	const i64 t_beg = (i64)blockIdx.x * per, t_end = t_beg + per < tiles ? t_beg + per : tiles;
	i32 cur = -1, l_acc = 0, sc_acc = 0;             // the block the running sums belong to (uniform); per-thread partial sums
	auto flush = [&]() {
		if (cur < 0) return;
		for (int o = 32; o; o >>= 1) { l_acc += __shfl_xor(l_acc, o); sc_acc += __shfl_xor(sc_acc, o); }
		if (lane == 0) { s_len[wv] = l_acc; s_sc[wv] = sc_acc; }
		__syncthreads();
		if (tid == 0) { const i32 a = s_len[0] + s_len[1] + s_len[2] + s_len[3], b = s_sc[0] + s_sc[1] + s_sc[2] + s_sc[3]; if (a) atomicAdd(&bl_len[cur], a); if (b) atomicAdd(&bl_score[cur], b); }
		__syncthreads();
		l_acc = 0; sc_acc = 0;
	};
	for (i64 tile = t_beg; tile < t_end; tile++) {
		const i64 w0 = tile * 256, w1 = w0 + 256 < nf ? w0 + 256 : nf;
		const i32 kf = find_block(fragbase, nfb, w0), kl = find_block(fragbase, nfb, w1 - 1);
		const bool straddle = kf != kl;
		if (!straddle && kf != cur) { flush(); cur = kf; }
		// a record's contribution: into the running sums, or -- on a tile with a block edge inside -- straight to its block
		auto add = [&](i64 i, i32 L, i32 score) {
			if (!straddle) { l_acc += L; sc_acc += score; }
			else { const i32 b = find_block(fragbase, nfb, i); if (L) atomicAdd(&bl_len[b], L); if (score) atomicAdd(&bl_score[b], score); }
		};
		if (tid == 0) s_n = 0;
		__syncthreads();
		{
			const i64 i = w0 + tid;
			if (i < nf) {
				const i32 t = ftype[i];
				const i32 fj = fjob[i];
				const i32 Lr = t == FT_DP ? ((fj < 0 || jlarge[fj]) ? -1 : nops[fj]) : alen[i];       // -1: a large DP job, written after the striped kernel
				if (t == FT_SEED) { const i32 l = frag[i].qlen; add(i, l, l); }
				else if (Lr < 0) { }
				else if (Lr > MAT_SERIAL) s_list[atomicAdd(&s_n, 1)] = tid;
				else {
					const gsa_frag f = frag[i];
					const i64 o = aoff[i]; const i32 L = Lr;
					const uint8_t *qs = query + f.qpos, *rs = ref + f.rpos;
					i32 score = 0;
					// All the loads of a record first, then its stores: the string pools are byte pointers like the sources, so a store in
					// between holds every later load back until it is through -- 32 dependent round trips per record where two will do.
					uint8_t a1[MAT_SERIAL], a2[MAT_SERIAL];
					if (t == FT_DP) {
						// ops are forward M/D/I; 'D' puts '-' into aln1, 'I' into aln2 (ksw2_alignment.cpp:264-272)
						const uint8_t *op = ops + opsoff[fj];
						uint8_t ch[MAT_SERIAL];
#pragma unroll
						for (i32 p = 0; p < MAT_SERIAL; p++) ch[p] = p < L ? op[p] : (uint8_t)0;
						i32 i1 = 0, i2 = 0;
#pragma unroll
						for (i32 p = 0; p < MAT_SERIAL; p++) {
							const bool c1 = ch[p] == 'M' || ch[p] == 'I', c2 = ch[p] == 'M' || ch[p] == 'D';
							a1[p] = c1 ? rs[i1] : (uint8_t)'-'; a2[p] = c2 ? qs[i2] : (uint8_t)'-';
							i1 += c1; i2 += c2;
						}
#pragma unroll
						for (i32 p = 0; p < MAT_SERIAL; p++) if (p < L) score += (gsa_nt4(a1[p]) == gsa_nt4(a2[p]));             // CountIdenticalPairs (:38-47)
					} else {
#pragma unroll
						for (i32 p = 0; p < MAT_SERIAL; p++) {
							a1[p] = (t != FT_INS && p < L) ? rs[p] : (uint8_t)'-';
							a2[p] = (t != FT_DEL && p < L) ? qs[p] : (uint8_t)'-';
						}
						if (t == FT_EQ) score = f.qlen - fmism[i];
					}
#pragma unroll
					for (i32 p = 0; p < MAT_SERIAL; p++) if (p < L) { aln1[o + p] = a1[p]; aln2[o + p] = a2[p]; }
					add(i, L, score);
				}
			}
		}
		__syncthreads();
		const int nlist = s_n;
		for (int g = wv; g < nlist; g += 4) {
			const i64 i = w0 + s_list[g];
			const i32 t = ftype[i];
			const gsa_frag f = frag[i];
			const i64 o = aoff[i]; const i32 L = t == FT_DP ? nops[fjob[i]] : alen[i];
			const uint8_t *qs = query + f.qpos, *rs = ref + f.rpos;
			i32 score = 0;
			if (t == FT_DEL) { for (i32 p = lane; p < L; p += 64) { aln1[o + p] = rs[p]; aln2[o + p] = '-'; } }
			else if (t == FT_INS) { for (i32 p = lane; p < L; p += 64) { aln1[o + p] = '-'; aln2[o + p] = qs[p]; } }
			else if (t == FT_EQ) { for (i32 p = lane; p < L; p += 64) { aln1[o + p] = rs[p]; aln2[o + p] = qs[p]; } score = f.qlen - fmism[i]; }
			else {
				const uint8_t *op = ops + opsoff[fjob[i]];
				i32 i1 = 0, i2 = 0;                      // consumed bases of the reference / query fragment so far
				for (i32 base = 0; base < L; base += 64) {
					const i32 p = base + lane;
					const uint8_t ch = p < L ? op[p] : 0;
					const int c1 = (ch == 'M' || ch == 'I') ? 1 : 0, c2 = (ch == 'M' || ch == 'D') ? 1 : 0;
					int s1 = c1, s2 = c2;                 // inclusive wave prefix sums
					for (int d = 1; d < 64; d <<= 1) { int a = __shfl_up(s1, d), b = __shfl_up(s2, d); if (lane >= d) { s1 += a; s2 += b; } }
					if (p < L) {
						const uint8_t a1 = c1 ? rs[i1 + s1 - 1] : '-', a2 = c2 ? qs[i2 + s2 - 1] : '-';
						aln1[o + p] = a1; aln2[o + p] = a2;
						score += (gsa_nt4(a1) == gsa_nt4(a2));
					}
					i1 += __shfl(s1, 63); i2 += __shfl(s2, 63);
				}
				for (int d = 32; d; d >>= 1) score += __shfl_xor(score, d);
			}
			if (lane == 0) add(i, L, score);
		}
		__syncthreads();      // (s_list / s_n are reused by the next tile)
	}
	flush();
}

// The records of the large DP jobs, after the striped kernel: one workgroup per job; (record, aln_len,
// score) go to a patch list the host applies to the records and block sums it already holds.
// Two sources: jobs launched early from the leaf table (rec_of = e_rec, -1 = their leaf was dropped) and
// whatever large job only turned up in the job list (lg != nullptr: job = lg[3g], record = jfrag[job]).
__global__ void __launch_bounds__(256) k_materialize_large(i32 nlarge, const i32 *__restrict__ lg, const i32 *__restrict__ rec_of, const i32 *__restrict__ nops,
                                                            const i64 *__restrict__ aoff, const uint8_t *__restrict__ ops, const i64 *__restrict__ opsoff,
                                                            const uint8_t *__restrict__ query, const uint8_t *__restrict__ ref, gsa_frag *frag, uint8_t *aln1, uint8_t *aln2, i32 *patch,
                                                            const i32 *mail_src, i32 *mail_dst)
{
	__shared__ int s_w1[4], s_w2[4], s_sc;
	const int g = blockIdx.x;
	if (g >= nlarge) return;
	// the last launch of a contig also puts the (by now final) mailbox in front of the patch list: one copy takes all of it home
	if (mail_dst && g == 0 && threadIdx.x < MAIL_N) mail_dst[threadIdx.x] = mail_src[threadIdx.x];
	const i32 job = lg ? lg[3 * g] : g;
	const i64 i = rec_of[job];
	if (i < 0) { if (threadIdx.x == 0) { patch[3 * g] = -1; patch[3 * g + 1] = 0; patch[3 * g + 2] = 0; } return; }
	const i32 L = nops[job];
	const i32 sc = write_dp_record_wg(i, L, ops + opsoff[job], aoff, query, ref, frag, aln1, aln2, s_w1, s_w2, &s_sc);
	if (threadIdx.x == 0) { patch[3 * g] = (i32)i; patch[3 * g + 1] = L; patch[3 * g + 2] = sc; }
}

// A bundle of contigs (Bundle, gsa_internal.h): the records as they travel carry query positions and string offsets relative to
// THEIR contig (what gsa_align_contig of that contig alone returns); the device works on positions in the concatenation and on
// one string pool.  One thread per record, on the records' way out; the pool offset of every contig goes to pinned memory.
__global__ void __launch_bounds__(256) k_bundle_rebase(i64 ub, const i32 *__restrict__ nf_ptr, i32 nfb, const i32 *__restrict__ fragbase, const i32 *__restrict__ bcontig, const i32 *__restrict__ blk0,
                                                        const i32 *__restrict__ off, const i64 *__restrict__ aoff, gsa_rec *rec16, i64 *h_a0)
{
	GID(ub);
	if (i >= nf_ptr[0]) return;
	const i32 k = find_block(fragbase, nfb, i), ci = bcontig[k];
	const i64 f0 = fragbase[blk0[ci]], a0 = aoff[f0];
	gsa_rec r = rec16[i];
	if (r.seed.qpos >= 0) r.seed.qpos -= off[ci]; else r.gap.aln_off -= (u32)a0;
	rec16[i] = r;
	if (i == f0) h_a0[ci] = a0;
}

i64 frags_count(gsa_ctx *c)
{
	if (c->n_frags < 0) {
		i32 nf = 0;
		hipStreamSynchronize(c->stream);
		if (hipMemcpy(&nf, c->d_mail.as<i32>() + M_NF, 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
		c->n_frags = nf;
	}
	return c->n_frags;
}

// stage 7: build the records of the final block list (c->blocks), S6.  No read-back: the record
// count stays in the mailbox, the host only knows upper bounds (two records per seed; gap bases <=
// the blocks' spans).
int stage7_fill(gsa_ctx *c)
{
	hipStream_t st = c->stream;
	c->n_frags = 0; c->n_aln = 0; c->nf_ub = 0; c->span_ub = 0;
	const i32 nfb = (i32)c->blocks.size();
	if (nfb == 0) return GSA_OK;
	if (!pin_ensure<i32>(c, c->p_blk, (size_t)3 * (nfb + 1) + MAIL_N + 8)) return GSA_ERR_NOMEM;       // staging (the stage-8 sums land here later)
	i32 *seedbase = c->p_blk.as<i32>(), *sbeg = seedbase + nfb + 1;
	i64 ns = 0, span = 0;
	for (i32 k = 0; k < nfb; k++) {
		const HostBlock &hb = c->blocks[k];
		const Leaf &lf = c->h_leaf[hb.leaf_beg], &ll = c->h_leaf[hb.leaf_end - 1];
		sbeg[k] = lf.beg; seedbase[k] = (i32)ns;
		ns += ll.end - sbeg[k];
		span += (i64)(ll.q_last_end - lf.q_first) + (ll.r_last_end - lf.r_first);
	}
	seedbase[nfb] = (i32)ns;
	if (span >= (1ll << 31) - 4096 || 2 * ns >= (1ll << 31) - 4096) return gsa_fail(c, GSA_ERR_LIMIT, "contig too large for 32-bit record / gap offsets");
	c->nf_ub = 2 * ns; c->span_ub = span;
	// (one buffer, one copy: seedbase[nfb + 1] | sbeg[nfb], as they sit in the pinned staging area.  Every GPU operation of a
	//  contig costs the command processor the same few microseconds whatever it does: a 5 Mb contig is ~65 of them and three
	//  contigs in flight are bound by exactly that)
	ENS(i32, fb_seedbase, 2 * (size_t)nfb + 2); ENS(i32, bl_alnlen, 3 * (size_t)nfb + 3);
	GSA_CHECK(c, hipMemcpyAsync(c->fb_seedbase.p, seedbase, (size_t)(2 * nfb + 1) * 4, hipMemcpyHostToDevice, st));
	if (c->bnd.n) {
		// contig of every final block | first block of every contig: what k_bundle_rebase needs on the records' way out
		const size_t nt = (size_t)nfb + (size_t)c->bnd.n + 1;
		if (!pin_ensure<i32>(c, c->p_bblk, nt) || !pin_ensure<i64>(c, c->p_ba0, (size_t)c->bnd.n + 1)) return GSA_ERR_NOMEM;
		ENS(i32, d_bblk, nt);
		i32 *t = c->p_bblk.as<i32>();
		for (int k = 0; k < c->bnd.n; k++) for (i32 b = c->b_blk0[(size_t)k]; b < c->b_blk0[(size_t)k + 1]; b++) t[b] = k;
		for (int k = 0; k <= c->bnd.n; k++) t[(size_t)nfb + (size_t)k] = c->b_blk0[(size_t)k] < nfb ? c->b_blk0[(size_t)k] : nfb - 1;
		GSA_CHECK(c, hipMemcpyAsync(c->d_bblk.p, t, nt * 4, hipMemcpyHostToDevice, st));
	}
	i32 *d_sbeg = c->fb_seedbase.as<i32>() + nfb + 1, *d_fragbase = c->bl_alnlen.as<i32>() + 2 * (size_t)nfb;      // bl_alnlen[nfb] | bl_score[nfb] | fragbase[nfb + 1]
	const i64 nfu = c->nf_ub;
	ENS(gsa_frag, f_rec, nfu + 1); ENS(i32, f_type, nfu + 1); ENS(i32, f_mism, nfu + 1); ENS(i32, f_score, nfu + 1); ENS(i32, f_job, nfu + 1); ENS(i32, f_alnlen, nfu + 1);
	ENS(i32, f_early, nfu + 2);
	// (e_rec[] of the early jobs is -1 since the pass that listed them, OpEarlyGaps)
#define GSA_SLOTS_ARGS { nfb, c->fb_seedbase.as<i32>(), d_sbeg, c->r_q.as<i32>(), c->r_len.as<i32>(), c->r_r.as<i64>(), c->e_id.as<i32>(), c->r_orig.as<i32>(), \
	               c->f_rec.as<gsa_frag>(), c->f_type.as<i32>(), c->f_mism.as<i32>(), d_fragbase, c->f_early.as<i32>(), c->d_mail.as<i32>() }
	if (nfb < 256) { OpSlotsT<8> op = GSA_SLOTS_ARGS; RC((lb_launch<1>(c, ns, op))); }
	else { OpSlotsT<0> op = GSA_SLOTS_ARGS; RC((lb_launch<1>(c, ns, op))); }
#undef GSA_SLOTS_ARGS
	LAUNCH(k_gap_class, nfu, nfu, c->d_mail.as<i32>(), c->f_rec.as<gsa_frag>(), c->q_dev, c->di.ref, c->e_list.as<i32>(), c->e_off1.as<i64>(), c->e_off2.as<i64>(),
	       c->f_type.as<i32>(), c->f_mism.as<i32>(), c->f_early.as<i32>(), c->e_rec.as<i32>());
	c->n_frags = -1;
	return GSA_OK;
}

// stage 8: S7 on the records built by stage7_fill.  Stream plan: the striped DP kernel (the latency
// floor of a contig) runs on the main stream; behind the small-job kernel on stream_aux[1] everything
// that does not depend on a large job is finished and sent to the host meanwhile -- strings, record
// sums, per-block sums, the records themselves.  After the striped kernel only the large jobs' strings,
// a patch list and the string pools remain.
int stage78_extend(gsa_ctx *c)
{
	hipStream_t st = c->stream, sx = c->stream_aux[1];
	const i32 nfb = (i32)c->blocks.size(); const i64 nfu = c->nf_ub;
	c->n_aln = 0; c->n_large = 0; c->n_jobs = 0;
	if (nfb == 0 || nfu == 0) { c->n_frags = 0; return GSA_OK; }
	if (c->profiling) hipEventRecord(c->ev[8], st);
	i32 *mail = c->d_mail.as<i32>();
	const i64 nju = nfu / 2 + 1;                 // at most one gap record per seed
	ENS(i32, j_frag, nju + 1); ENS(i64, j_opsoff, nju + 2); ENS(i32, j_nops, nju + 1);
	ENS(i64, w_best, nju + 1); ENS(i64, w_sum, nju + 1); ENS(i32, a_uniq, nju + 1); ENS(i32, a_cu, nju + 1);
	i64 *off1 = c->w_best.as<i64>(), *off2 = c->w_sum.as<i64>(); i32 *len1 = c->a_uniq.as<i32>(), *len2 = c->a_cu.as<i32>();
	ENS(i64, d_alnoff, nfu + 2);
	ENS(gsa_rec, f_rec16, nfu + 1);
	{ OpDpJobs op = { c->f_type.as<i32>(), c->f_early.as<i32>(), c->f_rec.as<gsa_frag>(), c->f_rec16.as<gsa_rec>(), c->j_frag.as<i32>(), off1, len1, off2, len2, c->j_opsoff.as<i64>(), c->f_job.as<i32>(),
	                  c->f_alnlen.as<i32>(), c->d_alnoff.as<i64>(), mail }; RC((lb_launch<2, 12>(c, nfu, op))); }
	// the records are final here except for the string length of a DP gap: they leave now (all nf_ub of them: the count is
	// still on the device), on a third stream; the DP gaps' lengths follow as a short list the host patches in
	hipStream_t sc = c->stream_aux[2];
	if (!pin_ensure<gsa_rec>(c, c->p_frags, (size_t)nfu + 1)) return GSA_ERR_NOMEM;
	GSA_CHECK(c, hipEventRecord(c->ev[19], st)); GSA_CHECK(c, hipStreamWaitEvent(sc, c->ev[19], 0));
	ENS(uint8_t, d_ops, c->span_ub + 64);
	i32 *d_blen = c->bl_alnlen.as<i32>(), *d_bscore = d_blen + nfb, *d_fragbase = d_blen + 2 * (size_t)nfb;      // (one buffer since stage 7: one copy home)
	Ksw2Launch kl;
	RC(run_ksw2_jobs(c, (i32)nju, c->di.ref, off1, len1, c->q_dev, off2, len2, c->d_ops.as<uint8_t>(), c->j_opsoff.as<i64>(), c->j_nops.as<i32>(), c->span_ub, &kl,
	                 nullptr, nullptr, true));
	// (run_ksw2_jobs read the mailbox: the record count and the size of the string pools are known now)
	// The records leave only now, behind the host's look at the size classes: a 166 MB copy (a 250 Mb contig) in flight keeps
	// the link busy for 3 ms, and the few bytes the classification pass stores into pinned memory for that look queued
	// behind it -- the small-DP kernels started 3 ms late.  (Since round 3 the records travel as 16-byte gsa_rec: 66 MB.)
	if (c->bnd.n)
		hipLaunchKernelGGL(k_bundle_rebase, dim3(grid_for((size_t)nfu, TPB)), dim3(TPB), 0, sc, nfu, mail + M_NF, nfb, d_fragbase, c->d_bblk.as<i32>(), c->d_bblk.as<i32>() + nfb,
		                   c->bnd.off, c->d_alnoff.as<i64>(), c->f_rec16.as<gsa_rec>(), c->p_ba0.as<i64>());
	const i32 *hm = c->p_dp.as<i32>();      // (the mailbox as run_ksw2_jobs read it: the record count is final since stage 7)
	c->n_frags = hm[M_NF]; c->n_aln = hm[M_NALN]; c->n_large = kl.nlarge; c->dbg[7] = (u64)kl.nlarge;
#ifdef GSA_EXPERIMENTS      // (what the results' way home costs: GSA_SKIP_D2H=1 leaves the string pools on the device, 2 the records too -- results are garbage, timing only)
	static const int skip_d2h = [] { const char *e = getenv("GSA_SKIP_D2H"); return e ? atoi(e) : 0; }();
#else
	const int skip_d2h = 0;
#endif
	GSA_CHECK(c, hipMemcpyAsync(c->p_frags.p, c->f_rec16.p, skip_d2h >= 2 ? 16 : (size_t)c->n_frags * sizeof(gsa_rec), hipMemcpyDeviceToHost, sc));
	// everything that can only leave at the very end sits in ONE buffer: final mailbox | patch list of the large DP jobs |
	// string pool 1 | string pool 2 -- a single copy behind the last kernel instead of a chain of four
	const size_t npatch = (size_t)kl.nlarge + (size_t)c->n_early;
	const size_t t_patch = MAIL_N * sizeof(i32), t_aln1 = (t_patch + 12 * npatch + 255) & ~(size_t)255, t_aln2 = (t_aln1 + (size_t)c->n_aln + 255) & ~(size_t)255;
	const size_t t_total = t_aln2 + (size_t)c->n_aln;
	c->n_jobs = hm[M_NJOB];
	if (!pin_ensure<i32>(c, c->p_jpatch, 2 * (size_t)c->n_jobs + 2) || !pin_ensure<char>(c, c->p_tail, t_total + 256) || !pin_ensure<i32>(c, c->p_blk, (size_t)3 * (nfb + 1) + 8)) return GSA_ERR_NOMEM;
	ENS(uint8_t, d_tail, t_total + 256);
	uint8_t *d_tail = c->d_tail.as<uint8_t>(), *d_aln1 = d_tail + t_aln1, *d_aln2 = d_tail + t_aln2;
	c->h_tmail = (const i32 *)c->p_tail.p; c->h_tpatch = (const i32 *)((char *)c->p_tail.p + t_patch);
	c->h_taln1 = (char *)c->p_tail.p + t_aln1; c->h_taln2 = (char *)c->p_tail.p + t_aln2;
	// ---- behind the small jobs (stream_aux[1]; when there is no small job it starts at the fork) ----
	if (!kl.small_in_flight) { GSA_CHECK(c, hipEventRecord(c->ev[10], st)); GSA_CHECK(c, hipStreamWaitEvent(sx, c->ev[10], 0)); }
	// (record, string length) of the small DP jobs: the record numbers are known since the job list, the lengths behind the small kernels
	if (c->n_jobs > 0) {
		GSA_CHECK(c, hipMemcpyAsync(c->p_jpatch.p, c->j_frag.p, (size_t)c->n_jobs * 4, hipMemcpyDeviceToHost, sc));
		GSA_CHECK(c, hipStreamWaitEvent(sc, kl.small_in_flight ? c->ev[12] : c->ev[10], 0));
		GSA_CHECK(c, hipMemcpyAsync(c->p_jpatch.as<i32>() + c->n_jobs, c->j_nops.p, (size_t)c->n_jobs * 4, hipMemcpyDeviceToHost, sc));
	}
	GSA_CHECK(c, hipEventRecord(c->ev[15], sc));
	// the string pools go home behind the small gaps' strings too (same stream, idle by then): the large jobs' strings are
	// written straight into the pinned copy later, so nothing is left to copy behind the striped kernel
	const bool pools_early = npatch > 0 && c->n_aln > 0;
	const i32 *jlarge = c->d_dp_large.as<i32>() + 3 * ((size_t)nju + 1);
	// (a gap's room in the pools is what it can need AT MOST; the bytes no record ends up owning are zero, not whatever the buffer held:
	//  results are byte-identical from call to call pool slack included -- tools/stress_consistency.py compares whole pools)
	if (c->n_aln > 0) GSA_CHECK(c, hipMemsetAsync(d_aln1, 0, t_total - t_aln1, sx));
	{ const i64 tiles_ub = (nfu + 255) / 256;
	  hipLaunchKernelGGL(k_materialize, dim3((unsigned)(tiles_ub < MAT_WGS ? tiles_ub : MAT_WGS)), dim3(256), 0, sx, nfb, mail + M_NF, d_fragbase, c->f_type.as<i32>(), c->f_mism.as<i32>(), c->f_job.as<i32>(),
	                     jlarge, c->j_nops.as<i32>(), c->f_alnlen.as<i32>(), c->d_alnoff.as<i64>(), c->d_ops.as<uint8_t>(), c->j_opsoff.as<i64>(), c->q_dev, c->di.ref,
	                     c->f_rec.as<gsa_frag>(), d_aln1, d_aln2, d_blen, d_bscore); }
	GSA_CHECK(c, hipEventRecord(c->ev[17], sx));      // the strings of everything but the large jobs are written
	if (pools_early) {
		GSA_CHECK(c, hipStreamWaitEvent(sc, c->ev[17], 0));
		GSA_CHECK(c, hipMemcpyAsync((char *)c->p_tail.p + t_aln1, d_aln1, skip_d2h ? 256 : t_total - t_aln1, hipMemcpyDeviceToHost, sc));
		GSA_CHECK(c, hipEventRecord(c->ev[23], sc));
	}
	// (per-block sums: left by k_materialize itself; the large jobs' records count as zero there, the host adds them from the patch list)
	if (c->profiling) dp_count_cells(c, (i32)nju, len1, len2, sx);      // (measurement: sum of m*n and m+n over the jobs, read with the final mailbox)
	i32 *h_len = c->p_blk.as<i32>(), *h_score = h_len + nfb, *h_fragbase = h_score + nfb;
	GSA_CHECK(c, hipMemcpyAsync(h_len, d_blen, (size_t)3 * nfb * 4, hipMemcpyDeviceToHost, sx));      // h_len | h_score | h_fragbase
	(void)h_score; (void)h_fragbase;
	GSA_CHECK(c, hipEventRecord(c->ev[13], sx));
	// ---- behind the striped kernels ----
	GSA_CHECK(c, hipStreamWaitEvent(st, c->ev[17], 0));      // the other strings are written
	GSA_CHECK(c, hipStreamWaitEvent(st, c->ev[13], 0));      // per-block sums are on the host
	GSA_CHECK(c, hipStreamWaitEvent(st, c->ev[15], 0));      // the records are on the host
	if (npatch > 0) {
		// the large jobs' records: strings, patch list and the final mailbox are stored straight into pinned memory by the
		// kernels (coalesced rows of 256 bytes): the host only waits for the last kernel
		if (pools_early) GSA_CHECK(c, hipStreamWaitEvent(st, c->ev[23], 0));      // (the pools' copy must not overwrite them)
		uint8_t *h_aln1 = (uint8_t *)c->h_taln1, *h_aln2 = (uint8_t *)c->h_taln2; i32 *h_patch = (i32 *)c->h_tpatch, *h_mail = (i32 *)c->h_tmail;
		if (kl.nlarge > 0)
			hipLaunchKernelGGL(k_materialize_large, dim3((unsigned)kl.nlarge), dim3(256), 0, st, kl.nlarge, c->d_dp_large.as<i32>(), c->j_frag.as<i32>(), c->j_nops.as<i32>(),
			                   c->d_alnoff.as<i64>(), c->d_ops.as<uint8_t>(), c->j_opsoff.as<i64>(), c->q_dev, c->di.ref,
			                   c->f_rec.as<gsa_frag>(), h_aln1, h_aln2, h_patch, mail, c->n_early > 0 ? (i32 *)nullptr : h_mail);
		if (c->n_early > 0) {
			GSA_CHECK(c, hipStreamWaitEvent(st, c->ev[14], 0));      // the early striped launch (stream_aux[0])
			c->early_consumed = true;
			hipLaunchKernelGGL(k_materialize_large, dim3((unsigned)c->n_early), dim3(256), 0, st, c->n_early, (const i32 *)nullptr, c->e_rec.as<i32>(), c->e_nops.as<i32>(),
			                   c->d_alnoff.as<i64>(), c->e_ops.as<uint8_t>(), c->e_opsoff.as<i64>(), c->q_dev, c->di.ref,
			                   c->f_rec.as<gsa_frag>(), h_aln1, h_aln2, h_patch + 3 * (size_t)kl.nlarge, mail, h_mail);
		}
	} else {
		// no large job: the mailbox goes home by itself, the pools (if any) behind it
		GSA_CHECK(c, hipMemcpyAsync(c->p_tail.p, mail, MAIL_N * sizeof(i32), hipMemcpyDeviceToHost, st));
		if (c->n_aln) GSA_CHECK(c, hipMemcpyAsync((char *)c->p_tail.p + t_aln1, d_aln1, skip_d2h ? 256 : t_total - t_aln1, hipMemcpyDeviceToHost, st));
	}
	if (c->profiling) hipEventRecord(c->ev[9], st);
	GSA_CHECK(c, hipGetLastError());
	return GSA_OK;
}
