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

#define TPB 256
#define GID(n) i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; if (i >= (n)) return
#define LAUNCH(k, n, ...) hipLaunchKernelGGL(k, dim3(grid_for((size_t)(n), TPB)), dim3(TPB), 0, st, __VA_ARGS__)
#define ENS(T, buf, n) do { if (!dev_ensure<T>(c, c->buf, (size_t)(n))) return GSA_ERR_NOMEM; } while (0)
#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

enum { FT_SEED = 0, FT_DEL = 1, FT_INS = 2, FT_EQ = 3, FT_DP = 4 };

__device__ __forceinline__ i32 find_block(const i32 *__restrict__ base, i32 nfb, i64 slot)
{
	i32 lo = 0, hi = nfb;                      // last k with base[k] <= slot
	while (hi - lo > 1) { i32 m = (lo + hi) >> 1; if (base[m] <= slot) lo = m; else hi = m; }
	return lo;
}

// per seed slot: does a gap record follow this seed?  (IdentifyNormalPairs :241-265)
__global__ void k_slot_counts(i64 ns, i32 nfb, const i32 *__restrict__ seedbase, const i32 *__restrict__ sbeg, const i32 *__restrict__ q,
                              const i32 *__restrict__ len, const i64 *__restrict__ r, i32 *cnt)
{
	i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > ns) return;
	if (i == ns) { cnt[i] = 0; return; }
	const i32 k = find_block(seedbase, nfb, i);
	const i32 s = sbeg[k] + (i32)(i - seedbase[k]);
	const bool last = (i + 1 == seedbase[k + 1]);
	i32 n = 1;
	if (!last) {
		const i32 qg = q[s + 1] - (q[s] + len[s]); const i64 rg = r[s + 1] - (r[s] + len[s]);
		if (qg > 0 || rg > 0) n = 2;
	}
	cnt[i] = n;
}

// write the records; classify the gap (GenerateFragAlignment :311-342)
__global__ void k_slot_emit(i64 ns, i32 nfb, const i32 *__restrict__ seedbase, const i32 *__restrict__ sbeg, const i32 *__restrict__ q,
                            const i32 *__restrict__ len, const i64 *__restrict__ r, const i32 *__restrict__ cnt, const i32 *__restrict__ pos,
                            const uint8_t *__restrict__ query, const uint8_t *__restrict__ ref, gsa_frag *frag, i32 *ftype, i32 *fmism, i32 *fblock, i32 *fragbase)
{
	GID(ns);
	const i32 k = find_block(seedbase, nfb, i);
	const i32 s = sbeg[k] + (i32)(i - seedbase[k]);
	const i32 p = pos[i];
	if (i == seedbase[k]) fragbase[k] = p;
	gsa_frag f; f.bseed = 1; f.qpos = q[s]; f.qlen = len[s]; f.rlen = len[s]; f.rpos = r[s]; f.aln_off = 0; f.aln_len = 0; f._pad = 0;
	frag[p] = f; ftype[p] = FT_SEED; fmism[p] = 0; fblock[p] = k;
	if (cnt[i] == 2) {
		i32 qg = q[s + 1] - (q[s] + len[s]); if (qg < 0) qg = 0;
		i64 rg64 = r[s + 1] - (r[s] + len[s]); i32 rg = rg64 < 0 ? 0 : (i32)rg64;
		gsa_frag g; g.bseed = 0; g.qpos = q[s] + len[s]; g.rpos = r[s] + len[s]; g.qlen = qg; g.rlen = rg; g.aln_off = 0; g.aln_len = 0; g._pad = 0;
		i32 t, mism = 0;
		if (qg == 0) t = FT_DEL;
		else if (rg == 0) t = FT_INS;
		else {
			t = FT_DP;
			if (qg == rg) {
				// CheckFragPairMismatch: positions where the QUERY is ambiguous are skipped
				const uint8_t *qs = query + g.qpos, *rs = ref + g.rpos;
				for (i32 x = 0; x < qg && mism <= GSA_MAX_MISMATCH; x++) { const int a = gsa_nt4(qs[x]); if (a != 4 && a != gsa_nt4(rs[x])) mism++; }
				if (mism <= GSA_MAX_MISMATCH) t = FT_EQ;
			}
		}
		frag[p + 1] = g; ftype[p + 1] = t; fmism[p + 1] = mism; fblock[p + 1] = k;
	}
}

__global__ void k_dp_flags(i64 nf, const i32 *__restrict__ ftype, i32 *flag)
{
	i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > nf) return;
	flag[i] = (i < nf && ftype[i] == FT_DP) ? 1 : 0;
}

__global__ void k_dp_jobs(i64 nf, const i32 *__restrict__ flag, const i32 *__restrict__ ex, const gsa_frag *__restrict__ frag,
                          i32 *jfrag, i64 *off1, i32 *len1, i64 *off2, i32 *len2, i32 *mn, i32 *fjob)
{
	GID(nf);
	if (!flag[i]) { fjob[i] = -1; return; }
	const i32 j = ex[i];
	jfrag[j] = (i32)i; off1[j] = frag[i].rpos; len1[j] = frag[i].rlen; off2[j] = frag[i].qpos; len2[j] = frag[i].qlen; mn[j] = frag[i].rlen + frag[i].qlen;
	fjob[i] = j;
}

__global__ void k_mn_tail(i32 nj, i32 *mn) { if (blockIdx.x == 0 && threadIdx.x == 0) mn[nj] = 0; }

__global__ void k_aln_len(i64 nf, const i32 *__restrict__ ftype, const gsa_frag *__restrict__ frag, const i32 *__restrict__ fjob, const i32 *__restrict__ nops, i32 *alen)
{
	i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
	if (i > nf) return;
	i32 a = 0;
	if (i < nf) {
		const i32 t = ftype[i];
		if (t == FT_DEL) a = frag[i].rlen; else if (t == FT_INS || t == FT_EQ) a = frag[i].qlen; else if (t == FT_DP) a = nops[fjob[i]];
	}
	alen[i] = a;
}

// one wavefront per record: write aln1/aln2 and the record's (aln_len, score) contribution
__global__ void __launch_bounds__(64) k_materialize(i64 nf, const i32 *__restrict__ ftype, const i32 *__restrict__ fmism, const i32 *__restrict__ fjob,
                                                     const i32 *__restrict__ alen, const i64 *__restrict__ aoff, const uint8_t *__restrict__ ops,
                                                     const i64 *__restrict__ opsoff, const uint8_t *__restrict__ query, const uint8_t *__restrict__ ref,
                                                     gsa_frag *frag, uint8_t *aln1, uint8_t *aln2, i32 *c_len, i32 *c_score)
{
	const i64 i = blockIdx.x;
	if (i >= nf) return;
	const int lane = threadIdx.x;
	const i32 t = ftype[i];
	const gsa_frag f = frag[i];
	const i64 o = aoff[i]; const i32 L = alen[i];
	if (t == FT_SEED) { if (lane == 0) { c_len[i] = f.qlen; c_score[i] = f.qlen; } return; }
	const uint8_t *qs = query + f.qpos, *rs = ref + f.rpos;
	i32 score = 0;
	if (t == FT_DEL) { for (i32 p = lane; p < L; p += 64) { aln1[o + p] = rs[p]; aln2[o + p] = '-'; } }
	else if (t == FT_INS) { for (i32 p = lane; p < L; p += 64) { aln1[o + p] = '-'; aln2[o + p] = qs[p]; } }
	else if (t == FT_EQ) { for (i32 p = lane; p < L; p += 64) { aln1[o + p] = rs[p]; aln2[o + p] = qs[p]; } score = f.qlen - fmism[i]; }
	else {
		// ops are forward M/D/I; 'D' puts '-' into aln1, 'I' into aln2 (ksw2_alignment.cpp:264-272)
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
				score += (gsa_nt4(a1) == gsa_nt4(a2));             // CountIdenticalPairs (:38-47)
			}
			i1 += __shfl(s1, 63); i2 += __shfl(s2, 63);
		}
		for (int d = 32; d; d >>= 1) score += __shfl_xor(score, d);
	}
	if (lane == 0) { c_len[i] = L; c_score[i] = score; frag[i].aln_off = o; frag[i].aln_len = L; }
}

__global__ void k_block_sums(i32 nfb, i64 nf, const i32 *__restrict__ fragbase, const i64 *__restrict__ ps_len, const i64 *__restrict__ ps_score, i32 *bl_len, i32 *bl_score)
{
	GID(nfb);
	const i64 b = fragbase[i], e = (i + 1 < nfb) ? fragbase[i + 1] : nf;
	bl_len[i] = (i32)(ps_len[e] - ps_len[b]); bl_score[i] = (i32)(ps_score[e] - ps_score[b]);
}

// stage 7: build the records of the final block list (c->blocks), S6
int stage7_fill(gsa_ctx *c)
{
	hipStream_t st = c->stream;
	c->n_frags = 0; c->n_aln = 0;
	const i32 nfb = (i32)c->blocks.size();
	if (nfb == 0) return GSA_OK;
	std::vector<i32> seedbase(nfb + 1), sbeg(nfb);
	i64 ns = 0;
	for (i32 k = 0; k < nfb; k++) {
		const HostBlock &hb = c->blocks[k];
		sbeg[k] = c->h_leaf[hb.leaf_beg].beg; seedbase[k] = (i32)ns;
		ns += c->h_leaf[hb.leaf_end - 1].end - sbeg[k];
	}
	seedbase[nfb] = (i32)ns;
	ENS(i32, fb_seedbase, nfb + 1); ENS(i32, fb_sbeg, nfb + 1); ENS(i32, fb_fragbase, nfb + 1);
	GSA_CHECK(c, hipMemcpyAsync(c->fb_seedbase.p, seedbase.data(), (size_t)(nfb + 1) * 4, hipMemcpyHostToDevice, st));
	GSA_CHECK(c, hipMemcpyAsync(c->fb_sbeg.p, sbeg.data(), (size_t)nfb * 4, hipMemcpyHostToDevice, st));
	ENS(i32, d_flag, ns + 1); ENS(i32, d_scan, ns + 1);
	i32 *cnt = c->d_flag.as<i32>(), *pos = c->d_scan.as<i32>();
	LAUNCH(k_slot_counts, ns + 1, ns, nfb, c->fb_seedbase.as<i32>(), c->fb_sbeg.as<i32>(), c->r_q.as<i32>(), c->r_len.as<i32>(), c->r_r.as<i64>(), cnt);
	RC(prim_exscan_i32(c, cnt, pos, (size_t)ns + 1));
	i32 nf32 = 0;
	GSA_CHECK(c, hipMemcpyAsync(&nf32, pos + ns, 4, hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipStreamSynchronize(st));       // also makes the staging vectors safe to drop
	const i64 nf = nf32; c->n_frags = nf;
	ENS(gsa_frag, f_rec, nf + 1); ENS(i32, f_type, nf + 1); ENS(i32, f_mism, nf + 1); ENS(i32, f_score, nf + 1); ENS(i32, f_job, nf + 1); ENS(i32, f_alnlen, nf + 1);
	// f_score doubles as "block of record" until the materialise step overwrites it
	LAUNCH(k_slot_emit, ns, ns, nfb, c->fb_seedbase.as<i32>(), c->fb_sbeg.as<i32>(), c->r_q.as<i32>(), c->r_len.as<i32>(), c->r_r.as<i64>(), cnt, pos,
	       c->d_query.as<uint8_t>(), c->di.ref, c->f_rec.as<gsa_frag>(), c->f_type.as<i32>(), c->f_mism.as<i32>(), c->f_score.as<i32>(), c->fb_fragbase.as<i32>());
	GSA_CHECK(c, hipGetLastError());
	return GSA_OK;
}

// stage 8: S7 on the records built by stage7_fill
int stage78_extend(gsa_ctx *c)
{
	hipStream_t st = c->stream;
	const i32 nfb = (i32)c->blocks.size(); const i64 nf = c->n_frags;
	c->n_aln = 0;
	if (nfb == 0 || nf == 0) return GSA_OK;
	if (c->profiling) hipEventRecord(c->ev[8], st);
	ENS(i32, d_flag, nf + 1); ENS(i32, d_scan, nf + 1);
	i32 *flag = c->d_flag.as<i32>(), *ex = c->d_scan.as<i32>();
	LAUNCH(k_dp_flags, nf + 1, nf, c->f_type.as<i32>(), flag);
	RC(prim_exscan_i32(c, flag, ex, (size_t)nf + 1));
	i32 nj = 0;
	GSA_CHECK(c, hipMemcpyAsync(&nj, ex + nf, 4, hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipStreamSynchronize(st));
	ENS(i32, j_frag, nj + 1); ENS(i64, j_opsoff, nj + 2); ENS(i32, j_nops, nj + 1);
	ENS(i64, w_best, nj + 1); ENS(i64, w_sum, nj + 1); ENS(i32, a_uniq, nj + 1); ENS(i32, a_cu, nj + 1); ENS(i32, a_brk, nj + 2);
	i64 *off1 = c->w_best.as<i64>(), *off2 = c->w_sum.as<i64>(); i32 *len1 = c->a_uniq.as<i32>(), *len2 = c->a_cu.as<i32>(), *mn = c->a_brk.as<i32>();
	LAUNCH(k_dp_jobs, nf, nf, flag, ex, c->f_rec.as<gsa_frag>(), c->j_frag.as<i32>(), off1, len1, off2, len2, mn, c->f_job.as<i32>());
	i64 ops_total = 0;
	if (nj > 0) {
		LAUNCH(k_mn_tail, 1, nj, mn);
		RC(prim_exscan_i32_i64(c, mn, c->j_opsoff.as<i64>(), (size_t)nj + 1));
		GSA_CHECK(c, hipMemcpyAsync(&ops_total, c->j_opsoff.as<i64>() + nj, 8, hipMemcpyDeviceToHost, st));
		GSA_CHECK(c, hipStreamSynchronize(st));
		ENS(uint8_t, d_ops, ops_total + 64);
		RC(run_ksw2_jobs(c, nj, c->di.ref, off1, len1, c->d_query.as<uint8_t>(), off2, len2, c->d_ops.as<uint8_t>(), c->j_opsoff.as<i64>(), c->j_nops.as<i32>(), ops_total));
	}
	// gapped-string offsets
	ENS(i64, d_alnoff, nf + 2);
	LAUNCH(k_aln_len, nf + 1, nf, c->f_type.as<i32>(), c->f_rec.as<gsa_frag>(), c->f_job.as<i32>(), c->j_nops.as<i32>(), c->f_alnlen.as<i32>());
	RC(prim_exscan_i32_i64(c, c->f_alnlen.as<i32>(), c->d_alnoff.as<i64>(), (size_t)nf + 1));
	i64 na = 0;
	GSA_CHECK(c, hipMemcpyAsync(&na, c->d_alnoff.as<i64>() + nf, 8, hipMemcpyDeviceToHost, st));
	GSA_CHECK(c, hipStreamSynchronize(st));
	c->n_aln = na;
	ENS(uint8_t, d_aln1, na + 64); ENS(uint8_t, d_aln2, na + 64);
	i32 *c_len = c->d_flag.as<i32>(), *c_score = c->f_score.as<i32>();
	hipLaunchKernelGGL(k_materialize, dim3((unsigned)nf), dim3(64), 0, st, nf, c->f_type.as<i32>(), c->f_mism.as<i32>(), c->f_job.as<i32>(), c->f_alnlen.as<i32>(),
	                   c->d_alnoff.as<i64>(), c->d_ops.as<uint8_t>(), c->j_opsoff.as<i64>(), c->d_query.as<uint8_t>(), c->di.ref,
	                   c->f_rec.as<gsa_frag>(), c->d_aln1.as<uint8_t>(), c->d_aln2.as<uint8_t>(), c_len, c_score);
	// per-block sums via prefix sums
	ENS(i64, d_i64a, nf + 2); ENS(i64, j_cells, nf + 2);
	LAUNCH(k_mn_tail, 1, (i32)nf, c_len); LAUNCH(k_mn_tail, 1, (i32)nf, c_score);
	RC(prim_exscan_i32_i64(c, c_len, c->d_i64a.as<i64>(), (size_t)nf + 1));
	RC(prim_exscan_i32_i64(c, c_score, c->j_cells.as<i64>(), (size_t)nf + 1));
	ENS(i32, bl_alnlen, nfb + 1); ENS(i32, bl_score, nfb + 1);
	LAUNCH(k_block_sums, nfb, nfb, nf, c->fb_fragbase.as<i32>(), c->d_i64a.as<i64>(), c->j_cells.as<i64>(), c->bl_alnlen.as<i32>(), c->bl_score.as<i32>());
	if (c->profiling) hipEventRecord(c->ev[9], st);
	GSA_CHECK(c, hipStreamSynchronize(st));
	if (c->profiling) { float ms; hipEventElapsedTime(&ms, c->ev[8], c->ev[9]); c->kernel_ms[5] = ms; }
	return GSA_OK;
}
