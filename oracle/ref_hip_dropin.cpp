// oracle/ref_hip_dropin.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// INTEGRATION.md section 2 as a compiled artefact: the reference's OWN program -- its main(), FASTA loader, index loader, MAF / ALN / VCF
// emitters, compiled from /root/reference where the sources lie -- with the body of GenomeComparison()'s per-sequence loop
// (/root/reference/src/GSAlign.cpp:483-540: eight pthread stages) replaced by ONE call into libgsa_hip.so.  oracle/Makefile compiles the
// reference's GSAlign.cpp with -DGenomeComparison=GenomeComparison_cpu (its definition keeps the old name's body out of the way, every global
// it defines stays) and links this file's GenomeComparison() in its place: oracle/_ref/GSAlign_ref_hip.  tests/test_gpu_cli.py compares its
// MAF / VCF bytes with the unmodified reference's.
#include "structure.h"      // the reference's (-I /root/reference/src)
#include "gsa_hip.h"

extern int DupAlnNum;                                  // GSAlign.cpp:14-15 (defined there, not in structure.h)
extern int64_t TotalAlignmentMatches;

static gsa_ctx *Gpu;
static std::vector<int32_t> ChrLen;

static void GpuInit()                                  // once, after RestoreReferenceInfo() (main.cpp:306-314)
{
	gsa_index_view v;
	v.primary = Refbwt->primary; for (int i = 0; i < 5; i++) v.L2[i] = Refbwt->L2[i];
	v.bwt = Refbwt->bwt; v.bwt_words = Refbwt->bwt_size;
	v.sa = (const uint64_t *)Refbwt->sa; v.n_sa = Refbwt->n_sa;
	v.ref = RefSequence; v.G = GenomeSize;
	for (int i = 0; i < iChromsomeNum; i++) ChrLen.push_back(ChromosomeVec[i].len);
	v.chr_len = ChrLen.data(); v.n_chr = iChromsomeNum;
	gsa_params p = { MinSeedLength, MaxIndelSize, MinAlnBlockScore, MinAlnLength, MinSeqIdy, bSensitive ? 1 : 0, OneOnOneMode ? 1 : 0 };
	if (gsa_create(0, &v, &p, &Gpu) != GSA_OK) { fprintf(stderr, "libgsa_hip: %s\n", gsa_last_error(NULL)); exit(1); }
}

void GenomeComparison()
{
	GpuInit();
	fprintf(stderr, "Step2. Sequence analysis for all query chromosomes (libgsa_hip.so)\n");
	for (QueryChrIdx = 0; QueryChrIdx != iQueryChrNum; QueryChrIdx++) {
		const std::string &seq = QueryChrVec[QueryChrIdx].seq;
		gsa_result r;
		if (gsa_align_contig(Gpu, seq.data(), (int32_t)seq.size(), &r) != GSA_OK) { fprintf(stderr, "libgsa_hip: %s\n", gsa_last_error(Gpu)); exit(1); }
		AlnBlockVec.clear();
		int n = 0;
		for (int b = 0; b < r.n_blocks; b++) {          // rebuild AlnBlockVec for OutputMAF / OutputAlignment / VariantIdentification / OutputDotplot
			AlnBlock_t B; B.score = r.blocks[b].score; B.aln_len = r.blocks[b].aln_len; B.bDup = r.blocks[b].bdup != 0;
			B.coor.bDir = r.blocks[b].bdir != 0; B.coor.gPos = r.blocks[b].gpos; B.coor.ChromosomeIdx = r.blocks[b].chr;
			for (int k = 0; k < r.blocks[b].n_frag; k++) {
				gsa_frag f; gsa_rec_expand(r.recs, r.blocks[b].frag_off + k, &f);
				FragPair_t F; F.bSeed = f.bseed != 0; F.qPos = f.qpos; F.qLen = f.qlen; F.rLen = f.rlen; F.rPos = f.rpos; F.PosDiff = f.rpos - f.qpos;
				if (!f.bseed) { F.aln1.assign(r.aln1 + f.aln_off, (size_t)f.aln_len); F.aln2.assign(r.aln2 + f.aln_off, (size_t)f.aln_len); }
				B.FragPairVec.push_back(F);
			}
			// the tail of the loop body, unchanged in meaning (GSAlign.cpp:529-540): the blocks that come back have passed the identity filter
			if (B.bDup) DupAlnNum++;
			n++; LocalAlignmentNum++; TotalAlignmentLength += B.aln_len; TotalAlignmentMatches += B.score;
			AlnBlockVec.push_back(B);
		}
		if (n == 0) continue;
		if (OutputFormat == 1) OutputMAF();
		if (OutputFormat == 2) OutputAlignment();
		if (bVCF) VariantIdentification();
		if (bShowPlot && GnuPlotPath != NULL) OutputDotplot();
	}
	gsa_destroy(Gpu);
}
