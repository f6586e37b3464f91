// oracle/ref_glue.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A small C API around the *real* reference objects (hsinnan75/GSAlign
// v1.0.22, compiled in place from /root/reference by oracle/Makefile into
// oracle/_ref/libgsref.so).  Nothing here re-implements the algorithm: every
// stage below calls the reference's own function.  The only thing restated is
// the order in which GenomeComparison() (reference src/GSAlign.cpp:473-552)
// launches its stages, because that function offers no hook between stages and
// the parity tests need SeedVec / AlnBlockVec after *each* stage.
//
// Always single-threaded (iThreadNum = 1): the reference has a data race in
// S4/S5 when threaded, and block order on score ties
// depends on thread interleaving (B #10).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library.
#include "structure.h"          // the reference's header, found via -I$(REF)/src
#include <unistd.h>
#include <fcntl.h>

// ---- reference symbols that structure.h does not declare ------------------
// src/GSAlign.cpp:9-15 (file-scope globals with external linkage)
extern int64_t *RefChrScoreArr;
extern vector<FragPair_t> SeedVec;
extern uint32_t QrySeqPos, QryChrLength;
extern vector<pair<int, int> > SeedGroupVec;
extern int SeedNum, SeedGroupNum, GroupID, AlnBlockNum;
// src/GSAlign.cpp:51,126,377,393,415
extern void *IdentifyLocalMEM(void *arg);
extern int SeedGrouping();
extern void *GenerateAlignmentBlocks(void *arg);
extern void EstChromosomeSimilarity();
extern void RemoveRedundantAlnBlocks(int type);
// src/bwt_search.cpp:129
extern bwtint_t bwt_sa(bwtint_t k);
extern "C" int bwa_idx_build(const char *fa, const char *prefix);

namespace {
int g_stage = 0;
int g_tid = 0;
bool g_loaded = false;

struct StderrMute {           // the reference prints a \r progress line per chunk
	int saved;
	StderrMute() { fflush(stderr); saved = dup(2); int nul = open("/dev/null", O_WRONLY); dup2(nul, 2); close(nul); }
	~StderrMute() { fflush(stderr); dup2(saved, 2); close(saved); }
};
}

extern "C" {

// bwa_idx_load + RestoreReferenceInfo, as main() does (src/main.cpp:306,320-321)
int gsref_init(const char *prefix)
{
	if (g_loaded) return -1;   // ChrLocMap is a global that RestoreReferenceInfo only appends to
	StderrMute mute;
	iThreadNum = 1;
	if (!CheckBWAIndexFiles(prefix)) return -2;
	RefIdx = bwa_idx_load(prefix);
	if (RefIdx == 0) return -3;
	Refbwt = RefIdx->bwt;
	RestoreReferenceInfo();
	RefChrScoreArr = new int64_t[iChromsomeNum];
	pthread_mutex_init(&Lock, NULL);
	g_loaded = true;
	return 0;
}

int gsref_build_index(const char *fa, const char *prefix)
{
	StderrMute mute;
	return bwa_idx_build(fa, prefix);
}

// defaults and -sen handling follow src/main.cpp:202-214,272-277,323
void gsref_params(int slen, int ind, int clr, int alen, int idy, int sen, int one)
{
	MinSeedLength = slen; MaxIndelSize = ind; MinAlnBlockScore = clr; MinAlnLength = alen;
	MinSeqIdy = idy; bSensitive = (sen != 0); OneOnOneMode = (one != 0);
	if (bSensitive) MinSeedLength = 10;
	iThreadNum = 1; bVCF = true; bAllowDuplication = true; OutputFormat = 1;
}

int gsref_set_query(const char *name, const char *seq, int len)
{
	QueryChrVec.clear(); QueryChrVec.resize(1);
	QueryChrVec[0].name = name; QueryChrVec[0].seq.assign(seq, len);
	iQueryChrNum = 1; QueryChrIdx = 0;
	QrySeqPos = 0; QryChrLength = (uint32_t)len;
	SeedVec.clear(); SeedGroupVec.clear(); AlnBlockVec.clear();   // src/GSAlign.cpp:490
	g_stage = 0;
	return 0;
}

// Stage numbering used by every parity test (same numbers in gsa_oracle.cpp):
//  1 S1 IdentifyLocalMEM + SeedGrouping            src/GSAlign.cpp:492-495
//  2 S2 GenerateAlignmentBlocks                    :497
//  3 S3 CheckAlnBlockOverlaps                      :501
//  4 S4 CheckAlnBlockLargeGaps + RemoveBad         :504-505
//  5 S5 CheckAlnBlockSpanMultiSeqs + RemoveBad     :507-508
//  6 bDup reset, EstChromosomeSimilarity, RemoveRedundantAlnBlocks(1),(2)  :510-511
//  7 S6 FillAlnBlockGaps                           :513
//  8 S7 GenerateFragAlignment + identity filter + RemoveBad   :523-540
int gsref_run_to(int stage)
{
	StderrMute mute;
	vector<AlnBlock_t>::iterator it;
	while (g_stage < stage) {
		switch (++g_stage) {
		case 1: IdentifyLocalMEM(NULL); SeedNum = (int)SeedVec.size(); GroupID = 0; SeedGroupNum = SeedGrouping(); break;
		case 2: GenerateAlignmentBlocks(&g_tid); break;
		case 3: AlnBlockNum = (int)AlnBlockVec.size(); CheckAlnBlockOverlaps(&g_tid); break;
		case 4: AlnBlockNum = (int)AlnBlockVec.size(); CheckAlnBlockLargeGaps(&g_tid); RemoveBadAlnBlocks(); break;
		case 5: AlnBlockNum = (int)AlnBlockVec.size(); CheckAlnBlockSpanMultiSeqs(&g_tid); RemoveBadAlnBlocks(); break;
		case 6:
			for (it = AlnBlockVec.begin(); it != AlnBlockVec.end(); it++) it->bDup = false;
			EstChromosomeSimilarity(); RemoveRedundantAlnBlocks(1); RemoveRedundantAlnBlocks(2);
			break;
		case 7: AlnBlockNum = (int)AlnBlockVec.size(); FillAlnBlockGaps(&g_tid); break;
		case 8:
			for (it = AlnBlockVec.begin(); it != AlnBlockVec.end(); it++) it->aln_len = it->score = 0;
			GenerateFragAlignment(&g_tid);
			for (it = AlnBlockVec.begin(); it != AlnBlockVec.end(); it++) {
				if ((int)(100 * (1.0*it->score / it->aln_len)) < MinSeqIdy) it->score = 0;
				else it->coor = GenCoordinateInfo(it->FragPairVec[0].rPos);
			}
			RemoveBadAlnBlocks();
			break;
		default: return -1;
		}
	}
	return g_stage;
}

long long gsref_seed_count() { return (long long)SeedVec.size(); }
void gsref_seeds(int *qpos, int *qlen, long long *rpos)
{
	for (size_t i = 0; i < SeedVec.size(); i++) { qpos[i] = SeedVec[i].qPos; qlen[i] = SeedVec[i].qLen; rpos[i] = SeedVec[i].rPos; }
}
int gsref_group_count() { return (int)SeedGroupVec.size(); }
void gsref_groups(int *beg, int *end)
{
	for (size_t i = 0; i < SeedGroupVec.size(); i++) { beg[i] = SeedGroupVec[i].first; end[i] = SeedGroupVec[i].second; }
}

int gsref_block_count() { return (int)AlnBlockVec.size(); }
long long gsref_frag_total()
{
	long long n = 0;
	for (size_t i = 0; i < AlnBlockVec.size(); i++) n += (long long)AlnBlockVec[i].FragPairVec.size();
	return n;
}
long long gsref_aln_total()
{
	long long n = 0;
	for (size_t i = 0; i < AlnBlockVec.size(); i++)
		for (size_t j = 0; j < AlnBlockVec[i].FragPairVec.size(); j++) n += (long long)AlnBlockVec[i].FragPairVec[j].aln1.length();
	return n;
}
// coor is only meaningful after stage 8
void gsref_block_meta(int *score, int *aln_len, int *bdup, int *nfrag, int *bdir, int *gpos, int *chr)
{
	for (size_t i = 0; i < AlnBlockVec.size(); i++) {
		const AlnBlock_t &b = AlnBlockVec[i];
		score[i] = b.score; nfrag[i] = (int)b.FragPairVec.size();
		aln_len[i] = g_stage >= 8 ? b.aln_len : 0;
		bdup[i] = g_stage >= 6 ? (b.bDup ? 1 : 0) : 0;
		bdir[i] = g_stage >= 8 ? (b.coor.bDir ? 1 : 0) : 0;
		gpos[i] = g_stage >= 8 ? b.coor.gPos : 0;
		chr[i]  = g_stage >= 8 ? b.coor.ChromosomeIdx : 0;
	}
}
void gsref_frags(int *bseed, int *qpos, int *qlen, long long *rpos, int *rlen, int *alnlen)
{
	size_t k = 0;
	for (size_t i = 0; i < AlnBlockVec.size(); i++)
		for (size_t j = 0; j < AlnBlockVec[i].FragPairVec.size(); j++, k++) {
			const FragPair_t &f = AlnBlockVec[i].FragPairVec[j];
			bseed[k] = f.bSeed ? 1 : 0; qpos[k] = f.qPos; qlen[k] = f.qLen; rpos[k] = f.rPos; rlen[k] = f.rLen;
			alnlen[k] = (int)f.aln1.length();
		}
}
void gsref_frag_aln(char *a1, char *a2)
{
	size_t p = 0;
	for (size_t i = 0; i < AlnBlockVec.size(); i++)
		for (size_t j = 0; j < AlnBlockVec[i].FragPairVec.size(); j++) {
			const FragPair_t &f = AlnBlockVec[i].FragPairVec[j];
			memcpy(a1 + p, f.aln1.data(), f.aln1.length()); memcpy(a2 + p, f.aln2.data(), f.aln2.length());
			p += f.aln1.length();
		}
}

// ---- function-level known answers ------------------------------------------
// src/ksw2_alignment.cpp:251 ; out buffers must hold m+n+1 bytes
int gsref_ksw2(const char *s1, int m, const char *s2, int n, char *out1, char *out2)
{
	string a(s1, m), b(s2, n);
	ksw2_alignment(m, a, n, b);
	memcpy(out1, a.data(), a.length()); memcpy(out2, b.data(), b.length());
	return (int)a.length();
}
// src/KmerAnalysis.cpp:78 (uses the current query + loaded reference)
int gsref_gap_similarity(int q1, int q2, long long r1, long long r2) { return CalGapSimilarity(q1, q2, r1, r2) ? 1 : 0; }
// src/bwt_search.cpp:141 ; locs must hold 100 entries
int gsref_bwt_search(int start, int stop, int *len, long long *locs)
{
	bwtSearchResult_t r = BWT_Search(QueryChrVec[0].seq, start, stop);
	*len = r.len;
	for (int i = 0; i < r.freq; i++) locs[i] = (long long)r.LocArr[i];
	if (r.LocArr) delete[] r.LocArr;
	return r.freq;
}
long long gsref_bwt_sa(unsigned long long k) { return (long long)bwt_sa(k); }
long long gsref_genome_size() { return (long long)GenomeSize; }
const char *gsref_refseq() { return RefSequence; }

} // extern "C"
