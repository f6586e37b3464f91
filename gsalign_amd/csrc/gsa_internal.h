// gsalign_amd/csrc/gsa_internal.h -- shared between the translation units of
// libgsa_hip.so.  Device-side index view, the context, small helpers.
#ifndef GSA_INTERNAL_H
#define GSA_INTERNAL_H
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include "../../include/gsa_hip.h"

typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;
typedef int32_t i32;

// hard constants of the reference
#define GSA_CHUNK 10000          // SeedExplorationChunk, GSAlign.cpp:5
#define GSA_MAX_SEED_FREQ 100    // MaxSeedFreq, bwt_search.cpp:3
#define GSA_MAX_SEED_GAP 5000    // MaxSeedGap, structure.h:23
#define GSA_GAP_CHECK 300        // ProcessCandidateAlignment.cpp:131
#define GSA_MAX_MISMATCH 5       // ProcessCandidateAlignment.cpp:327
#define GSA_WIN_SEEDS 30         // GSAlign.cpp:331
#define GSA_WIN_SPAN 3000        // GSAlign.cpp:331

// ---- device view of the FM-index (a1) --------------------------------------
// bwt: 64-byte blocks; block b covers BWT rows 128b..128b+127.  Words 0-7 = four
// u64 running counts (A,C,G,T before the block); words 8-15 = 128 symbols, 2 bit
// each, MSB first inside each u32.
struct DevIndex {
	u64 primary, L2[5], seq_len;
	const uint4 *bwt;       // Occ blocks, 2 x uint4 per 64 rows: four u32 counts + 64 symbols (FmBlock, gsa_fm.h; built at gsa_create from the reference's layout)
	const u64 *occ_base;    // 64-bit base counts (A, C, G, T) per super-block of 2^occ_shift blocks; null when all counts fit 32 bits
	i32 occ_shift;
	const u64 *sa;          // sa[i] = SA of row 32 i ; sa[0] = -1   (the on-disk sampling)
	const u32 *sa32;        // dense SA, one entry per row, built on the device at gsa_create
	const u64 *sa64;        //   (32-bit entries when 2G < 2^32, else 64-bit); row 0 is the -1 sentinel
	const uint8_t *ref;     // 2G ASCII
	const u32 *ref2;        // the same text 2-bit packed, 16 bases per word, LSB first (built at gsa_create)
	i64 G;
	const i64 *chr_end;     // 2*n_chr sorted last coordinates (ChrLocMap keys)
	const i32 *chr_of_end;  // chromosome index per entry (ChrLocMap values)
	i32 n_ends;
	const u64 *kmer;        // k-mer -> (x0,x1,x2,loc+1) after the first kmer_k bases, x2 = 0: absent (built at gsa_create);
	                        // four u32 per entry when kmer_e16 (text < 2^32), else four u64
	i32 kmer_e16;
	i32 kmer_k;
	const u64 *kmer_lo;     // a second, short jump table of kmer_lo_k = MinSeedLength bases (same entry format) for starts whose match ends
	i32 kmer_lo_k;          //   before kmer_k: built by gsa_set_params when MinSeedLength < kmer_k (-sen: 10), used by the dense search
	const u32 *pres;        // presence table of all pres_k-mers of the text (pres_k = min(MinSeedLength, 16)), GROUPED: one 32-byte line answers
	i32 pres_k;             //   for four consecutive start positions (layout: pres4_* in k_seed.hip); rebuilt by gsa_set_params
};

// ---- several query contigs in ONE pass (gsa_align_many bundles short contigs) --------------
// The reference clears all state per query sequence (GSAlign.cpp:483-490), so contigs are independent; a contig of a few Mb is
// ~60 GPU operations whatever its size, and operations -- not bytes -- bound such contigs.  A bundle is the concatenation of n
// contigs, each padded with 'N' to a chunk edge (GSA_CHUNK: chunk grids restart per contig, matches stop at N exactly as they stop
// at the end of a sequence), searched / chained / extended as one virtual contig of length off[n]: query positions are positions
// in the concatenation, reference positions stay what they are.  Two places must know the contig of a seed: (1) the PosDiff key
// that forms the seed groups gets a per-contig stride (groups never span contigs), (2) integer means over PosDiff values truncate
// toward zero, so they are taken over the TRUE PosDiff rPos - (q - off[contig]).  n = 0: a single contig.
struct Bundle {
	i32 n;                          // contigs in the bundle (0: not a bundle)
	i32 lmax;                       // PosDiff shift: key = rPos - qLocal + lmax + contig * pds  (single contig: its length)
	i64 pds;                        // PosDiff key stride per contig: 2G + lmax + MaxIndelSize + slack, a multiple of 32
	const i32 *off;                 // [n + 1] start of contig c in the concatenation (multiples of GSA_CHUNK)
	const uint16_t *chunk_contig;   // contig of every chunk of the concatenation
};
#if defined(__HIPCC__)
__device__ __forceinline__ i32 bundle_contig(const Bundle &b, i32 q) { return b.n ? (i32)b.chunk_contig[q / GSA_CHUNK] : 0; }
__device__ __forceinline__ i32 bundle_off(const Bundle &b, i32 q) { return b.n ? b.off[b.chunk_contig[q / GSA_CHUNK]] : 0; }
// the same without a branch around the loads (the fused passes' load(): a context without a bundle points both tables at zeros)
__device__ __forceinline__ i32 bundle_off_flat(const Bundle &b, i32 q) { return b.off[b.chunk_contig[b.n ? q / GSA_CHUNK : 0]]; }
#endif

struct DevBuf {
	void *p = nullptr; size_t cap = 0;
	size_t len = 0;            // bytes last asked for (dev_ensure); the index tables are asked for once, so this is their true size (gsa_clone_to_device copies that much)
	template <class T> T *as() const { return (T *)p; }
};

struct Params {
	i32 MinSeedLength, MaxIndelSize, MinAlnBlockScore, MinAlnLength, MinSeqIdy;
	i32 bSensitive, OneOnOne;
};

#define GSA_CHECK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return gsa_fail((ctx), GSA_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

int gsa_fail(gsa_ctx *ctx, int code, const std::string &msg);

// ---- rocPRIM-backed primitives (gsa_prim.hip) --------------------------------
// All asynchronous on `stream`; `tmp` is a reusable scratch buffer that grows.
int gsa_sort_pairs_u64_u32(gsa_ctx *, const u64 *kin, u64 *kout, const u32 *vin, u32 *vout, size_t n, int begin_bit, int end_bit);      // gsa_sort.hip: stable LSD radix sort, kin / vin untouched

static inline int ceil_log2_u64(u64 v) { int b = 0; while ((1ull << b) < v && b < 63) b++; return b; }
// workgroups for n work-items.  A launch holds fewer than 2^32 work-items (the dispatch packet's grid size is a 32-bit count of work-items; the
// runtime takes a larger one modulo 2^32 without a word -- round 5's presence-table bug): a caller with more elements than that strides.
// A launch that would need more is NOT made to fit silently and does not kill the host process either: the grid is clamped (so the launch itself is legal),
// the overflow is latched, and the entry point that issued it returns GSA_ERR_LIMIT (gsa_take_grid_overflow in gsa_run_to / gsa_create_opts).
inline std::atomic<unsigned long long> gsa_grid_overflow{0};
static inline unsigned grid_for(size_t n, unsigned block)
{
	size_t g = (n + block - 1) / block;
	if (g * block >= ((size_t)1 << 32)) { gsa_grid_overflow.store((unsigned long long)(g * block)); g = (((size_t)1 << 32) - 1) / block; }
	return (unsigned)(g ? g : 1);
}
static inline unsigned long long gsa_take_grid_overflow() { return gsa_grid_overflow.exchange(0); }

#endif
