/* include/gsa_hip.h -- C ABI of libgsa_hip.so, the MI355X (gfx950) hot path of a
 * GSAlign-compatible whole-genome aligner.
 *
 * The reference (hsinnan75/GSAlign v1.0.22) has no plugin / FFI interface; its
 * hot path is the set of functions GenomeComparison() launches per query contig
 * (reference src/GSAlign.cpp:483-540) and they communicate through globals.
 * This header is the seam a maintainer would bind instead (SURVEY.md section
 * 8(b)); every entry point cites the reference code it replaces.  Plain C:
 * pointers and sizes only, no C++/torch types.  One gsa_ctx per GPU; a context
 * is not thread-safe; calls are synchronous for the caller.
 *
 * All results are bit-identical to the reference's: seeds, blocks, gap records,
 * op strings, scores.
 */
#ifndef GSA_HIP_H
#define GSA_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct gsa_ctx gsa_ctx;

/* error codes (negative); gsa_last_error() has the text */
enum {
	GSA_OK = 0,
	GSA_ERR_ARG = -1,      /* bad argument                      */
	GSA_ERR_HIP = -2,      /* a HIP runtime call failed         */
	GSA_ERR_NOMEM = -3,    /* device or host allocation failed  */
	GSA_ERR_STATE = -4,    /* call order violated               */
	GSA_ERR_LIMIT = -5     /* an internal capacity was exceeded */
};

/* The in-memory index the reference keeps in Refbwt / RefSequence / ChromosomeVec
 * (structure.h:28-38,141-155; filled by bwa_idx_load + RestoreReferenceInfo,
 * bwt_index.cpp:147-264).  Host pointers; gsa_create copies them to the device
 * and does not keep them. */
typedef struct {
	uint64_t primary;          /* bwt_t::primary                                        */
	uint64_t L2[5];            /* bwt_t::L2, L2[0] = 0, L2[4] = seq_len = 2G            */
	const uint32_t *bwt;       /* bwt_t::bwt: interleaved Occ checkpoints + 2-bit BWT   */
	uint64_t bwt_words;        /* bwt_t::bwt_size                                       */
	const uint64_t *sa;        /* bwt_t::sa, sa[0] = (uint64_t)-1, sa_intv = 32         */
	uint64_t n_sa;             /* bwt_t::n_sa                                           */
	const char *ref;           /* RefSequence: 2G ASCII bytes, forward + reverse compl. */
	int64_t G;                 /* GenomeSize (forward length)                           */
	const int32_t *chr_len;    /* ChromosomeVec[i].len                                  */
	int32_t n_chr;             /* iChromsomeNum                                         */
} gsa_index_view;

/* The tunables the reference parses into globals (main.cpp:202-214,236-299,323). */
typedef struct {
	int32_t min_seed_len;      /* -slen  MinSeedLength     [15]; forced to 10 by -sen  */
	int32_t max_indel;         /* -ind   MaxIndelSize      [25]                        */
	int32_t min_block_score;   /* -clr   MinAlnBlockScore  [200] (50 with -sen)        */
	int32_t min_aln_len;       /* -alen  MinAlnLength      [200]                       */
	int32_t min_identity;      /* -idy   MinSeqIdy         [70]                        */
	int32_t sensitive;         /* -sen   bSensitive        [0]                         */
	int32_t one_on_one;        /* -one   OneOnOneMode      [0]                         */
} gsa_params;
void gsa_default_params(gsa_params *p);

/* ---- records ------------------------------------------------------------ */
/* one located exact match; the reference's FragPair_t with bSeed = true
 * (structure.h:103-113), order = CompByPosDiff (ProcessCandidateAlignment.cpp:3-7) */
typedef struct { int32_t qpos; int32_t len; int64_t rpos; } gsa_seed;

/* FragPair_t of a finished block (structure.h:103-113), expanded.  aln_off/aln_len address the two gapped strings of a
 * non-seed pair inside gsa_result::aln1 / aln2.  This is the VIEW type: results travel as the 16-byte gsa_rec defined below and are expanded
 * with gsa_rec_expand / gsa_expand_frags where a consumer wants FragPair_t-shaped records. */
typedef struct {
	int32_t bseed, qpos, qlen, rlen;
	int64_t rpos;
	int64_t aln_off;
	int32_t aln_len;
	int32_t _pad;
} gsa_frag;

/* The record as it crosses PCIe: 16 bytes instead of gsa_frag's 40; a 250 Mb contig at 1 % divergence has 4.1 M
 * records.  FillAlnBlockGaps / IdentifyNormalPairs (ProcessCandidateAlignment.cpp:241-276) put at most ONE gap pair
 * between two consecutive seeds, and it starts where the seed in front of it ends -- so a block's records are
 * seed [gap] seed [gap] ... seed, a seed needs (qPos, len, rPos), and a gap only its two lengths and its string:
 * its positions are those of the record in front of it.  w0 >= 0: seed {qpos, len, rpos}; w0 < 0: gap
 * {-1 - qlen, rlen, aln_len, aln_off} (string offsets fit 32 bits: GSA_ERR_LIMIT guards that per contig). */
typedef union {
	struct { int32_t qpos, len; int64_t rpos; } seed;
	struct { int32_t nqlen, rlen, aln_len; uint32_t aln_off; } gap;
} gsa_rec;

static inline int gsa_rec_is_seed(const gsa_rec *r) { return r->seed.qpos >= 0; }
/* record i of `recs` as a FragPair_t (a gap reads the seed in front of it: i > 0 then, by construction) */
static inline void gsa_rec_expand(const gsa_rec *recs, int64_t i, gsa_frag *f)
{
	const gsa_rec *r = recs + i;
	if (r->seed.qpos >= 0) {
		f->bseed = 1; f->qpos = r->seed.qpos; f->qlen = f->rlen = r->seed.len; f->rpos = r->seed.rpos; f->aln_off = 0; f->aln_len = 0;
	} else {
		const gsa_rec *s = r - 1;
		f->bseed = 0; f->qpos = s->seed.qpos + s->seed.len; f->rpos = s->seed.rpos + s->seed.len;
		f->qlen = -1 - r->gap.nqlen; f->rlen = r->gap.rlen; f->aln_off = (int64_t)r->gap.aln_off; f->aln_len = r->gap.aln_len;
	}
	f->_pad = 0;
}
static inline void gsa_expand_frags(const gsa_rec *recs, int64_t n, gsa_frag *out)
{
	for (int64_t i = 0; i < n; i++) gsa_rec_expand(recs, i, out + i);
}

/* AlnBlock_t (structure.h:115-122) after the identity filter (GSAlign.cpp:529-540) */
typedef struct {
	int32_t score, aln_len, bdup, n_frag;
	int64_t frag_off;          /* first record of this block             */
	int32_t bdir, gpos, chr;   /* Coordinate_t from GenCoordinateInfo    */
	int32_t _pad;
} gsa_block;

/* everything GenomeComparison() leaves in AlnBlockVec for one query contig,
 * in the reference's order (RemoveBadAlnBlocks, GSAlign.cpp:540).  Memory is
 * owned by the context and valid until the next gsa_align_contig/gsa_destroy. */
typedef struct {
	int32_t n_blocks;
	int64_t n_frags;
	int64_t n_aln;
	const gsa_block *blocks;
	const gsa_rec *recs;       /* n_frags compact records, block after block (gsa_block::frag_off / n_frag) */
	const char *aln1, *aln2;   /* reference-side / query-side gapped strings */
} gsa_result;

/* ---- life cycle --------------------------------------------------------- */
int  gsa_create(int device, const gsa_index_view *idx, const gsa_params *prm, gsa_ctx **out);
/* gsa_create with layout options.  GSA_CREATE_WIDE: the device layout of a text with >= 2^32 BWT rows (64-bit dense SA,
 * 32-byte k-mer entries) whatever the text length -- what a full human index (bwt_t::seq_len = 6.2 G, structure.h:28-38)
 * gets by itself; on a small index it lets the tests drive that code path. */
#define GSA_CREATE_WIDE 1u
/* GSA_CREATE_KMER_K(k), 2 <= k <= 15: the jump table that replaces the first k steps of BWT_Search (bwt_search.cpp:152-165) is built for
 * k-mers of this length if it fits (default: by text length and free device memory; tests: a long table on a short text). */
#define GSA_CREATE_KMER_K(k) (((uint32_t)(k) & 15u) << 8)
/* GSA_CREATE_PRIO(mode), mode 1..3: stream priorities for a host that drives SEVERAL contexts on one GPU (clones inherit it).  The short bookkeeping passes of
 * chaining / refinement / extension run on a stream of the greatest priority, the seed-search kernels on a stream of their own (1, 3: normal priority, 2: least),
 * the striped DP normal (3: least).  Results do not depend on it.  0 (default): every stream at the default priority. */
#define GSA_CREATE_PRIO(mode) (((uint32_t)(mode) & 3u) << 16)
/* GSA_CREATE_REF_PAC: idx->ref points at the bytes of the .pac file (G bases, four per byte, as bns_fasta2bntseq wrote them: bntseq.c:110-211) instead of at RefSequence.
 * The reference unpacks .pac into 2G ASCII bytes on the host (RestoreReferenceInfo, bwt_index.cpp:229-264) before anything else can start; with this flag the device does
 * that loop itself -- G / 4 bytes cross PCIe instead of 2G, and a host program can unpack its own copy (for its emitters) WHILE gsa_create builds the device tables. */
#define GSA_CREATE_REF_PAC 4u
int  gsa_create_opts(int device, const gsa_index_view *idx, const gsa_params *prm, uint32_t flags, gsa_ctx **out);
/* Optional, before gsa_create: sets device memory aside for the two largest device tables of an index whose text has `seq_len` BWT rows (bwt_t::seq_len, the fifth
 * word of the .bwt header: structure.h:28-38) -- the dense suffix array and the k-mer table, 84 GB for a human index, ~0.4 s of hipMalloc -- so that a host can do that
 * while it is still READING the index files (the reference's bwa_idx_load reads first and allocates as it goes, bwt_index.cpp:147-227).  `flags` as for gsa_create_opts.
 * The next gsa_create[_opts] on `device` adopts what fits and frees the rest; gsa_release_reserved frees a reservation nobody adopted.  Thread-safe. */
int  gsa_reserve_index(int device, uint64_t seq_len, uint32_t flags);
void gsa_release_reserved(int device);
/* Tunables that are not aligner parameters (the library reads no environment variable).  A clone starts with its parent's values;
 * gsa_align_many takes its policy from ctx[0].
 *   "split_min"      bases; gsa_align_many seeds a contig of at least this length on several contexts when contexts would idle (20 000 000)
 *   "bundle_contig"  bases; contigs up to this length share passes (16 000 000; 0 = never)      "bundle_cap"  bases per bundle, about (64 000 000)
 *   "seed_budget"    wave-iterations a 10 000-bp chunk may take in the speculative seed kernel before the dense kernels redo it (256)
 *   "dp_lane"        cells; gap alignments up to this size run one per lane (512; 0 = one per wavefront / quarter wavefront)
 *   "seed_mode"      1 = speculative kernel + dense kernels for the chunks it gives up on (default), 0 = every chunk through the
 *                    right-to-left sweep, 2 = one search per start instead of the sweep
 *   "pd_bitmap"      0 = seed groups by the PosDiff sort (SeedGrouping as written, GSAlign.cpp:126-143) even where the bitmap scan applies
 *   "sweep_shape"    launch shape of the repeat-regime seed kernel (k_dense_sweep): -1 (default) by the number of dense chunks, 0 = four chunks per
 *                    workgroup and 160-start segments (many chunks), 1 = one chunk per workgroup and 40-start segments (few).  Results do not depend on it
 *   "dp_side"        1 = the striped DP launches its lower size class on a stream of its own, beside the upper class, instead of behind it (measured: the DP
 *                    span of a 250 Mb contig 4.9 -> 1.9 ms, but the refinement passes beside it starve: contig latency 13.7 -> 14.3 ms, throughput -4 % / +4 %); default 0
 *   "walk_chain_min" seeds; a contig with more seeds than this walks its window chain (chaining, GSAlign.cpp:326-338) in slices of the candidate list,
 *                    one launch (default 100 000; below that one workgroup holds the chain in LDS).  Results do not depend on it (tests: 0)
 *   "pd_two_level_min" blocks; PosDiff bitmaps larger than this are scanned in two passes (touched blocks listed, then counted); default 2 000 000
 *                    (references above ~1 Gbp).  "pres_from_kmer" 0 = the presence table always from a scan of the text.  Results do not depend on either
 *   "pd_bytes"       how k_seed_select marks the occupied PosDiff values of a contig whose hits scatter over the genome (-sen: thousands of chance hits per
 *                    10 000-bp chunk): 1 (default) = through a byte per value, plain stores, packed into the bitmap by a pass of its own, when there are at least
 *                    512 hits per chunk, at most 256 values per hit and at most 2^32 values (2 GB per context for a 64 Mb bundle against a 12 Mb reference); 0 = always with atomics
 *                    on the bitmap; 2 = always through the bytes (tests).  Results do not depend on it
 *   "dp_safe", "dp_fake_timeout"   test hooks: one striped DP job per launch; the next n contigs report a stripe hand-off time-out once
 * Unknown names and values outside an option's range (negative sizes, seed_budget 0, ...): GSA_ERR_ARG, nothing changed.
 * Until round 4 some of these were environment variables read by the library (GSA_SPLIT_MIN, GSA_BUNDLE_CONTIG, GSA_BUNDLE_CAP, GSA_SEED_BUDGET,
 * GSA_DP_LANE, GSA_SEED_MODE, GSA_NO_PDBITMAP, GSA_FORCE_WIDE, GSA_KMER_K, GSA_DP_SAFE, GSA_DP_FAKE_TIMEOUT, GSA_NO_BIND): the library ignores
 * them now -- a C host sets the option (or the gsa_create_opts flag) itself; INTEGRATION.md lists the mapping. */
int  gsa_set_option(gsa_ctx *ctx, const char *name, int64_t value);
/* A further context on the same GPU that borrows `parent`'s device-resident index (read-only) and owns everything else.
 * The reference runs -t N threads inside one contig (GSAlign.cpp:477-526); contigs are independent (all per-contig state
 * is cleared at GSAlign.cpp:490), so a host drives N contexts from N threads on N contigs instead and the GPU overlaps
 * them.  Each context is single-threaded; `parent` must outlive its clones.  While clones of a context are alive,
 * gsa_set_params on THAT context may not change min_seed_len / sensitive (GSA_ERR_STATE: the clones read its presence bitmap
 * and short k-mer table); a clone may change its own parameters freely -- it then builds tables of its own. */
int  gsa_clone(gsa_ctx *parent, gsa_ctx **out);
/* A context on GPU `device` with a device index of ITS OWN, copied from `parent`'s device to device (over xGMI between GPUs) instead of
 * uploaded and rebuilt.  The reference loads its index once per process and every thread reads it (RefIdx / RefSequence / ChrLocMap,
 * bwt_index.cpp:147-264; the workers of GSAlign.cpp:477-526 share it); N GPUs need N copies, and gsa_create per GPU pays the PCIe upload
 * (10.7 GB for a human index) and the device-side table builds every time.  This pays one device-to-device copy of the finished tables.
 * The new context is independent of `parent` afterwards (it may outlive it), starts with parent's parameters and options, and may itself
 * be gsa_clone'd.  `device` may be parent's own GPU (a second, independent copy: what the tests on a one-GPU box exercise). */
int  gsa_clone_to_device(gsa_ctx *parent, int device, gsa_ctx **out);
void gsa_destroy(gsa_ctx *ctx);
/* Puts the CALLING host thread on the CPUs of the socket `device` hangs off (sysfs local_cpulist of its PCI function; a no-op
 * without that information).  The reference's worker threads (GSAlign.cpp:479, pthread_create) run
 * wherever the OS puts them, which costs nothing there; a thread that drives a GPU through ~60 short operations per 5 Mb
 * contig pays the inter-socket hop on every one (0.86 -> 0.65-0.75 ms per 5 Mb contig); chromosome-sized contigs ran 5 %
 * slower with bound threads, so nothing in the library calls this by itself: threads started by gsa_align_many inherit the
 * caller's affinity. */
int  gsa_bind_host_thread(int device);
/* Pinned host memory for query contigs (QueryChrVec[i].seq, main.cpp:82-114): the upload inside gsa_align_contig is then
 * one asynchronous DMA transfer.  Any other host memory works too (staged by the runtime). */
void *gsa_host_alloc(size_t bytes);
void  gsa_host_free(void *p);
/* The same for memory the host already owns (the std::string a FASTA loader filled): page-locks [p, p + bytes) in place so that the upload is a DMA
 * transfer instead of a staged copy (hipHostRegister: 10 ms per GB on the test host, hipHostMalloc of fresh pinned memory: 160 ms per GB).  The
 * reference keeps QueryChrVec[i].seq in ordinary memory (main.cpp:82-114); a host that does the same calls this once per sequence after loading.
 * Returns GSA_OK or GSA_ERR_HIP (then the buffer simply stays pageable). */
int   gsa_host_register(void *p, size_t bytes);
int   gsa_host_unregister(void *p);
int  gsa_set_params(gsa_ctx *ctx, const gsa_params *prm);
const char *gsa_last_error(gsa_ctx *ctx);   /* ctx may be NULL: error of the last failed gsa_create */

/* ---- the drop-in call ----------------------------------------------------
 * Replaces the body of the per-contig loop of GenomeComparison()
 * (GSAlign.cpp:483-540): S1 IdentifyLocalMEM ... S7 GenerateFragAlignment,
 * identity filter, GenCoordinateInfo, final RemoveBadAlnBlocks.
 * query = raw contig bytes exactly as loaded from FASTA (any case, IUPAC ok). */
int gsa_align_contig(gsa_ctx *ctx, const char *query, int32_t qlen, gsa_result *out);
/* The same with the contig ALREADY IN DEVICE MEMORY of the context's GPU (a loader that decodes FASTA on the GPU, or contigs
 * kept resident between runs): used in place, nothing crosses PCIe on the way in.  d_query: 16-byte aligned, qlen raw bytes
 * as above, valid and unmodified until the call returns.  gsa_device_alloc / gsa_device_upload / gsa_device_free are the
 * plain-C way to such a buffer for hosts that do not link a HIP runtime themselves. */
int gsa_align_contig_device(gsa_ctx *ctx, const char *d_query, int32_t qlen, gsa_result *out);
/* Upload of the NEXT contig while the current one is aligned.  The reference reads QueryChrVec[i].seq from host memory when the
 * loop reaches it (GSAlign.cpp:483-490, loader main.cpp:82-114); a GPU has to copy it first -- 4.8 ms per 250 Mb -- and that copy
 * hides behind the stages of the contig in front of it: a context has two device buffers for query sequences, gsa_prefetch_contig
 * (or _bundle) starts the copy of `query` into the one the next gsa_align_contig / gsa_align_bundle will not use and returns at
 * once; the following gsa_align_contig(ctx, query, qlen) -- same pointer, same length -- finds the contig on the device.  Call
 * order: gsa_prefetch_contig(next); gsa_align_contig(current); ...  `query` must stay valid and unmodified until it has been
 * aligned (or gsa_cancel_prefetch returned).  At most two contigs wait at a time (GSA_ERR_STATE beyond that).  After a prefetch
 * gsa_rewind may report GSA_ERR_STATE (the previous contig's device copy is gone).  gsa_align_many does all this by itself. */
int gsa_prefetch_contig(gsa_ctx *ctx, const char *query, int32_t qlen);
int gsa_prefetch_bundle(gsa_ctx *ctx, const char *const *query, const int32_t *qlen, int32_t n);
int gsa_cancel_prefetch(gsa_ctx *ctx);
void *gsa_device_alloc(int device, size_t bytes);
void  gsa_device_free(int device, void *p);
int   gsa_device_upload(int device, void *dst, const void *src, size_t bytes);

/* ---- the per-contig loop itself ------------------------------------------
 * Replaces the loop `for (QueryChrIdx = 0; QueryChrIdx < iQueryChrNum; ...)` of GenomeComparison() (GSAlign.cpp:483-548):
 * n contigs on n_ctx contexts -- contexts of different GPUs (gsa_create per device) and/or several contexts of one GPU
 * (gsa_clone) -- one host thread per context, contigs handed out longest first.  Contigs are independent in the
 * reference (all per-contig state is cleared at GSAlign.cpp:490), so results do not depend on n_ctx.
 * on_result(user, contig, res) is called by the worker that finished `contig`, possibly from several threads at once;
 * *res is valid during the callback only.  Returns the first error (the failing context holds the text). */
typedef int (*gsa_result_fn)(void *user, int32_t contig, const gsa_result *res);
#define GSA_MANY_IN_ORDER 1u   /* hand the contigs out in the order given (default: longest first) */
#define GSA_MANY_DEVICE   2u   /* query[] are device pointers (gsa_align_contig_device); all contexts on one GPU */
#define GSA_MANY_NO_SPLIT 4u   /* never seed one contig on several contexts (see below) */
#define GSA_MANY_NO_BUNDLE 8u  /* never align several short contigs in one pass (see gsa_align_bundle) */
#define GSA_MANY_NO_PREFETCH 16u /* upload every contig when its turn comes (default: a context uploads its next contig while it aligns the current one) */
/* With FEWER contigs than contexts (one chromosome, two GPUs: BASELINE configs[3]) the contexts are dealt out in groups, one
 * group per contig, sized by contig length, and a contig of at least 20 Mb (gsa_set_option "split_min") is seeded by chunk range on all
 * contexts of its group -- gsa_seed_chunks ... gsa_finish_contig below, driven from the library's own threads, hits moved
 * device to device (peer to peer between GPUs).  Results do not depend on the grouping. */
int gsa_align_many(gsa_ctx *const *ctx, int32_t n_ctx, const char *const *query, const int32_t *qlen, int32_t n,
                   uint32_t flags, gsa_result_fn on_result, void *user);

/* Several contigs in ONE pass.  The reference clears all per-sequence state between query sequences (GSAlign.cpp:483-490), so
 * contigs are independent; a short contig costs the GPU ~60 operations whatever its size, so short contigs are concatenated
 * (each padded with N to a 10 000-bp chunk edge: IdentifyLocalMEM's chunk grid restarts per sequence, GSAlign.cpp:61-94) and go
 * through all stages together; seed groups never span contigs, AlnBlockVec is kept per contig.  out[k] is EXACTLY what
 * gsa_align_contig(query[k]) returns -- positions, record and string offsets relative to contig k -- and stays valid until the
 * next call on ctx.  n <= 4096, total length < 2^31.  flags: GSA_MANY_DEVICE (query[] are device pointers on ctx's GPU).
 * gsa_align_many bundles contigs of at most 16 Mb by itself (gsa_set_option "bundle_contig"; bundles of about "bundle_cap" = 64 Mb at most). */
int gsa_align_bundle(gsa_ctx *ctx, const char *const *query, const int32_t *qlen, int32_t n, uint32_t flags, gsa_result *out);

/* ---- one long contig on several GPUs ---------------------------------------
 * IdentifyLocalMEM hands 10 000-bp chunks of the contig to whichever thread is free (GSAlign.cpp:61-94) and seeds never
 * cross a chunk edge, so the seed search of one contig splits by chunk range: every GPU runs gsa_seed_chunks on its range
 * [chunk_beg, chunk_end) of the SAME contig, the hits (sort key + length/rank word per located hit) travel to the GPU
 * that owns the contig (gsa_export_hits -> any transport -> gsa_import_hits; buffers may be host or device memory), and
 * the owner runs the rest -- SeedGrouping ... GenerateFragAlignment -- with gsa_finish_contig.  The result is the one
 * gsa_align_contig gives (hit order does not matter: the seeds are sorted next). */
int gsa_seed_chunks(gsa_ctx *ctx, const char *query, int32_t qlen, int32_t chunk_beg, int32_t chunk_end);
int64_t gsa_hit_count(gsa_ctx *ctx);
int gsa_export_hits(gsa_ctx *ctx, uint64_t *keys, uint32_t *vals);
int gsa_import_hits(gsa_ctx *ctx, const uint64_t *keys, const uint32_t *vals, int64_t n);
/* the hits of gsa_seed_chunks where they lie (device pointers, valid until the next call on ctx): same-node transport */
int gsa_hit_buffers(gsa_ctx *ctx, const uint64_t **keys, const uint32_t **vals);
int gsa_finish_contig(gsa_ctx *ctx, gsa_result *out);

/* ---- stage-level entry points (what the parity tests drive) --------------
 * gsa_set_query uploads a contig; gsa_run_to(stage) advances the same
 * eight-stage sequence the oracle uses:
 *  1 IdentifyLocalMEM + SeedGrouping            GSAlign.cpp:51-107,126-143,492-495
 *  2 GenerateAlignmentBlocks                    GSAlign.cpp:305-391,497
 *  3 CheckAlnBlockOverlaps                      ProcessCandidateAlignment.cpp:189-239
 *  4 CheckAlnBlockLargeGaps + RemoveBadAlnBlocks  :120-156,72-79 ; KmerAnalysis.cpp:78-121
 *  5 CheckAlnBlockSpanMultiSeqs + RemoveBadAlnBlocks  :81-118
 *  6 EstChromosomeSimilarity + RemoveRedundantAlnBlocks(1),(2)   GSAlign.cpp:393-471
 *  7 FillAlnBlockGaps                           ProcessCandidateAlignment.cpp:241-276
 *  8 GenerateFragAlignment + identity filter    ProcessCandidateAlignment.cpp:290-351, GSAlign.cpp:523-540
 */
/* `query` is uploaded asynchronously: the buffer must stay valid and unmodified until the next synchronising call on this
 * context returns (gsa_run_to, gsa_seed_chunks, gsa_align_contig) -- a loader may not refill a pinned buffer earlier. */
int gsa_set_query(gsa_ctx *ctx, const char *query, int32_t qlen);
int gsa_set_query_device(gsa_ctx *ctx, const char *d_query, int32_t qlen);   /* see gsa_align_contig_device */
/* back to stage 0 with the contig gsa_set_query uploaded (no new copy): the same alignment can be run again, e.g. with
 * other parameters (the reference re-reads the contig from its FASTA buffer: GSAlign.cpp:483) */
int gsa_rewind(gsa_ctx *ctx);
int gsa_run_to(gsa_ctx *ctx, int stage);

/* stage 1 outputs: SeedVec in final order, and the SeedGrouping ranges */
int64_t gsa_seed_count(gsa_ctx *ctx);
int  gsa_get_seeds(gsa_ctx *ctx, gsa_seed *out);                   /* out[gsa_seed_count] */
int  gsa_group_count(gsa_ctx *ctx);
int  gsa_get_groups(gsa_ctx *ctx, int32_t *beg, int32_t *end);
/* stage >= 2 outputs: the current AlnBlockVec */
int  gsa_get_blocks(gsa_ctx *ctx, gsa_result *out);

/* ---- leaf operators ------------------------------------------------------ */
/* BWT_Search (bwt_search.cpp:141-185) for a batch of (start, stop) windows of
 * the current query: out_len[i] = match length, out_freq[i] = accepted hit
 * count (0 if rejected), out_loc[100*i ..] = located positions. */
int gsa_bwt_search_batch(gsa_ctx *ctx, int32_t n, const int32_t *start, const int32_t *stop,
                         int32_t *out_len, int32_t *out_freq, int64_t *out_loc);

/* ksw2_alignment (ksw2_alignment.cpp:251-273) for a batch of fragment pairs:
 * s1 = reference-side fragment (length m), s2 = query-side fragment (length n),
 * given as offsets into two byte pools.  ops receives, per pair, the forward
 * M/D/I string ('D' = gap in s1, 'I' = gap in s2) at ops_off[i]; ops_len[i]
 * <= m+n.  Caller sizes ops as sum(m+n). */
int gsa_ksw2_batch(gsa_ctx *ctx, int32_t n_pairs,
                   const char *pool1, const int64_t *off1, const int32_t *len1,
                   const char *pool2, const int64_t *off2, const int32_t *len2,
                   char *ops, const int64_t *ops_off, int32_t *ops_len);

/* CalGapSimilarity (KmerAnalysis.cpp:78-121) on the current query, batch */
int gsa_gap_similarity_batch(gsa_ctx *ctx, int32_t n, const int32_t *q1, const int32_t *q2,
                             const int64_t *r1, const int64_t *r2, int32_t *similar);

/* ---- measurement ---------------------------------------------------------
 * Event counters and HIP-event timings of the last gsa_align_contig /
 * gsa_run_to sequence.  counters: [0] Occ blocks read by seed extension,
 * [1] LF steps, [2] located hits, [3] seeds, [4] DP cells, [5] DP jobs,
 * [6] sum(m+n) over DP jobs, [7] Occ blocks actually read incl. speculative walks
 * ([0] counts what the reference's walk reads = the algorithmic figure).  kernel_ms: per-kernel-family device
 * time measured with hipEvents on the library's stream:
 * [0] seed search [1] locate [2] sorts [3] chaining (S2) [4] refinement (S3-S6)
 * [5] DP + string materialisation [6] total device time [7] host list logic. */
int gsa_get_counters(gsa_ctx *ctx, uint64_t counters[8]);
int gsa_get_timings(gsa_ctx *ctx, float kernel_ms[8]);
/* Host wall clock the calling thread spent inside the library per phase, summed over the contigs aligned since the last
 * gsa_set_profiling: ms[0] query set-up (upload, or the wait for a prefetched one), ms[s] stage s = 1..8 of gsa_run_to (a stage ends
 * where the host has to look at a count: ms[1] seed search incl. its read-back ... ms[8] DP + strings + the results' D2H);
 * *n = contigs; on the context that owns the index, ms[9] / *n = mean duration of one query upload (the Uploader's copies, one at a time).
 * Always on (nine clock reads per contig). */
int gsa_get_wall_sums(gsa_ctx *ctx, double ms[10], int64_t *n);
/* What growing its buffers has cost this context since it was created: host wall time inside hipMalloc / hipHostMalloc / the frees (and the wait for
 * the context's streams in front of a free), number of allocations, bytes allocated.  A context allocates when it meets a larger contig than it has
 * seen (the reference's vectors grow the same way inside GenomeComparison, GSAlign.cpp:473-552): a steady-state loop shows no growth here. */
int gsa_get_alloc_stats(gsa_ctx *ctx, double *ms, int64_t *n, int64_t *bytes);
/* diagnosis: the `top` largest device buffers of the context to stderr (name, bytes held, bytes last asked for) */
int gsa_debug_buffers(gsa_ctx *ctx, int top);
/* flags: bit 0 = per-stage hipEvent timing; bit 1 = run the ACCOUNTING build of the seed kernel,
 * which also records, per search, how many Occ blocks the reference's walk reads, so that counters[0]
 * is exact (same seeds either way; the default build leaves counters[0] = 0); bit 2 = time the seed
 * search kernel only (kernel_ms[0]; two events per contig instead of ten); kernel_ms[6] is then the SUM of that time over
 * all contigs since the flag was set (what a benchmark divides by its contig count). */
int gsa_set_profiling(gsa_ctx *ctx, int flags);
/* How the seed search of the last contig went (IdentifyLocalMEM, GSAlign.cpp:51-107):
 * [0] most resolver rounds of a chunk, [1] chunks redone by the dense search (every start position in parallel: the
 * `freq > MaxSeedFreq` reject-and-restart regime of bwt_search.cpp:177-182, or -sen), [2] most wave-iterations of a chunk,
 * [3..5] slowest chunk: round 1 / resolver / up to the marks, in 10 ns ticks; [6..7] reserved. */
int gsa_get_seed_stats(gsa_ctx *ctx, uint64_t stats[8]);

#ifdef __cplusplus
}
#endif
#endif
