// gsalign_amd/csrc/gsa_ctx.h -- the context behind the opaque gsa_ctx handle.
#ifndef GSA_CTX_H
#define GSA_CTX_H
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include "gsa_internal.h"

// A leaf = a maximal run of seeds of one S2 block that neither S4 (large gaps)
// nor S5 (reference chromosome ends) cuts.  Host-side block bookkeeping works on
// ranges of leaves.
struct Leaf {
	i32 beg, end;        // seed range in the refined seed arrays
	i32 sumlen;          // sum of (trimmed) seed lengths
	i32 q_first, q_last_end;
	i64 r_first, r_last_end;
	i32 blk;             // S2 block this leaf belongs to
	i32 cut4, cut5;      // how this leaf starts: S4 cut / S5 cut (0/0 = block start)
	i32 blk_score;       // AddAlnBlock score of that S2 block
};

// Device "mailbox" (i32[MAIL_N]) of the counts the stages produce; the host reads the whole
// box in ONE pinned copy where it needs them instead of one read-back per count.
enum { M_NB = 0, M_NC = 1, M_NBLK = 2, M_NG = 3, M_NR = 4, M_NJ = 5, M_NL = 6, M_LBERR = 7, M_ANY = 8, M_NTINY = 40, M_TICKET = 48, M_NBRAW = 49, M_NR2 = 50, M_NF = 51, M_NJOB = 52, M_OPSTOT = 53, M_NALN = 54, M_DPERR = 64, M_NLARGE = 65, M_DPERR2 = 66, M_CELLS = 68 /* two u64: sum m*n, sum m+n */ /* 64..71: cleared together, one aligned 32-byte fill */, M_DPERR3 = 41, M_LBDONE = 42, M_LFSTEPS = 44 /* u64, accounting build */, M_NEARLY = 62, M_EOPS = 63, M_NTOUCH = 46 /* touched PosDiff-bitmap blocks (OpPdTouched) */, M_MAXBLK = 56 /* u64 {score, number} of the best-scoring stage-2 block */, MAIL_N = 72 };
#define LEAF_CHUNK 1024      // leaves copied together with the mailbox (more -> a second copy)

struct HostBlock {       // one entry of the reference's AlnBlockVec, as leaf range
	i32 leaf_beg, leaf_end;
	i32 score;
	i32 bdup;
	// filled at stage 8
	i32 aln_len, bdir, gpos, chr;
};

// A query slot: the device copy of a contig, or of a bundle's concatenation with its tables.  A context has two: while the contig
// of one slot goes through the stages, the next one is uploaded into the other on `stream_up` (gsa_prefetch_contig /
// gsa_prefetch_bundle; gsa_align_many does it by itself), so the H2D copy of a contig -- 4.8 ms for 250 Mb at 52 GB/s, the reference
// reads QueryChrVec[i].seq from host memory every iteration (GSAlign.cpp:483-490) -- hides behind its predecessor's stages.
struct QuerySlot {
	DevBuf d_query;                                // contig / concatenation
	DevBuf d_bndtab, p_bndtab;                     // bundle tables: off[n + 1] | chunk_contig[chunks] | (device-resident contigs: source pointers) -- device / pinned staging
	std::vector<i32> b_off, b_qlen;                // start of contig k in the concatenation [n + 1], its length
	std::vector<const char *> src;                 // the host buffers this slot was filled from (the identity of a prefetch)
	i32 n = 0;                                     // 0: one contig, > 0: a bundle of n
	i32 lmax = 0; i64 tot = 0; size_t o_cc = 0, o_src = 0;
	bool pending = false;                          // uploaded (or on its way) and not yet adopted by gsa_set_query / set_query_bundle
	std::atomic<int> busy{0};                      // uploads of this slot the Uploader has not finished yet
	std::atomic<int> up_err{0};                    // first hipError_t of an upload into this slot (the Uploader thread sets it, slot_finish turns it into GSA_ERR_HIP)
};

// Uploads of query sequences, ONE AT A TIME per device index, in the order they were asked for (a thread + a copy stream; the contexts
// that share an index -- gsa_clone -- share it).  Why not a copy stream per context: copies that run side by side share the link, so
// every one of them takes as long as all of them together; four contexts that each prefetch their next contig then all get it late, at
// the same moment, and stay in lockstep (measured: the host waited 5 ms per contig for an upload issued a whole contig earlier, and the
// upload-inclusive rate was 27.5 Gbp/s against 31.5 with resident contigs).  First come first served, a 250 MB contig lands 4.4 ms after
// its turn and the contexts stagger.  Nothing but DMA copies runs on the stream and only host threads wait for them: a GPU-side wait
// (an event behind the copy, a kernel queued behind it) is a barrier packet that blocks a shared hardware queue for milliseconds.
struct Uploader {
	struct Piece { void *dst; const void *src; size_t n; };
	struct Job { std::vector<Piece> pieces; std::atomic<int> *busy; std::atomic<int> *err; std::chrono::steady_clock::time_point t_push; };
	double copy_ms = 0, wait_ms = 0, bytes = 0; long long jobs = 0;      // (statistics, uploader thread only)
	int device = 0; hipStream_t st = nullptr;
	std::thread th; std::mutex mu; std::condition_variable cv; std::deque<Job> q; bool stop = false;
	size_t n_demand = 0;                           // the first n_demand jobs of q are uploads a context is waiting for (slot_demand), in the order they asked
	void run();
	void push(Job &&j) { j.busy->fetch_add(1); j.t_push = std::chrono::steady_clock::now(); { std::lock_guard<std::mutex> g(mu); q.push_back(std::move(j)); } cv.notify_one(); }
};

// What gsa_set_option sets (include/gsa_hip.h); a clone starts with its parent's values.  The library reads NO environment variable
// (experiment switches exist only in builds with -DGSA_EXPERIMENTS).
struct Options {
	int64_t split_min = 20000000;      // gsa_align_many: a contig of at least this many bases may be seeded by chunk range on several contexts
	int64_t bundle_contig = 16000000;  // ... contigs up to this length travel in bundles (0: never)
	int64_t bundle_cap = 64000000;     // ... of about this many bases at most
	int dp_lane = 512;                 // alignments of at most this many cells go one per lane (k_dp_lane); 0: round 2's tiny / small split
	int seed_mode = 1;                 // 0 sweep: every chunk through k_dense_sweep; 1: the speculative kernel + dense kernels for what it gives up on; 2: round 2's k_dense_search in place of the sweep
	int pd_bitmap = 1;                 // 0: groups by the PosDiff sort although MaxIndelSize <= 31 would allow the bitmap scan
	int sweep_shape = -1;              // k_dense_sweep's launch shape: -1 by the number of dense chunks, 0 = four chunks per two-wave workgroup / 160-start segments, 1 = one chunk per four-wave workgroup / 40-start segments
	int dp_small_side = 0;             // 1: k_dp_small on a stream of its own (stream_aux[3]) so that the late striped launch starts beside it instead of behind it on the caller's stream (experiment; results do not depend on it)
	int dp_side = 0;                   // 1: the striped DP's lower size class on a stream of its own, beside the upper class (0: behind it)
	int64_t walk_chain_min = 100000;   // contigs with more seeds than this walk their window chain in slices (k_walk_chain) instead of one workgroup's LDS (k_walk_windows); tests: 0
	int64_t pd_two_level_min = 2000000;   // PosDiff bitmaps of more blocks than this (a reference above ~1 Gbp) are scanned in two passes: list the touched blocks, count those (tests: 0)
	int dp_occupancy = 0;              // > 0: at most this many striped-DP workgroups per CU (LDS padding): leaves wave slots for the passes beside it (experiment; 0 = off)
	int pd_bytes = 1;                  // the PosDiff bitmap of a contig (bundle) whose hits scatter (-sen: thousands of chance hits per chunk) is filled through a byte per value, plain stores, and packed
	                                   // afterwards -- no device-scope atomics (k_pd_pack, k_seed.hip); 1 = when the hit count says so, 0 = never, 2 = always (tests)
	int pres_from_kmer = 1;            // the presence table is derived from the k-mer jump table when both hold k-mers of one length (0: always from a scan of the text; a test compares the two)
	int kmer_k = 0;                    // gsa_create_opts (GSA_CREATE_KMER_K): length of the jump table's k-mers (0: by text length and free memory)
};

struct gsa_ctx {
	int device = 0;
	Options opt;
	Uploader *up = nullptr; bool own_up = false;
	hipStream_t stream = nullptr;
	hipStream_t stream_seed = nullptr;      // (experiment, GSA_SEED_CUS: the seed-search kernels on a stream restricted to part of the CUs)
	hipEvent_t ev_seed_fork = nullptr;
	u64 seed_ticket = 0;                    // value of the seed kernel's ticket counter (d_cnt[16]) before the next launch
	int n_cus = 0;
	hipStream_t stream_aux[4] = {nullptr, nullptr, nullptr, nullptr};   // [3]: the striped DP's lower size class beside its upper one (option dp_side); [0] early striped DP, [1] tiny DP + strings + sums, [2] records to the host (four streams in all: one per hardware queue)
	std::string err;
	Params prm;
	DevIndex di;
	gsa_ctx *index_owner = nullptr;                // gsa_clone: the context whose device index this one borrows (nullptr = own)
	gsa_ctx *lender = nullptr;                     // gsa_clone: the context `di` was copied from -- its presence bitmap / short k-mer table are read through that copy
	std::atomic<int> n_borrowers{0};               // live clones made FROM this context (gsa_set_params must not rebuild the tables they read)
	int prio_mode = 0;                             // GSA_CREATE_PRIO: stream priorities (0: none; see ctx_private_init)
	bool force_wide = false;                       // GSA_CREATE_WIDE: 64-bit dense SA + 32-byte k-mer entries whatever the text length (the >= 2^32-row layout)
	bool profiling = false;
	bool prof_seed = false;                        // time the seed kernel only (two events instead of ten per contig)
	bool count_blocks = false;                     // run the accounting build of the seed kernel (exact algorithmic Occ-block count)
	u64 dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	hipEvent_t ev[28];
	float kernel_ms[8];
	double wall_ms[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; int64_t wall_n = 0;      // host wall clock spent in [0] the query set-up (upload / wait for the prefetch) and [s] stage s of gsa_run_to, summed over contigs since gsa_set_profiling (gsa_get_wall_sums)
	double acc_seed_ms = 0.0;                      // seed-search kernel time summed over the contigs since gsa_set_profiling (flag bit 2), see gsa_get_timings
	u64 counters[8];

	// index (device)
	DevBuf d_bwt, d_bwt_ref, d_occ_base, d_sa, d_ref, d_chr_end, d_chr_of_end;
	std::vector<i64> h_chr_end, h_chr_fwd; std::vector<i32> h_chr_of_end, h_chr_len;
	i64 G = 0;

	// query
	QuerySlot qs[2]; int q_cur = -1;               // q_cur: slot of the current contig; -1: the caller's device buffer / none; -2: overwritten by a prefetch (no gsa_rewind)
	i32 qlen = 0; int stage = 0;
	const uint8_t *q_dev = nullptr;                // the contig on the device: d_query (uploaded by gsa_set_query) or the caller's buffer (gsa_set_query_device)
	bool split = false; i64 rng_beg = 0, rng_end = 0;     // gsa_seed_chunks: stage 1 on a chunk range only, the hits of other ranges are imported
	int qbits = 1, pdbits = 1;
	i64 pd_span = 0;                               // number of PosDiff key values: 2G + qlen + 2, a bundle: contigs x stride

	// ---- a bundle of contigs in one pass (Bundle, gsa_internal.h; gsa_align_bundle in gsa_api.hip) ----
	Bundle bnd = { 0, 0, 0, nullptr, nullptr }; bool bundle_call = false;
	std::vector<i32> b_off, b_qlen;               // start of contig k in the concatenation [n + 1], its length (copied from the slot that holds the bundle)
	std::vector<std::vector<HostBlock> > b_lists;  // the AlnBlockVec of every contig while the list logic runs (stages 3-6)
	std::vector<i32> b_blk0;                       // first block of contig k in the joined final list [n + 1]
	DevBuf d_bblk, p_bblk, p_ba0;                          // contig of every final block | first block per contig (device); string-pool offset of every contig (pinned, written by k_bundle_rebase)
	std::vector<i32> b_nblk; std::vector<i64> b_frag0;      // blocks that survive the identity filter, and the first record, per contig

	// scratch for rocPRIM
	DevBuf tmp;
	DevBuf leaf[9];                                // device staging of the leaf operators' batches (kept: a batch call allocates nothing once warm)
	// device counters block (u64[16]) + pinned host mirror
	DevBuf d_cnt; u64 *h_cnt = nullptr;
	DevBuf d_mail; i32 *h_mail = nullptr;          // count mailbox + pinned mirror
	DevBuf d_lb_status[2]; u32 lb_epoch = 0, lb_base = 0;   // look-back scan state (gsa_scan.h); the ticket counter sits in the mailbox
	DevBuf p_leaf;                                 // pinned landing zone for the leaf table
	int ev_pending = 0;                            // bit0: stage-1 events, bit1: stage-2 events not yet read
	bool s2_host = false;                          // n_b/n_c/n_blocks2/h_blk_* fetched for the stage-2 view

	// ---- stage 1 ----
	DevBuf d_ref2;                                 // 2-bit packed reference text
	DevBuf d_kmer;                                 // top-of-tree jump table
	DevBuf d_kmer_lo;                              // its short companion (MinSeedLength bases), see DevIndex::kmer_lo
	DevBuf d_pres;                                 // MinSeedLength-mer presence bitmap
	DevBuf d_sa_dense;                             // one SA entry per BWT row (built at gsa_create)
	DevBuf d_cand_s, d_cand_len, d_cand_x0, d_cand_freq, d_onpath, d_cand_cnt;
	size_t cand_cap_per_chunk = 1536;
	DevBuf d_heavy, dn_lf, dn_x0;               // chunks the speculative seed kernel gave up on; next(s) / accepted match per start of the dense chunks
	bool seed_sweep_next = false, seed_sweep_probe = false; int seed_sweep_run = 0, seed_sweep_period = 8;   // the previous contig handed most chunks to the sweep: the next one starts there (k_seed.hip, stage1_seed)
	u32 seed_budget = 256;                         // wave-iterations a chunk may take in the speculative kernel before it goes to the dense path (GSA_SEED_BUDGET)
	DevBuf d_chunk_hits, d_chunk_base;             // located hits per chunk and their exclusive prefix   // memoised matches of the search kernel + on-path bits
	DevBuf d_key_a, d_key_b, d_val_a, d_val_b;     // sort ping-pong
	i64 n_seeds = 0; bool hits_sorted = false;     // the hits in d_key_a / d_val_a are in (qPos, rank) order (k_seed_select over the whole contig; not after an import)
	DevBuf s_q, s_len, s_r, s_gid;                 // seeds in (PosDiff,qPos) order + group id
	DevBuf d_flag, d_scan;                         // generic i32 flag / scan arrays (n+1)
	i32 n_groups = 0;
	bool pd_path = false, seed_view_ready = false, pdbm_dirty = true; i64 pd_words = 0;      // groups from the PosDiff bitmap (no PosDiff sort on the hot path)
	DevBuf w_j0; u32 walk_ticket = 0, walk_epoch = 0;   // window chain of large contigs (k_walk_chain): ticket counter + one entry word per slice, never reset -- the host passes the counter's value and the launch epoch
	DevBuf d_zero;                                 // 256 zero bytes: Bundle::off / chunk_contig of a context without a bundle
	DevBuf d_pdby; bool pd_bytes = false;          // a byte per PosDiff value (all zero between contigs), see Options::pd_bytes
	DevBuf d_pdcb;                                 // coarse bitmap: one bit per block of 32 words of d_pdbm (the blocks that hold a hit)
	DevBuf d_pdbm, d_gpre, d_key_c, d_val_c;      // bitmap of occupied PosDiff values, group starts below each word, (group, qPos, rank) keys
	DevBuf g_beg;                                  // group start indices (n_groups+1)

	// ---- stage 2 ---- (arrays over seeds of active groups, (group,qPos,rPos) order)
	i64 n_a = 0;
	DevBuf a_q, a_len, a_r, a_gb, a_ge;            // + group begin/end index per seed
	DevBuf a_uniq, a_cu, a_alive, a_ws, a_wid;
	DevBuf a_next, a_brk, a_aurank, a_aulist, a_runinfo;
	DevBuf w_best, w_sum, w_n;
	DevBuf d_btab;                                 // (window, bucket) -> count hash table of the outlier filter
	DevBuf d_flag2, d_scan2, d_i64a;
	i64 n_b = 0, n_c = 0;
	DevBuf b_q, b_len, b_r, b_gb, b_ge;            // after compaction #1
	DevBuf c_q, c_len, c_r, c_gb, c_ge;            // after compaction #2
	DevBuf c_bid;                                  // S2 block id per seed (-1 = none)
	i32 n_blocks2 = 0;
	DevBuf blk_beg, blk_end, blk_score;            // S2 blocks kept by AddAlnBlock
	std::vector<i32> h_blk_beg, h_blk_end, h_blk_score;

	// ---- stages 3-5 ---- refined seed arrays (after RemoveOverlaps)
	i64 n_r = 0;
	DevBuf r_q, r_len, r_r, r_bid, r_tmp_q, r_tmp_len, r_tmp_r, r_tmp_bid;
	DevBuf r_cut4, r_cut5, r_simjob, r_simres;
	DevBuf d_leaf; std::vector<Leaf> h_leaf;
	std::vector<HostBlock> blocks;                 // current AlnBlockVec
	std::vector<i32> h_r_q, h_r_len; std::vector<i64> h_r_r;    // host copies of refined seeds (for getters / emit)
	bool have_host_seeds = false;

	// ---- stages 7-8 ----
	i64 n_frags = 0, n_aln = 0;                    // (n_frags < 0: still in the mailbox, see frags_count())
	i32 n_large = 0;                               // large DP jobs of the current contig (their records are patched on the host)
	DevBuf p_jpatch; i32 n_jobs = 0;               // pinned: record numbers, then string lengths, of the DP jobs of the job list (the host patches its records)
	DevBuf d_tail, p_tail;                         // final mailbox | patch list | string pool 1 | string pool 2: device / pinned (one copy at the end)
	const i32 *h_tmail = nullptr, *h_tpatch = nullptr; char *h_taln1 = nullptr, *h_taln2 = nullptr;      // the parts of p_tail
	i64 nf_ub = 0, span_ub = 0;                    // host-known upper bounds: records, and bases in gaps (ops / gapped strings)
	DevBuf fb_seedbase, fb_sbeg, fb_fragbase;      // per final block
	DevBuf f_rec;                                  // gsa_frag records (device working set)
	DevBuf f_rec16;                                // the same as 16-byte gsa_rec: what goes to the host
	DevBuf f_type, f_mism, f_alnlen, f_job, f_score;
	DevBuf j_frag, j_opsoff, j_nops, d_ops, j_cells;
	DevBuf d_dp_tiny;                              // order array of the four-per-wavefront DP kernel
	DevBuf d_dp_arena;                             // k_dp_lane: direction nibbles of the alignments in flight (per-wave regions)
	DevBuf d_dp_bnd, d_dp_ctr, d_dp_jobs, d_dp_large;   // striped DP: boundary granules, tickets, job descriptors, (job,m,n) of the large jobs
	// large DP gaps are known once the leaf table exists: they are launched there (stream_aux[0]) and run under stages 6-7
	DevBuf e_id, e_rec, e_list, e_off1, e_off2, e_opsoff, e_nops, e_ops, e_rev, r_head, f_early, r_orig, r_tmp_orig, p_early;
	bool early_consumed = false;                   // stage 7 enqueued its wait for the early launch (else gsa_run_to waits before it returns)
	bool early_listed = false;                     // stage 2 left the list of large gaps on its way to the host (event ev[16])
	i32 n_early = 0; bool early_in_flight = false; std::vector<i32> h_early;      // (seed, m, n) per early job
	bool dp_dirty = true;                          // ticket counters / error words of the striped DP need clearing (fresh buffer, or a failed launch)
	bool dp_timeout = false, dp_safe = false;      // a striped launch tripped its wait bound; repeat the contig with one job per launch
	int dp_fake_timeout = 0;                       // test hook (GSA_DP_FAKE_TIMEOUT=n at gsa_create): the next n contigs report a hand-off time-out once, so the retry paths run
	u32 dp_epoch = 0;                              // tag of the boundary granules of the current striped launch
	DevBuf p_dp, p_sj, p_sj_early;                 // pinned: mailbox + large-job list; stripe job tables (read by the kernels in place)
	DevBuf d_alnoff;
	DevBuf bl_alnlen, bl_score;
	std::vector<gsa_rec> h_frags; std::vector<gsa_block> h_blocks; std::vector<char> h_aln1, h_aln2;
	int frags_stage = 0;                           // stage for which h_frags/h_blocks were built
	// stage-8 results land in pinned host memory (one async D2H each, no pageable staging)
	DevBuf p_frags, p_blk; bool result_pinned = false;
	// what growing buffers cost this context (dev_ensure / pin_ensure: hipMalloc, hipHostMalloc, the frees and the quiesce in front of them) -- gsa_get_alloc_stats
	double alloc_ms = 0; long long alloc_n = 0, alloc_bytes = 0;
};

// A buffer is about to be freed: nothing of this context may still use it -- not only the main stream: the early striped DP launch runs on
// stream_aux[0] beside stages 3-7 and shares direction / boundary / ticket buffers with the late launch, strings and record copies run on
// stream_aux[1] / [2].  (Until round 4 only the main stream was waited for: a late launch that outgrew a shared buffer while the early one
// was still running freed it under the kernel -- a rare "memory access fault", seen once in a four-context run.)
static inline void ctx_quiesce(gsa_ctx *c)
{
	hipStreamSynchronize(c->stream);
	for (int i = 0; i < 4; i++) if (c->stream_aux[i]) hipStreamSynchronize(c->stream_aux[i]);
}

void *dev_take_reserved(int device, size_t bytes, size_t *got);   // gsa_api.hip  (gsa_reserve_index: memory set aside for the dense SA / the k-mer table)
// exact = true: the index tables -- asked for once, never grown: no slack (the dense SA of a human index is 49 GB: half again was 25 GB of HBM and 0.1 s of gsa_create)
template <class T> static inline T *dev_ensure(gsa_ctx *c, DevBuf &b, size_t n, bool exact = false)
{
	size_t bytes = (n ? n : 1) * sizeof(T);
	b.len = bytes;
	if (bytes <= b.cap) return (T *)b.p;
	const auto t0_ = std::chrono::steady_clock::now();
	if (b.p) { ctx_quiesce(c); hipFree(b.p); b.p = nullptr; b.cap = 0; }
	size_t want = exact ? bytes + 256 : bytes + bytes / 2 + 256;      // (half again: a context that meets a somewhat larger contig or bundle than it has seen does not stop to reallocate)
	if (exact && bytes >= ((size_t)1 << 30)) { size_t got = 0; if (void *r = dev_take_reserved(c->device, want, &got)) { b.p = r; b.cap = got; return (T *)b.p; } }
	if (hipMalloc(&b.p, want) != hipSuccess) { gsa_fail(c, GSA_ERR_NOMEM, "hipMalloc"); return nullptr; }
	b.cap = want;
	c->alloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count(); c->alloc_n++; c->alloc_bytes += (long long)want;
	return (T *)b.p;
}

template <class T> static inline T *pin_ensure(gsa_ctx *c, DevBuf &b, size_t n)
{
	size_t bytes = (n ? n : 1) * sizeof(T);
	if (bytes <= b.cap) return (T *)b.p;
	const auto t0_ = std::chrono::steady_clock::now();
	if (b.p) { ctx_quiesce(c); hipHostFree(b.p); b.p = nullptr; b.cap = 0; }
	size_t want = bytes + bytes / 4 + 4096;
	if (hipHostMalloc(&b.p, want) != hipSuccess) { gsa_fail(c, GSA_ERR_NOMEM, "hipHostMalloc"); return nullptr; }
	b.cap = want;
	c->alloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count(); c->alloc_n++; c->alloc_bytes += (long long)want;
	return (T *)b.p;
}

// stage drivers (one per translation unit)
int build_dense_sa(gsa_ctx *c, u64 n_sa);   // k_seed.hip
int unpack_pac(gsa_ctx *c, const uint8_t *d_pac, i64 G, uint8_t *d_ref);   // k_seed.hip  (GSA_CREATE_REF_PAC: RefSequence from the .pac bytes, on the device)
int build_occ(gsa_ctx *c, const void *ref_layout, u64 n_blocks128);   // k_seed.hip: the device's Occ blocks from the reference's layout
int build_presence(gsa_ctx *c);             // k_seed.hip  (after MinSeedLength changed)
int stage1_seed(gsa_ctx *c);          // k_seed.hip
int stage1_import_hits(gsa_ctx *c, const u64 *keys, const u32 *vals, i64 n);   // k_seed.hip
int stage1_finish_split(gsa_ctx *c);  // k_seed.hip
int stage1_restore_pdbm(gsa_ctx *c);  // k_seed.hip  (the PosDiff bitmap again from the hits a finished stage 2 left in d_key_a)
int seed_view_sort(gsa_ctx *c);       // k_seed.hip  (PosDiff-sorted seeds + groups: stage-1 view, or front of stage 2 without the PosDiff bitmap)
int stage2_chain(gsa_ctx *c);         // k_chain.hip
int launch_early_dp(gsa_ctx *c);      // k_chain.hip  (striped DP for the large gaps listed at the end of stage 2)
int stage2_fetch_host(gsa_ctx *c);    // k_chain.hip  (counts + S2 block table for the stage-2 view)
void collect_events(gsa_ctx *c);      // gsa_api.hip  (deferred hipEventElapsedTime of stages 1-2)
int stage345_refine(gsa_ctx *c);      // k_refine.hip  (device part of S3, S4, S5 + leaf table)
int stage7_fill(gsa_ctx *c);          // k_extend.hip  (S6: gap records of the final block list)
i64 frags_count(gsa_ctx *c);          // k_extend.hip  (record count, fetched from the mailbox when still unknown)
int stage78_extend(gsa_ctx *c);       // k_extend.hip  (S7: classification, DP, gapped strings, block sums)
int run_gapsim_jobs(gsa_ctx *c, i32 n, const i32 *d_n, const i32 *d_q1, const i32 *d_q2, const i64 *d_r1, const i64 *d_r2, i32 *d_res, const i32 *d_jseed, i32 *d_cut4);   // k_gapsim.hip
void dp_count_cells(gsa_ctx *c, i32 n_ub, const i32 *len1, const i32 *len2, hipStream_t stream);   // k_dp.hip (profiling)
struct LgJob { i32 job, m, n; };
int launch_stripes(gsa_ctx *c, hipStream_t ss, std::vector<LgJob> &large, const uint8_t *pool1, const i64 *off1, const uint8_t *pool2, const i64 *off2,
                   uint8_t *ops, const i64 *ops_off, i32 *ops_len, uint8_t *rev, int err_slot);   // k_dp.hip
struct Ksw2Launch { i32 n = 0, nsmall = 0, nlarge = 0; bool small_in_flight = false; };      // what run_ksw2_jobs left running
int run_ksw2_jobs(gsa_ctx *c, i32 n_ub, const uint8_t *pool1, const i64 *off1, const i32 *len1,
                  const uint8_t *pool2, const i64 *off2, const i32 *len2, uint8_t *ops, const i64 *ops_off, i32 *ops_len, i64 ops_total, Ksw2Launch *out,
                  const i32 *jfrag = nullptr, gsa_frag *frag = nullptr, bool mail_clean = false);   // k_dp.hip  (jfrag/frag: the small kernels also set aln_len of the job's record;
                  // mail_clean: the caller's last pass already zeroed mail[M_DPERR .. M_DPERR + 7])

#endif
