// gsalign_amd/csrc/gsa_api.hip -- C-ABI entry points of libgsa_hip.so (include/gsa_hip.h):
// context life cycle, query upload, the eight-stage driver and the getters.
// Host-side block-list bookkeeping (what the reference does on AlnBlockVec with
// std::sort) lives in gsa_blocks.cpp.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>
#include <pthread.h>
#include <sched.h>
#include "gsa_ctx.h"

static thread_local std::string g_create_error;

int gsa_fail(gsa_ctx *ctx, int code, const std::string &msg)
{
	if (ctx) ctx->err = msg; else g_create_error = msg;
	return code;
}

int host_stage4_5_6(gsa_ctx *c, int stage);    // gsa_blocks.cpp
int host_stage8_finish(gsa_ctx *c);            // gsa_blocks.cpp
int build_block_view(gsa_ctx *c);              // gsa_blocks.cpp
void bundle_split_lists(gsa_ctx *c);           // gsa_blocks.cpp  (a bundle of contigs: one AlnBlockVec per contig for stages 4-6 ...
void bundle_join_lists(gsa_ctx *c);            // gsa_blocks.cpp   ... joined again, contig after contig, for stages 7-8)

void collect_events(gsa_ctx *c)
{
	if (!c->profiling && !c->prof_seed) { c->ev_pending = 0; return; }
	float ms;
	if ((c->ev_pending & 1) && !c->profiling) { if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) { c->kernel_ms[0] = ms; c->acc_seed_ms += ms; } c->ev_pending = 0; (void)hipGetLastError(); return; }
	if (c->ev_pending & 1) {
		if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->kernel_ms[0] = ms;
		if (hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) c->kernel_ms[1] = ms;
		if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) c->kernel_ms[2] = ms;
	}
	if (c->ev_pending & 2) { if (hipEventElapsedTime(&ms, c->ev[4], c->ev[5]) == hipSuccess) c->kernel_ms[3] = ms; }
	c->ev_pending = 0;
	(void)hipGetLastError();
}

void Uploader::run()
{
	(void)hipSetDevice(device);
	for (;;) {
		Job j;
		{ std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return stop || !q.empty(); }); if (q.empty()) return; j = std::move(q.front()); q.pop_front(); if (n_demand > 0) n_demand--; }
		const auto t0 = std::chrono::steady_clock::now();
		// a failed copy (a host pointer the runtime cannot read, a device fault) must not look like an upload: the slot keeps the first error
		// and the align call that adopts the slot fails with GSA_ERR_HIP (slot_finish)
		hipError_t e_first = hipSuccess;
		for (const Piece &p : j.pieces) if (p.n) { const hipError_t e = hipMemcpyAsync(p.dst, p.src, p.n, hipMemcpyHostToDevice, st); if (e != hipSuccess && e_first == hipSuccess) e_first = e; bytes += p.n; }
		{ const hipError_t e = hipStreamSynchronize(st); if (e != hipSuccess && e_first == hipSuccess) e_first = e; }
		(void)hipGetLastError();
		if (e_first != hipSuccess && j.err) { int none = 0; j.err->compare_exchange_strong(none, (int)e_first); }
		const auto t1 = std::chrono::steady_clock::now();
		{ std::lock_guard<std::mutex> g(mu); copy_ms += std::chrono::duration<double, std::milli>(t1 - t0).count(); wait_ms += std::chrono::duration<double, std::milli>(t0 - j.t_push).count(); jobs++; }
		j.busy->fetch_sub(1, std::memory_order_release);
	}
}

static int uploader_start(gsa_ctx *c)
{
	Uploader *u = new Uploader(); u->device = c->device;
	if (hipStreamCreateWithFlags(&u->st, hipStreamNonBlocking) != hipSuccess) { delete u; return gsa_fail(c, GSA_ERR_HIP, "hipStreamCreate (upload stream)"); }
	u->th = std::thread([u] { u->run(); });
	c->up = u; c->own_up = true;
	return GSA_OK;
}

static void uploader_stop(gsa_ctx *c)
{
	if (!c->up || !c->own_up) { c->up = nullptr; return; }
	Uploader *u = c->up;
	{ std::lock_guard<std::mutex> g(u->mu); u->stop = true; }
	u->cv.notify_all();
	if (u->th.joinable()) u->th.join();
#ifdef GSA_EXPERIMENTS
	if (getenv("GSA_UP_STATS") && u->jobs) fprintf(stderr, "[uploader] %lld jobs, %.1f MB each, %.2f ms in the queue, %.2f ms copying (%.1f GB/s)\n", (long long)u->jobs, u->bytes / 1e6 / u->jobs, u->wait_ms / u->jobs, u->copy_ms / u->jobs, u->bytes / 1e6 / u->copy_ms);
#endif
	if (u->st) hipStreamDestroy(u->st);
	delete u; c->up = nullptr;
}

static inline void slot_wait(QuerySlot &s) { while (s.busy.load(std::memory_order_acquire) > 0) std::this_thread::yield(); }
// A context needs this slot NOW: its upload, if it has not started, goes to the front of the queue (uploads somebody waits for before
// uploads of contigs whose turn comes later: at the start of a run every context has two contigs in the queue and the GPU has nothing)
static void slot_demand(gsa_ctx *c, QuerySlot &s)
{
	if (s.busy.load(std::memory_order_acquire) > 0 && c->up) {
		Uploader *u = c->up;
		std::lock_guard<std::mutex> g(u->mu);
		for (size_t k = u->n_demand; k < u->q.size(); k++) if (u->q[k].busy == &s.busy) {
			Uploader::Job j = std::move(u->q[k]); u->q.erase(u->q.begin() + (long)k); u->q.insert(u->q.begin() + (long)u->n_demand, std::move(j)); u->n_demand++;
			break;
		}
	}
	slot_wait(s);
}

// what every context owns, shared index or not: streams, events, counters, mailbox
static int ctx_private_init(gsa_ctx *c, gsa_ctx *share = nullptr)
{
#ifdef GSA_EXPERIMENTS
	// experiment (GSA_STREAM_PRIO=1): the main stream (seed kernels, fused passes) above the DP streams in the dispatcher's eyes
	static const int prio = [] { const char *e = getenv("GSA_STREAM_PRIO"); return e ? atoi(e) : 0; }();
	if (prio) {
		int lo = 0, hi = 0; GSA_CHECK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));      // (lo = least, hi = greatest priority: numerically lo >= hi)
		GSA_CHECK(c, hipStreamCreateWithPriority(&c->stream, hipStreamDefault, hi));
		for (int i = 0; i < 4; i++) GSA_CHECK(c, hipStreamCreateWithPriority(&c->stream_aux[i], hipStreamDefault, (i == 0 || i == 3 || (prio > 1 && i == 1)) ? lo : hi));
	} else
#endif
	if (c->prio_mode > 0) {
		// GSA_CREATE_PRIO(mode): the MAIN stream -- the ~45 short passes of chaining / refinement / the extend stage's bookkeeping: little work, but every one of
		// them is a dispatch that has to find wave slots between the long kernels of the other contexts (measured with four contexts on the human index:
		// those phases take 6x their time alone, 11 of a contig's 19 ms: profiles/r05_contig_phases_human_full.txt) -- at the greatest priority; the seed-search
		// kernels move to a stream of their own (mode 1, 3: normal; mode 2: least) so that they do not inherit it; the striped DP's stream normal (mode 3: least)
		int lo = 0, hi = 0; GSA_CHECK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));      // (lo = least, hi = greatest priority: numerically lo >= hi)
		const int mid = (lo + hi) / 2;
		GSA_CHECK(c, hipStreamCreateWithPriority(&c->stream, hipStreamDefault, hi));
		GSA_CHECK(c, hipStreamCreateWithPriority(&c->stream_seed, hipStreamDefault, c->prio_mode == 2 ? lo : mid));
		GSA_CHECK(c, hipEventCreateWithFlags(&c->ev_seed_fork, hipEventDisableTiming));
		for (int i = 0; i < 4; i++) GSA_CHECK(c, hipStreamCreateWithPriority(&c->stream_aux[i], hipStreamDefault, (c->prio_mode == 3 && (i == 0 || i == 3)) ? lo : mid));
	} else {
	GSA_CHECK(c, hipStreamCreate(&c->stream));
	for (int i = 0; i < 4; i++) GSA_CHECK(c, hipStreamCreate(&c->stream_aux[i]));
	}
	if (share) { c->up = share->up; c->own_up = false; } else if (int rc = uploader_start(c)) return rc;      // (Uploader, gsa_ctx.h)
	for (int i = 0; i < 28; i++) GSA_CHECK(c, hipEventCreate(&c->ev[i]));
	GSA_CHECK(c, hipMalloc(&c->d_zero.p, 256)); c->d_zero.cap = 256; GSA_CHECK(c, hipMemset(c->d_zero.p, 0, 256));      // (the bundle tables of a context that holds no bundle: bundle_off_flat)
	GSA_CHECK(c, hipMalloc(&c->d_cnt.p, 32 * sizeof(u64))); c->d_cnt.cap = 32 * sizeof(u64); GSA_CHECK(c, hipMemset(c->d_cnt.p, 0, 32 * sizeof(u64)));      // (16 counters + the seed kernel's ticket counter)
	GSA_CHECK(c, hipHostMalloc((void **)&c->h_cnt, 16 * sizeof(u64)));
	GSA_CHECK(c, hipMalloc(&c->d_mail.p, MAIL_N * sizeof(i32))); c->d_mail.cap = MAIL_N * sizeof(i32);
	GSA_CHECK(c, hipMemset(c->d_mail.p, 0, MAIL_N * sizeof(i32)));
	GSA_CHECK(c, hipHostMalloc((void **)&c->h_mail, MAIL_N * sizeof(i32)));
	return GSA_OK;
}

// One large index array, host to device.  Pageable memory goes through the runtime's staging buffers at ~12 GB/s; page-locked in place first (hipHostRegister:
// ~10 ms per GB) the copy is one DMA transfer at link speed -- 5.4 GB of a human index: 0.45 s -> 0.17 s.  A buffer that cannot be registered (already registered
// by the host, file-backed, too large for the limit) is copied as it is.
static hipError_t h2d_big(void *dst, const void *src, size_t bytes)
{
	if (bytes >= ((size_t)64 << 20) && hipHostRegister(const_cast<void *>(src), bytes, hipHostRegisterDefault) == hipSuccess) {
		const hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
		(void)hipHostUnregister(const_cast<void *>(src));
		return e;
	}
	(void)hipGetLastError();
	return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
}

// Device memory set aside for the two largest index tables before the index files are even read (gsa_reserve_index): their sizes follow from the text length
// alone.  One slot per device; gsa_create on that device adopts what fits (dev_take_reserved), gsa_release_reserved frees what was not used.
struct ReservedBuf { void *p = nullptr; size_t bytes = 0; };
static std::mutex g_res_mu; static ReservedBuf g_res[64][2];
void *dev_take_reserved(int device, size_t bytes, size_t *got)
{
	std::lock_guard<std::mutex> g(g_res_mu);
	for (int k = 0; k < 2; k++) { ReservedBuf &r = g_res[device & 63][k]; if (r.p && r.bytes >= bytes && r.bytes <= bytes + bytes / 8 + 4096) { void *p = r.p; *got = r.bytes; r.p = nullptr; r.bytes = 0; return p; } }
	return nullptr;
}

extern "C" {

void gsa_default_params(gsa_params *p)
{
	// main.cpp:202-214
	p->min_seed_len = 15; p->max_indel = 25; p->min_block_score = 200; p->min_aln_len = 200; p->min_identity = 70;
	p->sensitive = 0; p->one_on_one = 0;
}

const char *gsa_last_error(gsa_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

// Back to stage 0: nothing of the previous run may still be in flight or half-consumed (the early striped DP launch
// writes e_* buffers the next stage 2 would reuse).
static int reset_run_state(gsa_ctx *c)
{
	if (c->early_in_flight) { GSA_CHECK(c, hipStreamSynchronize(c->stream_aux[0])); c->early_in_flight = false; }
	c->n_early = 0; c->early_listed = false; c->early_consumed = false;
	c->stage = 0; c->split = false; c->b_lists.clear();
	c->n_seeds = 0; c->n_groups = 0; c->n_blocks2 = 0; c->blocks.clear(); c->frags_stage = 0; c->have_host_seeds = false; c->ev_pending = 0; c->s2_host = false;
	memset(c->counters, 0, sizeof(c->counters)); memset(c->kernel_ms, 0, sizeof(c->kernel_ms));
	return GSA_OK;
}

int gsa_set_params(gsa_ctx *c, const gsa_params *p)
{
	if (!c || !p) return GSA_ERR_ARG;
	if (p->min_seed_len < 1 || p->max_indel < 0) return gsa_fail(c, GSA_ERR_ARG, "bad parameter");
	GSA_CHECK(c, hipSetDevice(c->device));
	// The presence bitmap and the short k-mer table depend on MinSeedLength and are read by every context cloned FROM this one
	// (gsa_clone copies the pointers): rebuilding them here -- in place, or freed and reallocated -- would hand those clones a
	// bitmap of another k or freed memory.  A clone that changes its own parameters builds tables of its own.
	if (c->n_borrowers.load() > 0 && (p->sensitive ? 10 : p->min_seed_len) != c->prm.MinSeedLength)
		return gsa_fail(c, GSA_ERR_STATE, "gsa_set_params: contexts cloned from this one read its seed tables -- change -slen / -sen on the clones, or destroy them first");
	if (int rc = reset_run_state(c)) return rc;
	c->prm.MinSeedLength = p->sensitive ? 10 : p->min_seed_len;         // main.cpp:323
	c->prm.MaxIndelSize = p->max_indel; c->prm.MinAlnBlockScore = p->min_block_score; c->prm.MinAlnLength = p->min_aln_len;
	c->prm.MinSeqIdy = p->min_identity; c->prm.bSensitive = p->sensitive ? 1 : 0; c->prm.OneOnOne = p->one_on_one ? 1 : 0;
	return build_presence(c);
}

int gsa_create(int device, const gsa_index_view *idx, const gsa_params *prm, gsa_ctx **out)
{
	return gsa_create_opts(device, idx, prm, 0u, out);
}

int gsa_create_opts(int device, const gsa_index_view *idx, const gsa_params *prm, uint32_t flags, gsa_ctx **out)
{
	if (!idx || !out || !idx->bwt || !idx->sa || !idx->ref || !idx->chr_len || idx->n_chr <= 0 || idx->G <= 0) return gsa_fail(nullptr, GSA_ERR_ARG, "gsa_create: bad index view");
	if (idx->L2[4] != (uint64_t)(2 * idx->G)) return gsa_fail(nullptr, GSA_ERR_ARG, "gsa_create: L2[4] != 2G");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return gsa_fail(nullptr, GSA_ERR_HIP, "no HIP device available (libgsa_hip.so has no CPU path)");
	if (device < 0 || device >= ndev) return gsa_fail(nullptr, GSA_ERR_ARG, "gsa_create: bad device ordinal");
	if (flags & ~(uint32_t)(GSA_CREATE_WIDE | GSA_CREATE_KMER_K(15) | GSA_CREATE_PRIO(3) | GSA_CREATE_REF_PAC)) return gsa_fail(nullptr, GSA_ERR_ARG, "gsa_create_opts: unknown flag");
	{ const uint32_t kk = (flags >> 8) & 15u; if (kk == 1) return gsa_fail(nullptr, GSA_ERR_ARG, "gsa_create_opts: GSA_CREATE_KMER_K takes 2 .. 15 (0: chosen by text length and free memory)"); }
	gsa_ctx *c = new gsa_ctx();
	c->device = device; c->force_wide = (flags & GSA_CREATE_WIDE) != 0; c->opt.kmer_k = (int)((flags >> 8) & 15u); c->prio_mode = (int)((flags >> 16) & 3u);
	memset(c->kernel_ms, 0, sizeof(c->kernel_ms)); memset(c->counters, 0, sizeof(c->counters));
#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { gsa_fail(nullptr, GSA_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); gsa_destroy(c); return GSA_ERR_HIP; } } while (0)
	CK(hipSetDevice(device));
	(void)hipSetDeviceFlags(hipDeviceScheduleSpin);      // host waits spin instead of sleeping: the pipeline has ~20 short count read-backs per contig
	(void)hipGetLastError();
	if (int rcp = ctx_private_init(c)) { g_create_error = c->err; gsa_destroy(c); return rcp; }
	const size_t bwt_bytes = ((idx->bwt_words + 15) / 16) * 64;          // whole 64-byte blocks of the reference's layout (regrouped below: build_occ)
	CK(hipMalloc(&c->d_bwt_ref.p, bwt_bytes + 64)); c->d_bwt_ref.cap = bwt_bytes + 64;      // (freed once regrouped)
	CK(hipMemset(c->d_bwt_ref.p, 0, bwt_bytes + 64));
	CK(h2d_big(c->d_bwt_ref.p, idx->bwt, idx->bwt_words * 4));
	CK(hipMalloc(&c->d_sa.p, idx->n_sa * 8)); c->d_sa.cap = idx->n_sa * 8;
	CK(h2d_big(c->d_sa.p, idx->sa, idx->n_sa * 8));
	CK(hipMalloc(&c->d_ref.p, (size_t)2 * idx->G + 64)); c->d_ref.cap = (size_t)2 * idx->G + 64;
	if (flags & GSA_CREATE_REF_PAC) {
		// idx->ref = the .pac bytes (four bases per byte, first base in the top bits: bntseq.c's _get_pac): RestoreReferenceInfo's loop (bwt_index.cpp:229-264) runs
		// on the device -- G / 4 bytes cross PCIe instead of 2G, and the host need not have unpacked anything before the table builds start
		const size_t pac_bytes = ((size_t)idx->G + 3) / 4;
		void *d_pac = nullptr;
		CK(hipMalloc(&d_pac, pac_bytes + 64));
		if (h2d_big(d_pac, idx->ref, pac_bytes) != hipSuccess) { hipFree(d_pac); gsa_fail(nullptr, GSA_ERR_HIP, "hipMemcpy (.pac)"); gsa_destroy(c); return GSA_ERR_HIP; }
		const int rcu = unpack_pac(c, (const uint8_t *)d_pac, idx->G, (uint8_t *)c->d_ref.p);
		(void)hipStreamSynchronize(c->stream); hipFree(d_pac);
		if (rcu) { g_create_error = c->err; gsa_destroy(c); return rcu; }
	} else
	CK(h2d_big(c->d_ref.p, idx->ref, (size_t)2 * idx->G));
	// ChrLocMap (bwt_index.cpp:240-253) as a sorted table of last coordinates
	c->G = idx->G;
	{
		i64 tot = 0; std::vector<std::pair<i64, i32> > ends;
		for (int i = 0; i < idx->n_chr; i++) {
			c->h_chr_len.push_back(idx->chr_len[i]); c->h_chr_fwd.push_back(tot); tot += idx->chr_len[i];
			ends.push_back(std::make_pair(c->h_chr_fwd[i] + idx->chr_len[i] - 1, i));
			ends.push_back(std::make_pair(2 * idx->G - tot + idx->chr_len[i] - 1, i));
		}
		if (tot != idx->G) { gsa_fail(nullptr, GSA_ERR_ARG, "gsa_create: sum(chr_len) != G"); gsa_destroy(c); return GSA_ERR_ARG; }
		std::sort(ends.begin(), ends.end());
		for (size_t i = 0; i < ends.size(); i++) { c->h_chr_end.push_back(ends[i].first); c->h_chr_of_end.push_back(ends[i].second); }
	}
	c->d_chr_end.cap = c->h_chr_end.size() * 8; c->d_chr_of_end.cap = c->h_chr_of_end.size() * 4;
	CK(hipMalloc(&c->d_chr_end.p, c->h_chr_end.size() * 8)); CK(hipMemcpy(c->d_chr_end.p, c->h_chr_end.data(), c->h_chr_end.size() * 8, hipMemcpyHostToDevice));
	CK(hipMalloc(&c->d_chr_of_end.p, c->h_chr_of_end.size() * 4)); CK(hipMemcpy(c->d_chr_of_end.p, c->h_chr_of_end.data(), c->h_chr_of_end.size() * 4, hipMemcpyHostToDevice));
#undef CK
	c->di.primary = idx->primary; for (int i = 0; i < 5; i++) c->di.L2[i] = idx->L2[i]; c->di.L2[0] = 0;
	c->di.seq_len = idx->L2[4];
	c->di.bwt = nullptr; c->di.occ_base = nullptr; c->di.occ_shift = 0; c->di.sa = c->d_sa.as<u64>(); c->di.ref = c->d_ref.as<uint8_t>(); c->di.G = idx->G;
	c->di.chr_end = c->d_chr_end.as<i64>(); c->di.chr_of_end = c->d_chr_of_end.as<i32>(); c->di.n_ends = (i32)c->h_chr_end.size();
	c->di.sa32 = nullptr; c->di.sa64 = nullptr; c->di.kmer = nullptr; c->di.kmer_k = 0; c->di.kmer_lo = nullptr; c->di.kmer_lo_k = 0; c->di.kmer_e16 = 0; c->di.ref2 = nullptr; c->di.pres = nullptr; c->di.pres_k = 0;
	{
		const int rco = build_occ(c, c->d_bwt_ref.p, bwt_bytes / 64);
		hipFree(c->d_bwt_ref.p); c->d_bwt_ref.p = nullptr; c->d_bwt_ref.cap = 0;
		if (rco) { g_create_error = c->err; gsa_destroy(c); return rco; }
	}
	if (int rcd = build_dense_sa(c, idx->n_sa)) { g_create_error = c->err; gsa_destroy(c); return rcd; }
	gsa_params dp; gsa_default_params(&dp);
	int rc = gsa_set_params(c, prm ? prm : &dp);
	if (rc) { g_create_error = c->err; gsa_destroy(c); return rc; }
	gsa_release_reserved(device);      // (a reservation that did not fit this index)
	if (const unsigned long long ov = gsa_take_grid_overflow()) { gsa_fail(nullptr, GSA_ERR_LIMIT, "gsa_create: a table build needed a launch of " + std::to_string(ov) + " work-items (>= 2^32)"); gsa_destroy(c); return GSA_ERR_LIMIT; }
	*out = c;
	return GSA_OK;
}

// every device buffer a context can own (gsa_destroy frees them; gsa_debug_buffers lists them)
#define GSA_DEVBUFS(X) X(d_bwt) X(d_bwt_ref) X(d_occ_base) X(d_sa) X(d_ref) X(d_chr_end) X(d_chr_of_end) X(qs[0].d_query) X(qs[1].d_query) X(qs[0].d_bndtab) X(qs[1].d_bndtab) X(tmp) X(d_cnt) X(d_zero) X(d_mail) X(d_lb_status[0]) X(d_lb_status[1]) X(d_sa_dense) X(d_kmer) X(d_kmer_lo) X(d_pres) X(d_ref2) X(d_cand_s) X(d_cand_len) X(d_cand_x0) X(d_cand_freq) X(d_onpath) X(d_cand_cnt) X(d_heavy) X(dn_lf) X(dn_x0) X(d_chunk_hits) X(d_chunk_base) X(d_key_a) X(d_key_b) X(d_val_a) X(d_val_b) X(s_q) X(s_len) X(s_r) X(s_gid) X(d_flag) X(d_scan) X(g_beg) X(w_j0) X(d_pdbm) X(d_pdby) X(d_pdcb) X(d_gpre) X(d_key_c) X(d_val_c) X(a_q) X(a_len) X(a_r) X(a_gb) X(a_ge) X(a_uniq) X(a_cu) X(a_alive) X(a_ws) X(a_wid) X(a_next) X(a_brk) X(a_aurank) X(a_aulist) X(a_runinfo) X(w_best) X(w_sum) X(w_n) X(d_btab) X(d_flag2) X(d_scan2) X(d_i64a) X(b_q) X(b_len) X(b_r) X(b_gb) X(b_ge) X(c_q) X(c_len) X(c_r) X(c_gb) X(c_ge) X(c_bid) X(blk_beg) X(blk_end) X(blk_score) X(r_q) X(r_len) X(r_r) X(r_bid) X(r_tmp_q) X(r_tmp_len) X(r_tmp_r) X(r_tmp_bid) X(r_cut4) X(r_cut5) X(r_simjob) X(r_simres) X(d_leaf) X(fb_seedbase) X(fb_sbeg) X(fb_fragbase) X(f_rec) X(f_rec16) X(f_type) X(f_mism) X(f_alnlen) X(f_job) X(f_score) X(d_dp_tiny) X(d_dp_bnd) X(d_dp_ctr) X(d_dp_jobs) X(d_dp_large) X(d_tail) X(e_id) X(e_rec) X(e_list) X(e_off1) X(e_off2) X(e_opsoff) X(e_nops) X(e_ops) X(e_rev) X(r_head) X(f_early) X(r_orig) X(r_tmp_orig) X(j_frag) X(j_opsoff) X(j_nops) X(d_ops) X(j_cells) X(d_alnoff) X(bl_alnlen) X(bl_score) X(d_bblk) X(d_dp_arena) X(leaf[0]) X(leaf[1]) X(leaf[2]) X(leaf[3]) X(leaf[4]) X(leaf[5]) X(leaf[6]) X(leaf[7]) X(leaf[8])
void gsa_destroy(gsa_ctx *c)
{
	if (!c) return;
	hipSetDevice(c->device);
	if (c->stream) hipStreamSynchronize(c->stream);
	for (int i = 0; i < 2; i++) slot_wait(c->qs[i]);
	uploader_stop(c);
	if (c->lender) c->lender->n_borrowers.fetch_sub(1);      // (`parent` outlives its clones: gsa_hip.h)
#define X(n) &c->n,
	DevBuf *bufs[] = { GSA_DEVBUFS(X) };
#undef X
	// (a gsa_clone context borrows the index through `di` only: its index DevBufs are empty, a presence bitmap it built after
	//  a parameter change is its own)
	for (DevBuf *b : bufs) if (b->p) hipFree(b->p);
	if (c->h_cnt) hipHostFree(c->h_cnt);
	if (c->h_mail) hipHostFree(c->h_mail);
	for (DevBuf *b : { &c->p_frags, &c->p_tail, &c->p_leaf, &c->p_blk, &c->p_dp, &c->p_sj, &c->p_sj_early, &c->p_jpatch, &c->p_early, &c->qs[0].p_bndtab, &c->qs[1].p_bndtab, &c->p_bblk, &c->p_ba0 }) if (b->p) hipHostFree(b->p);
	for (int i = 0; i < 28; i++) if (c->ev[i]) hipEventDestroy(c->ev[i]);
	for (int i = 0; i < 4; i++) if (c->stream_aux[i]) hipStreamDestroy(c->stream_aux[i]);
	if (c->stream_seed) hipStreamDestroy(c->stream_seed);
	if (c->ev_seed_fork) hipEventDestroy(c->ev_seed_fork);
	if (c->stream) hipStreamDestroy(c->stream);
	delete c;
}

// A second context on the same GPU that shares `parent`'s device-resident index (read-only: BWT + Occ, both SAs, k-mer
// table, packed and ASCII text, chromosome table) and owns everything else.  What the reference does with N pthreads
// on one contig, a host does here with N contexts on N contigs: while one contig sits in the dependency chain of its
// largest DP problem, the next one's seed search and chaining fill the idle CUs, and its upload overlaps both.
int gsa_clone(gsa_ctx *parent, gsa_ctx **out)
{
	if (!parent || !out) return GSA_ERR_ARG;
	if (hipSetDevice(parent->device) != hipSuccess) return gsa_fail(nullptr, GSA_ERR_HIP, "hipSetDevice");
	gsa_ctx *c = new gsa_ctx();
	c->device = parent->device; c->force_wide = parent->force_wide; c->prio_mode = parent->prio_mode;
	c->index_owner = parent->index_owner ? parent->index_owner : parent; c->seed_budget = parent->seed_budget; c->opt = parent->opt;
	memset(c->kernel_ms, 0, sizeof(c->kernel_ms)); memset(c->counters, 0, sizeof(c->counters));
	if (int rc = ctx_private_init(c, c->index_owner)) { g_create_error = c->err; gsa_destroy(c); return rc; }
	c->di = parent->di; c->G = parent->G;
	c->lender = parent; parent->n_borrowers.fetch_add(1);
	c->h_chr_end = parent->h_chr_end; c->h_chr_fwd = parent->h_chr_fwd; c->h_chr_of_end = parent->h_chr_of_end; c->h_chr_len = parent->h_chr_len;
	c->prm = parent->prm;
	*out = c;
	return GSA_OK;
}

// A context on ANOTHER GPU (or the same one) with a device index of its own that is COPIED from `parent`'s, device to device, instead of uploaded
// and rebuilt: the Occ blocks, both suffix arrays, the k-mer tables, the presence table and both text forms are plain position-independent arrays, so a
// second GPU needs none of gsa_create's work -- the 10.7 GB over PCIe, the regrouping, the dense-SA walk, the table scans (1.3 - 2.7 s for the
// human index) -- only the bytes, over xGMI.  The host-side state being replicated is bwt_index.cpp:147-264's (RefIdx, RefSequence, ChrLocMap).
int gsa_clone_to_device(gsa_ctx *parent, int device, gsa_ctx **out)
{
	if (!parent || !out) return GSA_ERR_ARG;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { (void)hipGetLastError(); return gsa_fail(nullptr, GSA_ERR_ARG, "gsa_clone_to_device: bad device ordinal"); }
	gsa_ctx *own = parent->index_owner ? parent->index_owner : parent;
	if (hipSetDevice(parent->device) != hipSuccess) return gsa_fail(nullptr, GSA_ERR_HIP, "hipSetDevice");
	if (hipStreamSynchronize(parent->stream) != hipSuccess || hipStreamSynchronize(own->stream) != hipSuccess) return gsa_fail(nullptr, GSA_ERR_HIP, "gsa_clone_to_device: the parent's stream");      // (tables a gsa_set_params has just queued)
	if (hipSetDevice(device) != hipSuccess) return gsa_fail(nullptr, GSA_ERR_HIP, "hipSetDevice");
	(void)hipSetDeviceFlags(hipDeviceScheduleSpin); (void)hipGetLastError();
	gsa_ctx *c = new gsa_ctx();
	c->device = device; c->force_wide = parent->force_wide; c->prio_mode = parent->prio_mode; c->seed_budget = parent->seed_budget; c->opt = parent->opt;
	memset(c->kernel_ms, 0, sizeof(c->kernel_ms)); memset(c->counters, 0, sizeof(c->counters));
	if (int rc = ctx_private_init(c)) { g_create_error = c->err; gsa_destroy(c); return rc; }
	// every index table of the owner (and the tables a clone built for parameters of its own: presence bitmap, short k-mer table)
	struct Pair { const DevBuf *src; DevBuf *dst; };
	std::vector<Pair> tab = { { &own->d_bwt, &c->d_bwt }, { &own->d_occ_base, &c->d_occ_base }, { &own->d_sa, &c->d_sa }, { &own->d_ref, &c->d_ref }, { &own->d_chr_end, &c->d_chr_end },
		{ &own->d_chr_of_end, &c->d_chr_of_end }, { &own->d_sa_dense, &c->d_sa_dense }, { &own->d_kmer, &c->d_kmer }, { &own->d_ref2, &c->d_ref2 } };
	for (gsa_ctx *src = parent; src; src = src->lender) {      // (a clone reads these two through the context it was cloned from, unless it built its own)
		if (!c->d_pres.cap && src->d_pres.p && (const void *)parent->di.pres == src->d_pres.p) { tab.push_back({ &src->d_pres, &c->d_pres }); c->d_pres.cap = 1; }
		if (!c->d_kmer_lo.cap && src->d_kmer_lo.p && (const void *)parent->di.kmer_lo == src->d_kmer_lo.p) { tab.push_back({ &src->d_kmer_lo, &c->d_kmer_lo }); c->d_kmer_lo.cap = 1; }
	}
	c->d_pres.cap = c->d_kmer_lo.cap = 0;
	auto fail = [&](int code, const std::string &m) { gsa_fail(nullptr, code, m); gsa_destroy(c); return code; };
	for (const Pair &t : tab) {
		if (!t.src->p) continue;
		const size_t bytes = t.src->len ? t.src->len : t.src->cap;
		if (bytes == 0) return fail(GSA_ERR_STATE, "gsa_clone_to_device: an index table of unknown size");
		if (hipMalloc(&t.dst->p, bytes + 256) != hipSuccess) { (void)hipGetLastError(); return fail(GSA_ERR_NOMEM, "gsa_clone_to_device: hipMalloc of an index table"); }
		t.dst->cap = bytes + 256; t.dst->len = bytes;
		const hipError_t e = (device == own->device) ? hipMemcpyAsync(t.dst->p, t.src->p, bytes, hipMemcpyDeviceToDevice, c->stream)
		                                             : hipMemcpyPeerAsync(t.dst->p, device, t.src->p, own->device, bytes, c->stream);
		if (e != hipSuccess) return fail(GSA_ERR_HIP, std::string("gsa_clone_to_device: device-to-device copy: ") + hipGetErrorString(e));
	}
	// `di` with every pointer moved to the copy it points into
	c->di = parent->di;
	auto move_ptr = [&](const void *p) -> const void * {
		if (!p) return nullptr;
		for (const Pair &t : tab) if (t.src->p && (const char *)p >= (const char *)t.src->p && (const char *)p < (const char *)t.src->p + (t.src->len ? t.src->len : t.src->cap)) return (const char *)t.dst->p + ((const char *)p - (const char *)t.src->p);
		return (const void *)~(uintptr_t)0;
	};
	bool lost = false;
#define MOVE(field, T) do { const void *q_ = move_ptr((const void *)c->di.field); if (q_ == (const void *)~(uintptr_t)0) lost = true; c->di.field = (T)q_; } while (0)
	MOVE(bwt, const uint4 *); MOVE(occ_base, const u64 *); MOVE(sa, const u64 *); MOVE(sa32, const u32 *); MOVE(sa64, const u64 *); MOVE(ref, const uint8_t *); MOVE(ref2, const u32 *);
	MOVE(chr_end, const i64 *); MOVE(chr_of_end, const i32 *); MOVE(kmer, const u64 *); MOVE(kmer_lo, const u64 *); MOVE(pres, const u32 *);
#undef MOVE
	if (lost) return fail(GSA_ERR_STATE, "gsa_clone_to_device: an index pointer outside the tables of its owner");
	c->G = parent->G;
	c->h_chr_end = parent->h_chr_end; c->h_chr_fwd = parent->h_chr_fwd; c->h_chr_of_end = parent->h_chr_of_end; c->h_chr_len = parent->h_chr_len;
	c->prm = parent->prm;
	if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(GSA_ERR_HIP, "gsa_clone_to_device: waiting for the copies");
	*out = c;
	return GSA_OK;
}

// Host-thread placement.  A contig of a few Mb is some sixty short GPU operations with five host look-ins: a host thread on
// the socket the GPU does not hang off pays the inter-socket hop on every doorbell, pinned-memory poll and count read-back
// (measured on a two-socket MI355X host with the whole process placed by taskset, 5 Mb contigs, three contexts: 0.86 ms per
// contig from the far socket, 0.65 - 0.75 from the near one.  NOT applied by default anywhere: 250 Mb contigs ran 5 % slower with
// the driving threads bound -- 14.4 against 13.7 ms per step on the same box -- so placement is the integrator's decision).  The CPUs local to the device come from sysfs (local_cpulist of its PCI function).
static bool device_local_cpus(int device, cpu_set_t *set)
{
	char bus[64] = { 0 };
	if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, device) != hipSuccess) { (void)hipGetLastError(); return false; }
	for (char *p = bus; *p; p++) if (*p >= 'A' && *p <= 'Z') *p = (char)(*p - 'A' + 'a');
	char path[160]; snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bus);
	FILE *f = fopen(path, "r"); if (!f) return false;
	char line[4096]; const bool got = fgets(line, sizeof(line), f) != nullptr; fclose(f);
	if (!got) return false;
	CPU_ZERO(set); int n = 0;
	for (char *p = line; *p;) {
		while (*p == ',' || *p == ' ' || *p == '\n') p++;
		if (!*p) break;
		char *e; long a = strtol(p, &e, 10), b = a; if (e == p) break;
		if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); if (e == p) break; }
		for (long k = a; k <= b && k < CPU_SETSIZE; k++) { CPU_SET((int)k, set); n++; }
		p = e;
	}
	return n > 0;
}
int gsa_reserve_index(int device, uint64_t seq_len, uint32_t flags)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || seq_len == 0) { (void)hipGetLastError(); return GSA_ERR_ARG; }
	if (hipSetDevice(device) != hipSuccess) return GSA_ERR_HIP;
	const bool wide = (flags & GSA_CREATE_WIDE) != 0 || seq_len >= 0xFFFFFFF0ull;
	size_t want[2];
	want[0] = ((size_t)seq_len + 1 + 32) * (wide ? 8 : 4) + 256;                 // dense SA (build_dense_sa: rows + 32 entries, exact)
	{	// k-mer table: build_dense_sa's choice of k (text length, a quarter of the free memory, GSA_CREATE_KMER_K)
		int k = 0; while ((1ull << (2 * k)) < seq_len) k++;
		k += 2; if (k > 15) k = 15;
		size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); fr = 8ull << 30; }
		const size_t esz = wide ? 32 : 16;
		while (k > 2 && ((size_t)esz << (2 * k)) > fr / 4) k--;
		const int kk = (int)((flags >> 8) & 15u); if (kk >= 2 && kk <= 15 && ((size_t)esz << (2 * kk)) <= fr / 2) k = kk;
		want[1] = k >= 2 ? ((size_t)esz << (2 * k)) : 0;
	}
	gsa_release_reserved(device);
	for (int k = 0; k < 2; k++) {
		if (!want[k]) continue;
		void *p = nullptr;
		if (hipMalloc(&p, want[k]) != hipSuccess) { (void)hipGetLastError(); continue; }      // (no reservation: gsa_create allocates as ever)
		std::lock_guard<std::mutex> g(g_res_mu);
		g_res[device & 63][k].p = p; g_res[device & 63][k].bytes = want[k];
	}
	return GSA_OK;
}
void gsa_release_reserved(int device)
{
	void *f[2] = { nullptr, nullptr };
	{ std::lock_guard<std::mutex> g(g_res_mu); for (int k = 0; k < 2; k++) { f[k] = g_res[device & 63][k].p; g_res[device & 63][k].p = nullptr; g_res[device & 63][k].bytes = 0; } }
	if (f[0] || f[1]) { (void)hipSetDevice(device); for (int k = 0; k < 2; k++) if (f[k]) hipFree(f[k]); }
}

int gsa_bind_host_thread(int device)
{
	cpu_set_t set;
	if (!device_local_cpus(device, &set)) return GSA_OK;      // (no topology information: leave the thread where it is)
	(void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
	return GSA_OK;
}

// Pinned host memory for query contigs: a FASTA loader that reads into such a buffer makes the upload of
// gsa_align_contig one asynchronous DMA transfer (pageable memory is staged through the runtime's bounce buffers).
void *gsa_host_alloc(size_t bytes)
{
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 1) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
	return p;
}
void gsa_host_free(void *p) { if (p) (void)hipHostFree(p); }
int gsa_host_register(void *p, size_t bytes)
{
	if (!p || !bytes) return GSA_ERR_ARG;
	if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return GSA_ERR_HIP; }
	return GSA_OK;
}
int gsa_host_unregister(void *p)
{
	if (!p) return GSA_ERR_ARG;
	if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return GSA_ERR_HIP; }
	return GSA_OK;
}

int gsa_set_option(gsa_ctx *c, const char *name, int64_t value)
{
	if (!c || !name) return GSA_ERR_ARG;
	const std::string k(name);
	auto in = [&](int64_t lo, int64_t hi) { return value >= lo && value <= hi; };
	const int64_t BIG = (int64_t)1 << 40;
	if (k == "split_min") { if (!in(0, BIG)) return gsa_fail(c, GSA_ERR_ARG, "split_min: 0 .. 2^40 bases"); c->opt.split_min = value; }
	else if (k == "bundle_contig") { if (!in(0, BIG)) return gsa_fail(c, GSA_ERR_ARG, "bundle_contig: 0 (no bundles) .. 2^40 bases"); c->opt.bundle_contig = value; }
	else if (k == "bundle_cap") { if (!in(1, BIG)) return gsa_fail(c, GSA_ERR_ARG, "bundle_cap: 1 .. 2^40 bases"); c->opt.bundle_cap = value; }
	else if (k == "seed_budget") { if (!in(1, 0xffffffffll)) return gsa_fail(c, GSA_ERR_ARG, "seed_budget: 1 .. 2^32 - 1 wave-iterations"); c->seed_budget = (u32)value; }
	else if (k == "dp_lane") { if (!in(0, 1 << 20)) return gsa_fail(c, GSA_ERR_ARG, "dp_lane: 0 (round 2's tiny / small split) .. 2^20 cells"); c->opt.dp_lane = (int)value; }
	else if (k == "seed_mode") { if (value < 0 || value > 2) return gsa_fail(c, GSA_ERR_ARG, "seed_mode: 0 sweep, 1 speculative, 2 search"); c->opt.seed_mode = (int)value; }
	else if (k == "pd_bitmap") c->opt.pd_bitmap = value != 0;
	else if (k == "dp_side") c->opt.dp_side = value != 0;
	else if (k == "dp_small_side") c->opt.dp_small_side = value != 0;
	else if (k == "pd_two_level_min") { if (!in(0, BIG)) return gsa_fail(c, GSA_ERR_ARG, "pd_two_level_min: >= 0 blocks"); c->opt.pd_two_level_min = value; }
	else if (k == "dp_occupancy") { if (!in(0, 16)) return gsa_fail(c, GSA_ERR_ARG, "dp_occupancy: 0 (off) .. 16 workgroups per CU"); c->opt.dp_occupancy = (int)value; }
	else if (k == "pd_bytes") { if (!in(0, 2)) return gsa_fail(c, GSA_ERR_ARG, "pd_bytes: 0 never, 1 by the hit count, 2 always"); c->opt.pd_bytes = (int)value; }
	else if (k == "pres_from_kmer") c->opt.pres_from_kmer = value != 0;      // (takes effect at the next gsa_set_params that rebuilds the table)
	else if (k == "walk_chain_min") { if (!in(0, BIG)) return gsa_fail(c, GSA_ERR_ARG, "walk_chain_min: >= 0 seeds"); c->opt.walk_chain_min = value; }
	else if (k == "sweep_shape") { if (value < -1 || value > 1) return gsa_fail(c, GSA_ERR_ARG, "sweep_shape: -1, 0 or 1"); c->opt.sweep_shape = (int)value; }
	else if (k == "dp_safe") c->dp_safe = value != 0;                    // (test hook)
	else if (k == "dp_fake_timeout") { if (!in(0, 1 << 20)) return gsa_fail(c, GSA_ERR_ARG, "dp_fake_timeout: >= 0"); c->dp_fake_timeout = (int)value; }    // (test hook)
	else return gsa_fail(c, GSA_ERR_ARG, "gsa_set_option: unknown option " + k);
	return GSA_OK;
}

int gsa_get_wall_sums(gsa_ctx *c, double ms[10], int64_t *n)
{
	if (!c || !ms || !n) return GSA_ERR_ARG;
	memcpy(ms, c->wall_ms, sizeof(c->wall_ms)); *n = c->wall_n;
	if (c->up && c->own_up) { std::lock_guard<std::mutex> g(c->up->mu); ms[9] = c->up->jobs ? c->up->copy_ms / (double)c->up->jobs * (double)c->wall_n : 0.0; }      // (x n: the caller divides by n)
	return GSA_OK;
}

int gsa_get_alloc_stats(gsa_ctx *c, double *ms, int64_t *n, int64_t *bytes)
{
	if (!c) return GSA_ERR_ARG;
	if (ms) *ms = c->alloc_ms; if (n) *n = c->alloc_n; if (bytes) *bytes = c->alloc_bytes;
	return GSA_OK;
}
// The device buffers of a context, largest first, to stderr: name, bytes held, bytes last asked for (diagnosis: what a context's share of HBM is made of).
int gsa_debug_buffers(gsa_ctx *c, int top)
{
	if (!c) return GSA_ERR_ARG;
	struct E { const char *name; size_t cap, len; };
	std::vector<E> v;
#define X(n) if (c->n.p) v.push_back({ #n, c->n.cap, c->n.len });
	GSA_DEVBUFS(X)
#undef X
	std::sort(v.begin(), v.end(), [](const E &a, const E &b) { return a.cap > b.cap; });
	size_t tot = 0; for (const E &e : v) tot += e.cap;
	fprintf(stderr, "[gsa_debug_buffers] %zu device buffers, %.2f GB held\n", v.size(), (double)tot / 1e9);
	for (size_t k = 0; k < v.size() && (int)k < top; k++) fprintf(stderr, "  %-16s %10.1f MB held  %10.1f MB asked\n", v[k].name, (double)v[k].cap / 1e6, (double)v[k].len / 1e6);
	return GSA_OK;
}
int gsa_set_profiling(gsa_ctx *c, int enable) { if (!c) return GSA_ERR_ARG; c->acc_seed_ms = 0.0; memset(c->wall_ms, 0, sizeof(c->wall_ms)); c->wall_n = 0; if (c->up && c->own_up) { std::lock_guard<std::mutex> g(c->up->mu); c->up->copy_ms = c->up->wait_ms = c->up->bytes = 0; c->up->jobs = 0; } c->profiling = (enable & 1) != 0; c->count_blocks = (enable & 2) != 0; c->prof_seed = (enable & 4) != 0; return GSA_OK; }

static int query_geometry(gsa_ctx *c, int32_t qlen)
{
	c->qlen = qlen;
	c->bnd.n = 0; c->bnd.lmax = qlen; c->bnd.pds = 0; c->bnd.off = (const i32 *)c->d_zero.p; c->bnd.chunk_contig = (const uint16_t *)c->d_zero.p;      // (contig 0 at offset 0: readable without asking bnd.n)
	c->qbits = ceil_log2_u64((u64)qlen + 1); if (c->qbits < 1) c->qbits = 1;
	c->pd_span = 2 * c->G + (i64)qlen + 2;
	c->pdbits = ceil_log2_u64((u64)c->pd_span);
	if (c->qbits + c->pdbits > 64) return gsa_fail(c, GSA_ERR_LIMIT, "contig too long for the 64-bit seed key");
	return GSA_OK;
}

#define GSA_BUNDLE_MAX_CONTIGS 4096
// ---- query slots (QuerySlot, gsa_ctx.h) ----
// a slot's buffer grows: nothing may still read or write it (the stages of the contig it held are over -- the API is synchronous --
// but an upload may be on its way)
static void *slot_ensure_bytes(gsa_ctx *c, DevBuf &b, size_t bytes, bool pinned)
{
	if (bytes && bytes <= b.cap) return b.p;
	return pinned ? (void *)pin_ensure<uint8_t>(c, b, bytes) : (void *)dev_ensure<uint8_t>(c, b, bytes);
}

// The concatenation of a bundle: every contig starts on a chunk edge, as it does alone.
static int slot_layout(gsa_ctx *c, QuerySlot &s, const char *const *query, const int32_t *qlen, int32_t n)
{
	s.b_off.assign((size_t)n + 1, 0); s.b_qlen.assign(qlen, qlen + n); s.src.assign(query, query + n);
	i64 tot = 0; i32 lmax = 0;
	for (int k = 0; k < n; k++) {
		if (qlen[k] < 0 || (qlen[k] > 0 && !query[k])) return GSA_ERR_ARG;
		s.b_off[(size_t)k] = (i32)tot;
		tot += ((i64)qlen[k] + GSA_CHUNK - 1) / GSA_CHUNK * GSA_CHUNK;
		if (tot >= (1ll << 31) - 2 * GSA_CHUNK) return gsa_fail(c, GSA_ERR_LIMIT, "bundle longer than 2^31 bases");
		if (qlen[k] > lmax) lmax = qlen[k];
	}
	s.b_off[(size_t)n] = (i32)tot; s.tot = tot; s.lmax = lmax; s.n = n;
	return GSA_OK;
}

// the tail of every contig of a bundle up to its chunk edge reads 'N' (a match stops there exactly as it stops at the end of a sequence)
__global__ void __launch_bounds__(256) k_bundle_pad(const i32 *__restrict__ off, const i32 *__restrict__ len, uint8_t *dst)
{
	const i32 k = blockIdx.x; const i64 b = (i64)off[k] + len[k], e = off[k + 1];
	for (i64 t = b + threadIdx.x; t < e; t += 256) dst[t] = (uint8_t)'N';
}

// Hands the Uploader the copies that fill slot `s` from host memory: one contig (n = 0, query[0] / qlen[0]) or the concatenation of a
// bundle + its tables (the 'N' tails are written by slot_finish on the main stream: nothing but copies on the upload stream).
static int slot_upload(gsa_ctx *c, QuerySlot &s, const char *const *query, const int32_t *qlen, int32_t n)
{
	slot_wait(s);      // (a cancelled upload into this slot may still be on its way)
	s.up_err.store(0);      // (... and whatever became of it no longer matters)
	Uploader::Job job; job.busy = &s.busy; job.err = &s.up_err;
	if (n == 0) {
		s.n = 0; s.tot = qlen[0]; s.lmax = qlen[0]; s.src.assign(1, query[0]); s.b_qlen.assign(1, qlen[0]); s.b_off.clear();
		if (!slot_ensure_bytes(c, s.d_query, (size_t)qlen[0] + 64, false)) return GSA_ERR_NOMEM;
		size_t nb = (size_t)qlen[0];
#ifdef GSA_EXPERIMENTS
		{ static const long long cap = [] { const char *e = getenv("GSA_X_UPBYTES"); return e ? atoll(e) : -1ll; }(); if (cap >= 0 && (size_t)cap < nb) nb = (size_t)cap; }      // (timing experiment: the API pattern without the bytes)
#endif
		// (a buffer from gsa_host_alloc is pinned: one DMA transfer; pageable memory is staged by the runtime)
		if (nb > 0) { job.pieces.push_back({ s.d_query.p, query[0], nb }); c->up->push(std::move(job)); }
		return GSA_OK;
	}
	if (int rc = slot_layout(c, s, query, qlen, n)) return rc;
	const i64 n_chunks = s.tot / GSA_CHUNK;
	// tables: off[n + 1] (i32) | chunk_contig[n_chunks] (u16) | len[n] (i32)
	s.o_cc = ((size_t)n + 1) * 4; s.o_src = (s.o_cc + (size_t)n_chunks * 2 + 15) & ~(size_t)15;
	const size_t t_bytes = s.o_src + (size_t)n * 4;
	if (!slot_ensure_bytes(c, s.p_bndtab, t_bytes + 16, true) || !slot_ensure_bytes(c, s.d_bndtab, t_bytes + 16, false) || !slot_ensure_bytes(c, s.d_query, (size_t)s.tot + 64, false)) return GSA_ERR_NOMEM;
	uint8_t *ht = s.p_bndtab.as<uint8_t>();
	memcpy(ht, s.b_off.data(), s.o_cc);
	{ uint16_t *cc = (uint16_t *)(ht + s.o_cc); for (int k = 0; k < n; k++) for (i64 ch = s.b_off[(size_t)k] / GSA_CHUNK; ch < s.b_off[(size_t)k + 1] / GSA_CHUNK; ch++) cc[ch] = (uint16_t)k; }
	memcpy(ht + s.o_src, qlen, (size_t)n * 4);
	job.pieces.push_back({ s.d_bndtab.p, ht, t_bytes });
	for (int k = 0; k < n; k++) if (qlen[k] > 0) job.pieces.push_back({ s.d_query.as<uint8_t>() + s.b_off[(size_t)k], query[k], (size_t)qlen[k] });
	c->up->push(std::move(job));
	return GSA_OK;
}

// The host waits for the slot's upload (long over when the contig was prefetched a contig ago), then the main stream pads a bundle's contigs.
static int slot_finish(gsa_ctx *c, QuerySlot &s)
{
	slot_demand(c, s);
	if (const int e = s.up_err.exchange(0)) return gsa_fail(c, GSA_ERR_HIP, std::string("query upload (hipMemcpyAsync H2D): ") + hipGetErrorString((hipError_t)e));
	if (s.n > 0 && s.tot > 0) {
		const uint8_t *dt = s.d_bndtab.as<uint8_t>();
		hipLaunchKernelGGL(k_bundle_pad, dim3((unsigned)s.n), dim3(256), 0, c->stream, (const i32 *)dt, (const i32 *)(dt + s.o_src), s.d_query.as<uint8_t>());
		GSA_CHECK(c, hipGetLastError());
	}
	return GSA_OK;
}

static bool slot_holds(const QuerySlot &s, const char *const *query, const int32_t *qlen, int32_t n)
{
	if (!s.pending || s.n != n) return false;
	const int32_t m = n ? n : 1;
	if ((int32_t)s.src.size() != m || (int32_t)s.b_qlen.size() != m) return false;
	for (int32_t k = 0; k < m; k++) if (s.src[(size_t)k] != query[k] || s.b_qlen[(size_t)k] != qlen[k]) return false;
	return true;
}

// The slot the stages will read: the one a prefetch filled with exactly these buffers (the main stream waits for its upload), else a
// free one, filled on the main stream.
static int slot_acquire(gsa_ctx *c, const char *const *query, const int32_t *qlen, int32_t n, int *which)
{
	for (int i = 0; i < 2; i++) if (slot_holds(c->qs[i], query, qlen, n)) {
		c->qs[i].pending = false; *which = i;
		return slot_finish(c, c->qs[i]);
	}
	int i = (c->q_cur == 1 && !c->qs[1].pending) ? 1 : 0;
	if (c->qs[i].pending) i ^= 1;
	if (c->qs[i].pending) return gsa_fail(c, GSA_ERR_STATE, "both query slots hold prefetched contigs: align them (or gsa_cancel_prefetch) first");
	*which = i;
	if (int rc = slot_upload(c, c->qs[i], query, qlen, n)) return rc;
	return slot_finish(c, c->qs[i]);
}

static int prefetch(gsa_ctx *c, const char *const *query, const int32_t *qlen, int32_t n)
{
	GSA_CHECK(c, hipSetDevice(c->device));
	// (the same buffer may wait twice -- a caller that aligns one buffer again and again -- so a waiting copy of it is no reason to skip this one)
	// not the slot of the contig that was aligned last while the other one is free (gsa_rewind stays possible); a pending slot never
	int i = c->q_cur == 0 ? 1 : 0;
	if (c->qs[i].pending) i ^= 1;
	if (c->qs[i].pending) return gsa_fail(c, GSA_ERR_STATE, "gsa_prefetch: two contigs are waiting already");
	// the only free slot holds the contig whose stages are under way (stage views: gsa_set_query + gsa_run_to(k < 8)): the later stages still
	// read its bases there -- overwriting, or growing and so freeing, that buffer would hand them another contig or freed memory
	if (i == c->q_cur && c->stage > 0 && c->stage < 8) return gsa_fail(c, GSA_ERR_STATE, "gsa_prefetch: the free query slot holds the contig whose stages are still running (finish it with gsa_run_to(8) or gsa_set_query first)");
	if (i == c->q_cur) c->q_cur = -2;
	int rc = slot_upload(c, c->qs[i], query, qlen, n);
	if (rc == GSA_OK) c->qs[i].pending = true;
	return rc;
}

int gsa_prefetch_contig(gsa_ctx *c, const char *query, int32_t qlen)
{
	if (!c || !query || qlen < 0) return GSA_ERR_ARG;
	return prefetch(c, &query, &qlen, 0);
}

int gsa_prefetch_bundle(gsa_ctx *c, const char *const *query, const int32_t *qlen, int32_t n)
{
	if (!c || !query || !qlen) return GSA_ERR_ARG;
	if (n < 1 || n > GSA_BUNDLE_MAX_CONTIGS) return gsa_fail(c, GSA_ERR_ARG, "gsa_prefetch_bundle: 1 .. 4096 contigs");
	return prefetch(c, query, qlen, n);
}

int gsa_cancel_prefetch(gsa_ctx *c)
{
	if (!c) return GSA_ERR_ARG;
	if (!c->qs[0].pending && !c->qs[1].pending) return GSA_OK;
	c->qs[0].pending = c->qs[1].pending = false;      // (the copies run to their end; slot_upload waits for them before it reuses the slot)
	return GSA_OK;
}

int gsa_set_query(gsa_ctx *c, const char *query, int32_t qlen)
{
	if (!c || !query || qlen < 0) return GSA_ERR_ARG;
	GSA_CHECK(c, hipSetDevice(c->device));
	if (int rc = reset_run_state(c)) return rc;
	int w = 0;
	const auto t0 = std::chrono::steady_clock::now();
	if (int rc = slot_acquire(c, &query, &qlen, 0, &w)) return rc;
	c->wall_ms[0] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	c->q_cur = w; c->q_dev = c->qs[w].d_query.as<uint8_t>();
	return query_geometry(c, qlen);
}

// The contig is already in device memory (a loader that decodes FASTA on the GPU, or a contig kept resident between runs):
// used in place, no copy.
int gsa_set_query_device(gsa_ctx *c, const char *d_query, int32_t qlen)
{
	if (!c || !d_query || qlen < 0) return GSA_ERR_ARG;
	if (((uintptr_t)d_query & 15) != 0) return gsa_fail(c, GSA_ERR_ARG, "gsa_set_query_device: the buffer must be 16-byte aligned");
	GSA_CHECK(c, hipSetDevice(c->device));
	hipPointerAttribute_t at;
	if (hipPointerGetAttributes(&at, d_query) != hipSuccess || at.type != hipMemoryTypeDevice || at.device != c->device) { (void)hipGetLastError(); return gsa_fail(c, GSA_ERR_ARG, "gsa_set_query_device: not a device buffer of this context's GPU"); }
	if (int rc = reset_run_state(c)) return rc;
	c->q_dev = (const uint8_t *)d_query; c->q_cur = -1;
	return query_geometry(c, qlen);
}

void *gsa_device_alloc(int device, size_t bytes)
{
	void *p = nullptr;
	if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, (bytes ? bytes : 1) + 64) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
	return p;
}
void gsa_device_free(int device, void *p) { if (p && hipSetDevice(device) == hipSuccess) (void)hipFree(p); }
int gsa_device_upload(int device, void *dst, const void *src, size_t bytes)
{
	if (!dst || (!src && bytes)) return GSA_ERR_ARG;
	if (hipSetDevice(device) != hipSuccess || hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return GSA_ERR_HIP; }
	return GSA_OK;
}

// Back to the end of stage 1 of a contig whose hits were imported (gsa_finish_contig's retry): everything later is dropped.
static int rewind_to_hits(gsa_ctx *c)
{
	const i64 n_seeds = c->n_seeds; const i32 n_groups = c->n_groups;
	if (int rc = reset_run_state(c)) return rc;
	GSA_CHECK(c, hipStreamSynchronize(c->stream));
	c->n_seeds = n_seeds; c->n_groups = n_groups; c->counters[2] = c->counters[3] = (u64)n_seeds;
	c->stage = 1;
	return stage1_restore_pdbm(c);
}

// Back to stage 0 with the same contig: the query stays where gsa_set_query put it (device and host copy).
int gsa_rewind(gsa_ctx *c)
{
	if (!c) return GSA_ERR_ARG;
	if (c->qlen <= 0) return gsa_fail(c, GSA_ERR_STATE, "gsa_set_query first");
	if (c->q_cur == -2) return gsa_fail(c, GSA_ERR_STATE, "gsa_rewind: the contig's device copy was overwritten by a prefetch");
	GSA_CHECK(c, hipSetDevice(c->device));
	return reset_run_state(c);
}

int gsa_run_to(gsa_ctx *c, int stage)
{
	if (!c || stage < 0 || stage > 8) return GSA_ERR_ARG;
	if (c->bnd.n && stage < 8 && !c->bundle_call) return gsa_fail(c, GSA_ERR_STATE, "gsa_run_to: the context holds a bundle (gsa_align_bundle runs it); gsa_set_query first");
	// (between gsa_seed_chunks and gsa_finish_contig the context holds the hits of a chunk range only: stage 1 here would seed
	//  that range again and call the result a contig)
	if (c->split) return gsa_fail(c, GSA_ERR_STATE, "gsa_run_to: a split contig is finished with gsa_finish_contig");
	GSA_CHECK(c, hipSetDevice(c->device));
	int rc = GSA_OK;
	auto t_prev = std::chrono::steady_clock::now();
	while (c->stage < stage && rc == GSA_OK) {
		const int next = c->stage + 1;
		switch (next) {
		case 1: rc = stage1_seed(c); break;
		case 2: rc = stage2_chain(c); break;
		case 3: rc = stage345_refine(c); if (rc == GSA_OK && c->bnd.n) bundle_split_lists(c); break;      // device part of S3..S5, host list = S3 state
		case 4: case 5: case 6: rc = host_stage4_5_6(c, next); break;
		case 7: if (c->bnd.n) bundle_join_lists(c); rc = stage7_fill(c); break;
		case 8: rc = stage78_extend(c); if (rc == GSA_OK) rc = host_stage8_finish(c); break;
		}
		if (rc == GSA_OK) { c->stage = next; c->frags_stage = (next == 8) ? 8 : 0; }
		{ const auto t_now = std::chrono::steady_clock::now(); c->wall_ms[next] += std::chrono::duration<double, std::milli>(t_now - t_prev).count(); t_prev = t_now; }
	}
	if (stage == 8) c->wall_n++;
	if (const unsigned long long ov = gsa_take_grid_overflow()) rc = gsa_fail(c, GSA_ERR_LIMIT, "a kernel launch of " + std::to_string(ov) + " work-items (>= 2^32) was needed: contig too large for this build");
	// stages 1-2 leave work in flight (no count read-backs); the call returns with the stream idle
	if (rc == GSA_OK) { GSA_CHECK(c, hipStreamSynchronize(c->stream)); collect_events(c); }
	if (c->early_in_flight && (rc != GSA_OK || !c->early_consumed)) GSA_CHECK(c, hipStreamSynchronize(c->stream_aux[0]));      // a stage view (or an error) must not leave the early DP launch running
	return rc;
}

static int align_uploaded(gsa_ctx *c, gsa_result *out);
int gsa_align_contig(gsa_ctx *c, const char *query, int32_t qlen, gsa_result *out)
{
	if (!c || !out) return GSA_ERR_ARG;
	int rc = gsa_set_query(c, query, qlen); if (rc) return rc;
	return align_uploaded(c, out);
}
int gsa_align_contig_device(gsa_ctx *c, const char *d_query, int32_t qlen, gsa_result *out)
{
	if (!c || !out) return GSA_ERR_ARG;
	int rc = gsa_set_query_device(c, d_query, qlen); if (rc) return rc;
	return align_uploaded(c, out);
}
static int align_uploaded(gsa_ctx *c, gsa_result *out)
{
	int rc;
	c->dp_timeout = false;
	rc = gsa_run_to(c, 8);
	if (rc == GSA_ERR_STATE && c->dp_timeout && !c->dp_safe) {
		// a stripe waited longer than its bound for its predecessor (never observed with ticket-ordered stripes; kept as a
		// safety net): the same contig again with one DP job per launch -- all stripes of a job are resident then
		c->dp_timeout = false; c->dp_safe = true;
		rc = gsa_rewind(c);
		if (rc == GSA_OK) rc = gsa_run_to(c, 8);
		c->dp_safe = false;
	}
	if (rc) return rc;
	return gsa_get_blocks(c, out);
}

// ---- several contigs in one pass (Bundle, gsa_internal.h) ----
// The concatenation, device-resident contigs: one workgroup per chunk copies its 10 000 bytes (or the tail of its contig and 'N's).
struct BundleSrc { const uint8_t *p; i32 len, _pad; };
__global__ void __launch_bounds__(256) k_bundle_gather(const i32 *__restrict__ off, const uint16_t *__restrict__ chunk_contig, const BundleSrc *__restrict__ src, uint8_t *dst)
{
	const i32 chunk = blockIdx.x, ci = chunk_contig[chunk];
	const i64 g0 = (i64)chunk * GSA_CHUNK; const i32 l0 = (i32)(g0 - off[ci]);
	const BundleSrc sc = src[ci];
	const i32 have = sc.len - l0 < GSA_CHUNK ? sc.len - l0 : GSA_CHUNK;
	for (i32 t = threadIdx.x; t < GSA_CHUNK; t += 256) dst[g0 + t] = t < have ? sc.p[l0 + t] : (uint8_t)'N';
}

static int set_query_bundle(gsa_ctx *c, const char *const *query, const int32_t *qlen, int32_t n, bool dev_q)
{
	if (n < 1 || n > GSA_BUNDLE_MAX_CONTIGS) return gsa_fail(c, GSA_ERR_ARG, "gsa_align_bundle: 1 .. 4096 contigs");
	GSA_CHECK(c, hipSetDevice(c->device));
	if (int rc = reset_run_state(c)) return rc;
	int w = 0;
	if (!dev_q) { if (int rc = slot_acquire(c, query, qlen, n, &w)) return rc; }
	else {
		// device-resident contigs: gathered into a free slot by one kernel (a workgroup per chunk), sources in the slot's table
		w = (c->q_cur == 1 && !c->qs[1].pending) ? 1 : 0;
		if (c->qs[w].pending) w ^= 1;
		if (c->qs[w].pending) return gsa_fail(c, GSA_ERR_STATE, "both query slots hold prefetched contigs");
		QuerySlot &s = c->qs[w];
		slot_wait(s);
		for (int k = 0; k < n; k++) if (qlen[k] > 0) {      // (what gsa_set_query_device checks: a host pointer here would be a GPU fault, not an error code)
			hipPointerAttribute_t at;
			if (!query[k] || hipPointerGetAttributes(&at, query[k]) != hipSuccess || at.type != hipMemoryTypeDevice || at.device != c->device) { (void)hipGetLastError(); return gsa_fail(c, GSA_ERR_ARG, "gsa_align_bundle (GSA_MANY_DEVICE): not a device buffer of this context's GPU"); }
		}
		if (int rc = slot_layout(c, s, query, qlen, n)) return rc;
		const i64 n_chunks = s.tot / GSA_CHUNK;
		s.o_cc = ((size_t)n + 1) * 4; s.o_src = (s.o_cc + (size_t)n_chunks * 2 + 15) & ~(size_t)15;
		const size_t t_bytes = s.o_src + (size_t)n * sizeof(BundleSrc);
		if (!slot_ensure_bytes(c, s.p_bndtab, t_bytes + 16, true) || !slot_ensure_bytes(c, s.d_bndtab, t_bytes + 16, false) || !slot_ensure_bytes(c, s.d_query, (size_t)s.tot + 64, false)) return GSA_ERR_NOMEM;
		uint8_t *ht = s.p_bndtab.as<uint8_t>();
		memcpy(ht, s.b_off.data(), s.o_cc);
		{ uint16_t *cc = (uint16_t *)(ht + s.o_cc); for (int k = 0; k < n; k++) for (i64 ch = s.b_off[(size_t)k] / GSA_CHUNK; ch < s.b_off[(size_t)k + 1] / GSA_CHUNK; ch++) cc[ch] = (uint16_t)k; }
		{ BundleSrc *bs = (BundleSrc *)(ht + s.o_src); for (int k = 0; k < n; k++) { bs[k].p = (const uint8_t *)query[k]; bs[k].len = qlen[k]; bs[k]._pad = 0; } }
		GSA_CHECK(c, hipMemcpyAsync(s.d_bndtab.p, ht, t_bytes, hipMemcpyHostToDevice, c->stream));
		const uint8_t *dt = s.d_bndtab.as<uint8_t>();
		if (s.tot > 0) {
			hipLaunchKernelGGL(k_bundle_gather, dim3((unsigned)n_chunks), dim3(256), 0, c->stream, (const i32 *)dt, (const uint16_t *)(dt + s.o_cc), (const BundleSrc *)(dt + s.o_src), s.d_query.as<uint8_t>());
			GSA_CHECK(c, hipGetLastError());
		}
	}
	const QuerySlot &s = c->qs[w];
	c->q_cur = w; c->b_off = s.b_off; c->b_qlen = s.b_qlen;
	const uint8_t *dt = s.d_bndtab.as<uint8_t>();
	c->q_dev = s.d_query.as<uint8_t>();
	c->qlen = (i32)s.tot;
	c->bnd.n = n; c->bnd.lmax = s.lmax; c->bnd.off = (const i32 *)dt; c->bnd.chunk_contig = (const uint16_t *)(dt + s.o_cc);
	c->bnd.pds = (2 * c->G + (i64)s.lmax + (i64)c->prm.MaxIndelSize + 64 + 31) & ~31ll;      // (a PosDiff of contig k: rPos - qLocal + lmax in (0, 2G + lmax])
	c->qbits = ceil_log2_u64((u64)s.tot + 1); if (c->qbits < 1) c->qbits = 1;
	c->pd_span = (i64)n * c->bnd.pds + 2;
	c->pdbits = ceil_log2_u64((u64)c->pd_span);
	if (c->qbits + c->pdbits > 64) return gsa_fail(c, GSA_ERR_LIMIT, "bundle too long for the 64-bit seed key");
	return GSA_OK;
}

// the result of contig k of the bundle this context just aligned (valid until the next call on the context)
static void bundle_result(gsa_ctx *c, int k, const std::vector<i64> &a0, size_t blk_at, gsa_result *out)
{
	out->n_blocks = c->b_nblk[(size_t)k]; out->blocks = c->h_blocks.data() + blk_at;
	out->n_frags = c->b_frag0[(size_t)k + 1] - c->b_frag0[(size_t)k]; out->recs = c->p_frags.as<gsa_rec>() + c->b_frag0[(size_t)k];
	out->n_aln = a0[(size_t)k + 1] - a0[(size_t)k]; out->aln1 = c->h_taln1 + a0[(size_t)k]; out->aln2 = c->h_taln2 + a0[(size_t)k];
}

int gsa_align_bundle(gsa_ctx *c, const char *const *query, const int32_t *qlen, int32_t n, uint32_t flags, gsa_result *out)
{
	if (!c || !query || !qlen || !out || (flags & ~(uint32_t)GSA_MANY_DEVICE)) return GSA_ERR_ARG;
	int rc = set_query_bundle(c, query, qlen, n, (flags & GSA_MANY_DEVICE) != 0); if (rc) return rc;
	c->dp_timeout = false;
	c->bundle_call = true;
	struct Clear { gsa_ctx *c; ~Clear() { c->bundle_call = false; } } clear_{ c };
	rc = gsa_run_to(c, 8);
	if (rc == GSA_ERR_STATE && c->dp_timeout && !c->dp_safe) {      // (the safety net of gsa_align_contig; the concatenation stays where it is)
		c->dp_timeout = false; c->dp_safe = true;
		{ const Bundle keep = c->bnd; rc = reset_run_state(c); c->bnd = keep; }
		if (rc == GSA_OK) rc = gsa_run_to(c, 8);
		c->dp_safe = false;
	}
	if (rc) return rc;
	const size_t nfb = c->b_blk0.empty() ? 0 : (size_t)c->b_blk0[(size_t)n];
	if (nfb == 0 || c->n_frags == 0) {
		for (int k = 0; k < n; k++) { out[k].n_blocks = 0; out[k].n_frags = 0; out[k].n_aln = 0; out[k].blocks = nullptr; out[k].recs = nullptr; out[k].aln1 = out[k].aln2 = nullptr; }
		return GSA_OK;
	}
	// string-pool offset of every contig: k_bundle_rebase left it for the contigs that have records
	std::vector<i64> a0((size_t)n + 1, c->n_aln);
	for (int k = n - 1; k >= 0; k--) a0[(size_t)k] = c->b_blk0[(size_t)k] < c->b_blk0[(size_t)k + 1] ? c->p_ba0.as<i64>()[k] : a0[(size_t)k + 1];
	size_t at = 0;
	for (int k = 0; k < n; k++) { bundle_result(c, k, a0, at, &out[k]); at += (size_t)c->b_nblk[(size_t)k]; }
	return GSA_OK;
}

// ---- one contig seeded by several GPUs ----
int gsa_seed_chunks(gsa_ctx *c, const char *query, int32_t qlen, int32_t chunk_beg, int32_t chunk_end)
{
	if (!c || chunk_beg < 0 || chunk_end < chunk_beg) return GSA_ERR_ARG;
	int rc = gsa_set_query(c, query, qlen); if (rc) return rc;
	c->split = true; c->rng_beg = chunk_beg; c->rng_end = chunk_end;
	rc = stage1_seed(c);
	if (rc == GSA_OK) GSA_CHECK(c, hipStreamSynchronize(c->stream));
	collect_events(c);
	return rc;
}

int64_t gsa_hit_count(gsa_ctx *c) { return (c && c->split) ? c->n_seeds : 0; }

int gsa_export_hits(gsa_ctx *c, uint64_t *keys, uint32_t *vals)
{
	if (!c || !c->split) return c ? gsa_fail(c, GSA_ERR_STATE, "gsa_seed_chunks first") : GSA_ERR_ARG;
	if (c->n_seeds == 0) return GSA_OK;
	if (!keys || !vals) return GSA_ERR_ARG;
	GSA_CHECK(c, hipSetDevice(c->device));
	GSA_CHECK(c, hipMemcpyAsync(keys, c->d_key_a.p, (size_t)c->n_seeds * 8, hipMemcpyDefault, c->stream));
	GSA_CHECK(c, hipMemcpyAsync(vals, c->d_val_a.p, (size_t)c->n_seeds * 4, hipMemcpyDefault, c->stream));
	GSA_CHECK(c, hipStreamSynchronize(c->stream));
	return GSA_OK;
}

int gsa_import_hits(gsa_ctx *c, const uint64_t *keys, const uint32_t *vals, int64_t n)
{
	if (!c || n < 0 || (n > 0 && (!keys || !vals))) return GSA_ERR_ARG;
	if (!c->split || c->stage != 0) return gsa_fail(c, GSA_ERR_STATE, "gsa_seed_chunks first");
	GSA_CHECK(c, hipSetDevice(c->device));
	return stage1_import_hits(c, (const u64 *)keys, (const u32 *)vals, n);
}

int gsa_finish_contig(gsa_ctx *c, gsa_result *out)
{
	if (!c || !out) return GSA_ERR_ARG;
	if (!c->split || c->stage != 0) return gsa_fail(c, GSA_ERR_STATE, "gsa_seed_chunks first");
	GSA_CHECK(c, hipSetDevice(c->device));
	int rc = stage1_finish_split(c); if (rc) return rc;
	c->stage = 1; c->split = false;
	c->dp_timeout = false;
	rc = gsa_run_to(c, 8);
	if (rc == GSA_ERR_STATE && c->dp_timeout && !c->dp_safe) {
		// the same safety net as gsa_align_contig: stages 2-8 again with one DP job per launch, from the hits this context
		// still holds (d_key_a / d_val_a are read-only for stage 2; the PosDiff bitmap it consumed is set again from the keys)
		c->dp_timeout = false; c->dp_safe = true;
		rc = rewind_to_hits(c);
		if (rc == GSA_OK) rc = gsa_run_to(c, 8);
		c->dp_safe = false;
	}
	if (rc) return rc;
	return gsa_get_blocks(c, out);
}

// The hits of this context's chunk range where they lie (device memory; valid until the next call on this context): what an
// owner on the same node imports directly -- gsa_import_hits(owner, keys, vals, n) copies device to device, peer to peer when the
// two contexts sit on different GPUs.
int gsa_hit_buffers(gsa_ctx *c, const uint64_t **keys, const uint32_t **vals)
{
	if (!c || !keys || !vals) return GSA_ERR_ARG;
	if (!c->split) return gsa_fail(c, GSA_ERR_STATE, "gsa_seed_chunks first");
	*keys = c->d_key_a.as<uint64_t>(); *vals = c->d_val_a.as<uint32_t>();
	return GSA_OK;
}

// One contig on a group of contexts (the first one owns it): IdentifyLocalMEM hands 10 000-bp chunks to whichever thread is free
// (GSAlign.cpp:61-94); here every context of the group seeds a contiguous chunk range, the owner imports the others' hits
// device to device and runs the rest.
static int align_split(gsa_ctx *const *grp, int n_grp, const char *query, int32_t qlen, gsa_result *out)
{
	const int32_t n_chunks = (int32_t)(((int64_t)qlen + GSA_CHUNK - 1) / GSA_CHUNK);
	std::vector<int> rcs((size_t)n_grp, GSA_OK);
	std::vector<std::thread> th;
	auto range = [&](int k, int32_t &b, int32_t &e) { const int32_t base = n_chunks / n_grp, extra = n_chunks % n_grp; b = k * base + (k < extra ? k : extra); e = b + base + (k < extra ? 1 : 0); };
	for (int k = 1; k < n_grp; k++) th.emplace_back([&, k] { int32_t b, e; range(k, b, e); rcs[(size_t)k] = gsa_seed_chunks(grp[k], query, qlen, b, e); });
	{ int32_t b, e; range(0, b, e); rcs[0] = gsa_seed_chunks(grp[0], query, qlen, b, e); }
	for (std::thread &t : th) t.join();
	for (int k = 0; k < n_grp; k++) if (rcs[(size_t)k] != GSA_OK) { if (k) gsa_fail(grp[0], rcs[(size_t)k], std::string("helper context: ") + gsa_last_error(grp[k])); return rcs[(size_t)k]; }
	for (int k = 1; k < n_grp; k++) {
		const uint64_t *kp = nullptr; const uint32_t *vp = nullptr;
		int rc = gsa_hit_buffers(grp[k], &kp, &vp);
		if (rc == GSA_OK) rc = gsa_import_hits(grp[0], kp, vp, gsa_hit_count(grp[k]));
		if (rc != GSA_OK) return rc;
	}
	return gsa_finish_contig(grp[0], out);
}

int gsa_align_many(gsa_ctx *const *ctx, int32_t n_ctx, const char *const *query, const int32_t *qlen, int32_t n, uint32_t flags, gsa_result_fn on_result, void *user)
{
	if (flags & ~(uint32_t)(GSA_MANY_IN_ORDER | GSA_MANY_DEVICE | GSA_MANY_NO_SPLIT | GSA_MANY_NO_BUNDLE | GSA_MANY_NO_PREFETCH)) return GSA_ERR_ARG;
	if (!ctx || n_ctx <= 0 || n < 0 || (n > 0 && (!query || !qlen))) return GSA_ERR_ARG;
	for (int k = 0; k < n_ctx; k++) if (!ctx[k]) return GSA_ERR_ARG;
	const bool dev_q = (flags & GSA_MANY_DEVICE) != 0;
	if (dev_q) for (int k = 1; k < n_ctx; k++) if (ctx[k]->device != ctx[0]->device) return gsa_fail(ctx[0], GSA_ERR_ARG, "GSA_MANY_DEVICE: the contexts must share one GPU (the contigs live in its memory)");
	if (n == 0) return GSA_OK;
	// longest first: with dynamic hand-out this is the longest-processing-time-first rule
	std::vector<int32_t> order((size_t)n);
	for (int32_t i = 0; i < n; i++) order[(size_t)i] = i;
	if (!(flags & GSA_MANY_IN_ORDER)) std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return qlen[a] > qlen[b]; });
	std::atomic<int> err(GSA_OK);
	// Fewer contigs than contexts (BASELINE configs[3]: one chromosome, two GPUs): contexts would sit idle, so the contexts are
	// dealt out in GROUPS, one per contig, sized by contig length, and every group seeds its contig by chunk range (align_split).
	// Only contigs of at least GSA_SPLIT_MIN bases (default 20 Mb: below that the seed search is a fraction of a millisecond and
	// a second upload of the contig costs more than it saves) get more than one context.
	const int64_t split_min = ctx[0]->opt.split_min;
	int64_t longest = 0; for (int32_t i = 0; i < n; i++) if (qlen[i] > longest) longest = qlen[i];      // (in hand-out order the first contig need not be the longest)
	if (n < n_ctx && !dev_q && !(flags & GSA_MANY_NO_SPLIT) && longest >= split_min) {
		std::vector<int> gsz((size_t)n, 1);
		for (int left = n_ctx - n; left > 0; left--) {       // the next context goes where a context has the most bases to itself
			int best = -1; double load = 0;
			for (int i = 0; i < n; i++) { const int32_t ci = order[(size_t)i]; if ((int64_t)qlen[ci] < split_min) continue; const double l = (double)qlen[ci] / gsz[(size_t)i]; if (l > load) { load = l; best = i; } }
			if (best < 0) break;
			gsz[(size_t)best]++;
		}
		std::vector<std::thread> th; int at = 0;
		for (int i = 0; i < n; i++) {
			const int32_t ci = order[(size_t)i]; gsa_ctx *const *grp = ctx + at; const int ng = gsz[(size_t)i]; at += ng;
			th.emplace_back([&, ci, grp, ng] {
				gsa_result res;
				int rc = ng > 1 ? align_split(grp, ng, query[ci], qlen[ci], &res) : gsa_align_contig(grp[0], query[ci], qlen[ci], &res);
				if (rc == GSA_OK && on_result) rc = on_result(user, ci, &res);
				if (rc != GSA_OK) { int ok = GSA_OK; err.compare_exchange_strong(ok, rc); }
			});
		}
		for (std::thread &t : th) t.join();
		return err.load();
	}
	// Short contigs travel in BUNDLES: a contig of a few Mb is ~60 GPU operations whatever its size, and with several contexts in
	// flight the operations of one stretch the other's (profiles/archive/r03_timeline_multi_*.txt), so n short contigs are concatenated and
	// go through the stages as ONE pass (set_query_bundle / gsa_align_bundle; results per contig, identical to the ones they get
	// alone).  A unit of work = a long contig, or a bundle of consecutive short ones in hand-out order.  Bundle size: measured on
	// 5 Mb contigs, larger is better all the way (6 / 11 / 16 / 21 / 31 / 60 Mb per bundle: 5.3 / 7.6 / 9.7 / 10.8 / 13.9 / 16.0 Gbp/s on
	// one context, profiles/archive/r03_bundle_sweep.txt), so the short contigs are dealt into as few bundles as GSA_BUNDLE_CAP (default
	// 64 Mb) allows, rounded up to a multiple of the context count (one equal share per context).  Contigs above
	// GSA_BUNDLE_CONTIG (default 16 Mb) stay alone; GSA_BUNDLE_CONTIG=0 or GSA_MANY_NO_BUNDLE: no bundles.
	const int64_t bundle_contig = ctx[0]->opt.bundle_contig, bundle_cap = ctx[0]->opt.bundle_cap;
	std::vector<std::vector<int32_t> > units;
	{
		bool may = !(flags & GSA_MANY_NO_BUNDLE) && bundle_contig > 0;
		for (int k = 0; k < n_ctx; k++) if (ctx[k]->profiling || ctx[k]->count_blocks) may = false;      // (per-contig counters and stage timers)
		int64_t small_total = 0; int32_t n_small = 0;
		for (int32_t i = 0; i < n; i++) if ((int64_t)qlen[i] <= bundle_contig) { small_total += qlen[i]; n_small++; }
		if (n_small < 2) may = false;
		// the 64-bit seed key holds (contig x stride + PosDiff, position in the bundle): against a large reference (stride ~ 2G) only
		// so many contigs fit one bundle -- 32 against a human genome
		size_t max_contigs = GSA_BUNDLE_MAX_CONTIGS;
		{
			const int qb = ceil_log2_u64((u64)(bundle_cap + bundle_cap / 8 + bundle_contig) + 1);
			const u64 stride = (u64)(2 * ctx[0]->G) + (u64)bundle_contig + (u64)ctx[0]->prm.MaxIndelSize + 96;
			const u64 fit = qb < 62 ? (1ull << (63 - qb)) / stride : 0;
			if (fit < max_contigs) max_contigs = (size_t)fit;
			if (max_contigs < 2) may = false;
		}
#ifdef GSA_EXPERIMENTS
		static const int64_t bundle_target = [] { const char *e = getenv("GSA_BUNDLE_TARGET"); return e ? (int64_t)atoll(e) : 0ll; }();      // (a fixed bundle size)
#else
		const int64_t bundle_target = 0;
#endif
		int64_t n_bundles = (small_total + bundle_cap - 1) / (bundle_cap > 0 ? bundle_cap : 1); if (n_bundles < 1) n_bundles = 1;
		n_bundles = (n_bundles + n_ctx - 1) / n_ctx * n_ctx;
		int64_t target = small_total / n_bundles + 1;
		if (bundle_target > 0) target = bundle_target;
		int64_t cur = 0;
		for (int32_t i = 0; i < n; i++) {
			const int32_t ci = order[(size_t)i];
			const bool small = may && (int64_t)qlen[ci] <= bundle_contig;
			if (!small) { units.push_back(std::vector<int32_t>(1, ci)); cur = 0; continue; }
			const int64_t padded = ((int64_t)qlen[ci] + GSA_CHUNK - 1) / GSA_CHUNK * GSA_CHUNK;
			if (cur > 0 && (cur + padded / 2 > target || cur + padded > bundle_cap + bundle_cap / 8 || units.back().size() >= max_contigs)) cur = 0;      // (to the nearest contig)
			if (cur == 0) units.push_back(std::vector<int32_t>());
			units.back().push_back(ci); cur += padded > 0 ? padded : 1;
		}
	}
	const int32_t n_units = (int32_t)units.size();
	std::atomic<int32_t> next(0);
	auto one = [&](gsa_ctx *c, int32_t ci) {
		gsa_result res;
		int rc = dev_q ? gsa_align_contig_device(c, query[ci], qlen[ci], &res) : gsa_align_contig(c, query[ci], qlen[ci], &res);
		if (rc == GSA_OK && on_result) rc = on_result(user, ci, &res);
		return rc;
	};
	// A context works on unit u while unit u + 1 -- claimed one ahead -- is uploaded into its other query slot (gsa_prefetch_contig /
	// gsa_prefetch_bundle): the reference reads a sequence from host memory when its turn comes (GSAlign.cpp:483-490); here the H2D
	// copy of a chromosome (4.8 ms per 250 Mb) would otherwise sit in front of every contig's seed search.
	const bool pre = !dev_q && !(flags & GSA_MANY_NO_PREFETCH);
	auto loop = [&](gsa_ctx *c) {
		std::vector<const char *> bq, pq; std::vector<int32_t> bl, pl; std::vector<gsa_result> br;
		auto fetch = [&](int32_t u) {
			if (!pre || u >= n_units) return;
			const std::vector<int32_t> &un = units[(size_t)u];
			if (un.size() == 1) (void)gsa_prefetch_contig(c, query[un[0]], qlen[un[0]]);
			else { pq.clear(); pl.clear(); for (int32_t ci : un) { pq.push_back(query[ci]); pl.push_back(qlen[ci]); } (void)gsa_prefetch_bundle(c, pq.data(), pl.data(), (int32_t)un.size()); }
			// (a prefetch that fails -- memory -- leaves nothing pending: the contig is uploaded when its turn comes)
		};
		(void)gsa_cancel_prefetch(c);
		int32_t u = next.fetch_add(1);
		fetch(u);
		for (;;) {
			if (u >= n_units || err.load() != GSA_OK) { (void)gsa_cancel_prefetch(c); return; }
			const int32_t u_next = next.fetch_add(1);
			fetch(u_next);
			const std::vector<int32_t> &un = units[(size_t)u];
			int rc = GSA_OK;
			if (un.size() == 1) rc = one(c, un[0]);
			else {
				bq.clear(); bl.clear(); br.resize(un.size());
				for (int32_t ci : un) { bq.push_back(query[ci]); bl.push_back(qlen[ci]); }
				rc = gsa_align_bundle(c, bq.data(), bl.data(), (int32_t)un.size(), dev_q ? GSA_MANY_DEVICE : 0u, br.data());
				if (rc == GSA_OK) { for (size_t k = 0; k < un.size() && rc == GSA_OK && on_result; k++) rc = on_result(user, un[k], &br[k]); }
				else if (rc == GSA_ERR_LIMIT) { rc = GSA_OK; for (size_t k = 0; k < un.size() && rc == GSA_OK; k++) rc = one(c, un[k]); }      // (a capacity of the joint pass: one by one)
			}
			if (rc != GSA_OK) { int ok = GSA_OK; err.compare_exchange_strong(ok, rc); (void)gsa_cancel_prefetch(c); return; }
			u = u_next;
		}
	};
	std::vector<std::thread> th;
	// (thread placement is the caller's: gsa_bind_host_thread; the threads started here inherit the caller's affinity)
	for (int k = 1; k < n_ctx && k < n_units; k++) th.emplace_back(loop, ctx[k]);
	loop(ctx[0]);
	for (std::thread &t : th) t.join();
	return err.load();
}

int64_t gsa_seed_count(gsa_ctx *c) { return c ? c->n_seeds : 0; }

int gsa_get_seeds(gsa_ctx *c, gsa_seed *out)
{
	if (!c || (!out && c->n_seeds)) return GSA_ERR_ARG;
	if (c->stage < 1) return gsa_fail(c, GSA_ERR_STATE, "run stage 1 first");
	const size_t n = (size_t)c->n_seeds; if (!n) return GSA_OK;
	{ int rc = seed_view_sort(c); if (rc) return rc; GSA_CHECK(c, hipStreamSynchronize(c->stream)); }
	std::vector<i32> q(n), l(n); std::vector<i64> r(n);
	GSA_CHECK(c, hipMemcpy(q.data(), c->s_q.p, n * 4, hipMemcpyDeviceToHost));
	GSA_CHECK(c, hipMemcpy(l.data(), c->s_len.p, n * 4, hipMemcpyDeviceToHost));
	GSA_CHECK(c, hipMemcpy(r.data(), c->s_r.p, n * 8, hipMemcpyDeviceToHost));
	for (size_t i = 0; i < n; i++) { out[i].qpos = q[i]; out[i].len = l[i]; out[i].rpos = r[i]; }
	return GSA_OK;
}

int gsa_group_count(gsa_ctx *c)
{
	if (!c) return 0;
	if (c->n_groups < 0) {      // stage 1 leaves the count on the device
		if (seed_view_sort(c) != GSA_OK) return 0;
		hipStreamSynchronize(c->stream);
		i32 ng = 0;
		if (hipMemcpy(&ng, c->d_mail.as<i32>() + M_NG, 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
		c->n_groups = ng;
	}
	return c->n_groups;
}

int gsa_get_groups(gsa_ctx *c, int32_t *beg, int32_t *end)
{
	if (!c) return GSA_ERR_ARG;
	if (c->stage < 1) return gsa_fail(c, GSA_ERR_STATE, "run stage 1 first");
	const size_t ng = (size_t)gsa_group_count(c); if (!ng) return GSA_OK;
	std::vector<i32> gb(ng + 1);
	GSA_CHECK(c, hipMemcpy(gb.data(), c->g_beg.p, (ng + 1) * 4, hipMemcpyDeviceToHost));
	for (size_t i = 0; i < ng; i++) { beg[i] = gb[i]; end[i] = gb[i + 1]; }
	return GSA_OK;
}

int gsa_get_blocks(gsa_ctx *c, gsa_result *out)
{
	if (!c || !out) return GSA_ERR_ARG;
	if (c->stage < 2) return gsa_fail(c, GSA_ERR_STATE, "run stage 2 first");
	if (c->bnd.n && c->stage < 8) return gsa_fail(c, GSA_ERR_STATE, "gsa_get_blocks: the context holds a bundle -- only its finished result (gsa_align_bundle's out[]) is defined per contig");
	if (c->frags_stage != c->stage) { int rc = build_block_view(c); if (rc) return rc; }
	out->n_blocks = (int32_t)c->h_blocks.size(); out->blocks = c->h_blocks.data();
	if (c->result_pinned && c->stage == 8) {
		out->n_frags = c->n_frags; out->n_aln = c->n_aln;
		out->recs = c->p_frags.as<gsa_rec>(); out->aln1 = c->h_taln1; out->aln2 = c->h_taln2;
	} else {
		out->n_frags = (int64_t)c->h_frags.size(); out->n_aln = (int64_t)c->h_aln1.size();
		out->recs = c->h_frags.data(); out->aln1 = c->h_aln1.data(); out->aln2 = c->h_aln2.data();
	}
	return GSA_OK;
}

int gsa_get_counters(gsa_ctx *c, uint64_t counters[8]) { if (!c) return GSA_ERR_ARG; memcpy(counters, c->counters, sizeof(c->counters)); return GSA_OK; }
int gsa_get_timings(gsa_ctx *c, float ms[8]) { if (!c) return GSA_ERR_ARG; memcpy(ms, c->kernel_ms, sizeof(c->kernel_ms)); if (c->prof_seed && !c->profiling) ms[6] = (float)c->acc_seed_ms; return GSA_OK; }
int gsa_get_seed_stats(gsa_ctx *c, uint64_t st[8]) { if (!c) return GSA_ERR_ARG; memcpy(st, c->dbg, sizeof(c->dbg)); return GSA_OK; }

} // extern "C"
