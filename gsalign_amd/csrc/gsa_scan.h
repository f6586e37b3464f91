// gsalign_amd/csrc/gsa_scan.h -- single-pass "flag -> exclusive scan -> emit" kernels.
//
// The chaining / refinement / extension stages are long chains of tiny data-parallel
// passes (compute a flag per element, prefix-sum it, scatter by the result).  As
// separate launches every link costs a dependent dispatch (4-5 us on MI355X, more
// than its work at bacterial sizes), so each flag + scan + scatter triple is ONE
// launch here: a decoupled look-back scan (Merrill & Garland) whose tile status words
// travel through HBM as self-validating 64-bit words {epoch, flag, value} with
// agent-scope relaxed atomics -- the same hand-off as the DP boundary granules: no
// fence, no re-initialisation between launches (the epoch invalidates old words).
// Tiles take their number from a ticket counter in dispatch order, so a tile only
// ever waits for tiles that are already running.
#ifndef GSA_SCAN_H
#define GSA_SCAN_H
#include <algorithm>
#include "gsa_ctx.h"

#ifndef LB_TPB
#define LB_TPB 256
#endif
#ifndef LB_GRID_PER_CU
#define LB_GRID_PER_CU 2    // workgroups per CU of a fused pass (persistent: tiles are drawn from the ticket counter)
#endif
#ifndef LB_BATCH
#define LB_BATCH 4         // elements of a thread whose loads are in flight together (clamped Ops)
#endif
#ifndef LB_ITEMS
#define LB_ITEMS 8          // elements per thread (default).  Round 4 (late): 4 -> 8 once the passes read an element's inputs in front of the scan (Item): a tile's fixed
                            // cost -- ticket, barriers, look-back -- is what a pass pays; 2 / 4 / 8 / 12 / 16 per thread: 3.06 / 2.29 / 1.86 / 1.81 / 3.49 ms for the
                            // 21 passes of a 250 Mb contig (16 spills); the hash-table pass stays at 4, the record pass takes 12
#endif

#ifdef LB_TIMING      // (experiment build, one translation unit: where a tile's time goes -- sums of 100 MHz ticks over all tiles of all passes; gsa_debug_lb_prof reads them)
static __device__ unsigned long long g_lb_prof[8];
#define LB_T(K) { if (tid == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_lb_prof[K], t_ - t_last); t_last = t_; } }
#else
#define LB_T(K)
#endif
struct LbArgs {
	unsigned long long *status[2];   // tile status words, one array per scanned component
	u32 *ticket;                     // never reset: tile = ticket - base
	u32 base, epoch; i32 n_tiles;
	i32 *err;                        // set when a spin runs into its bound (a bug, not a state)
	u32 *finished;                   // tiles that are through (only Ops with a finish() hook count; the last one resets it)
};
#include <type_traits>
template <class T, class = void> struct lb_has_finish : std::false_type {};
template <class T> struct lb_has_finish<T, std::void_t<decltype(&T::finish)>> : std::true_type {};
// Ops with an Item: `Item load(i)` reads everything value() and emit() need about element i ONCE, in front of the scan -- the loads of a
// thread's ITEMS elements are independent and go out together -- and `value(item, i, c)` / `emit(item, i, v, ex)` work from the registers.
// Without it emit() re-reads its inputs behind the look-back (atomics: nothing loaded earlier may be kept) and, its pointers being plain,
// behind each of its own stores: ~10 dependent L2 round trips per element, ITEMS times in a row (OpDpJobs: 64 us per tile, 98 % of its
// wave-cycles waiting -- profiles/archive/r04_sq_human.txt).
template <class T, class = void> struct lb_has_item : std::false_type {};
template <class T> struct lb_has_item<T, std::void_t<typename T::Item>> : std::true_type {};
// Ops that declare `static constexpr bool clamped = true` promise that load(i) / value(i, c) may be called for ANY i in [0, n) without a guard and contain no
// branch around a load (neighbours are read at clamped indices and selected afterwards).  The pass then calls them unconditionally -- for the elements past the end
// at index n - 1, result ignored -- so that the ITEMS elements of a thread are one basic block and their loads go out level by level, all elements together.
// (Round 5: with a guard per element, and `a && b[i - 1]` inside the Ops, every load was waited for on its own -- ~100 loads and ~100 s_waitcnt vmcnt per pass,
// the elements one after the other: half of a tile's 42 us, profiles/r05_lb_pass_experiments.txt.)
template <class T, class = void> struct lb_is_clamped : std::false_type {};
template <class T> struct lb_is_clamped<T, std::void_t<decltype(T::clamped)>> : std::integral_constant<bool, T::clamped> {};
// optional `void prep(Item &it, i64 i)`: run for every element after ALL loads of the thread are issued -- the place for branches, divisions and rare slow paths
template <class T, class = void> struct lb_has_prep : std::false_type {};
template <class T> struct lb_has_prep<T, std::void_t<decltype(&T::prep)>> : std::true_type {};
// every 32-bit word of a loaded Item is "used" right behind the loads: without that the compiler sinks a load into the conditional block that needs its value
// (`b = u && ... r - q != ...`: the two a_r loads moved behind a branch on u, one branch per element, and the elements were serial again)
template <class T> __device__ __forceinline__ void lb_pin(T &x)
{
	static_assert(sizeof(T) % 4 == 0, "Items are made of 32- and 64-bit fields");
	u32 w[sizeof(T) / 4];
	__builtin_memcpy(w, &x, sizeof(T));
#pragma unroll
	for (unsigned k = 0; k < sizeof(T) / 4; k++) asm volatile("" : "+v"(w[k]));
	__builtin_memcpy(&x, w, sizeof(T));
}
// optional `emit_all(items, i0, in[], v[], ex[])` instead of emit(): all ITEMS rows of the thread at once (row k is element i0 + k * LB_TPB, in[k]: it exists)
template <class T, class = void> struct lb_has_emit_all : std::false_type {};
template <class T> struct lb_has_emit_all<T, std::void_t<decltype(T::has_emit_all)>> : std::true_type {};
template <class T, bool = lb_has_item<T>::value> struct lb_item_of { struct type {}; };
template <class T> struct lb_item_of<T, true> { using type = typename T::Item; };

__device__ __forceinline__ unsigned long long lb_pack(u32 epoch, u32 flag, u32 val) { return ((unsigned long long)(epoch & 0x3fffffffu) << 34) | ((unsigned long long)flag << 32) | val; }

// exclusive prefixes of tile `tile` for all NV components at once; agg[c] = the tile's totals (same in all threads).
// Must be called by every thread of the workgroup.  Wave 0 walks back through the predecessors' status words, 64 per window.
// Round 4 (late): (a) the components are looked up TOGETHER -- their loads are in flight at the same time; one after the other a
// two-component pass paid the walk twice; (b) a window is settled as soon as every tile between this one and the NEAREST
// predecessor that knows its inclusive prefix has published its total: the tiles further back than that one do not matter, and
// waiting for all 64 of a window meant waiting for the slowest of 64 workgroups.
template <int NV>
__device__ __forceinline__ void lb_tile_prefix(const LbArgs &lb, int tile, const i32 (&agg)[NV], i32 (&pre)[NV], i32 *s_bcast)
{
	const int tid = threadIdx.x, lane = tid & 63;
	const u32 ep = lb.epoch & 0x3fffffffu;
	if (tile == 0) {
		if (tid == 0) {
#pragma unroll
			for (int c = 0; c < NV; c++) { __hip_atomic_store(&lb.status[c][0], lb_pack(ep, 2, (u32)agg[c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_bcast[c] = 0; }
		}
	} else if (tid < 64) {
		if (lane == 0) {
#pragma unroll
			for (int c = 0; c < NV; c++) __hip_atomic_store(&lb.status[c][tile], lb_pack(ep, 1, (u32)agg[c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		i32 excl[NV]; bool fin[NV];                                   // (fin: wave-uniform)
#pragma unroll
		for (int c = 0; c < NV; c++) { excl[c] = 0; fin[c] = false; }
#ifndef LB_WAIT_TICKS
#define LB_WAIT_TICKS 500000000ull      // 5 s of the 100 MHz wall clock: the bound of a look-back wait (a bug, not a state, when it trips)
#endif
#ifndef LB_WIN
#define LB_WIN 64      // (experiment: predecessors looked at per round trip)
#endif
		for (int base = tile - 1;; base -= LB_WIN) {
			const int idx = base - lane;
			unsigned long long w[NV];
			int first[NV];
			u32 spins = 0; unsigned long long t_spin0 = 0;
			for (;;) {
				bool settled = true;
#pragma unroll
				for (int c = 0; c < NV; c++) {
					w[c] = lb_pack(ep, 2, 0);                            // tiles before tile 0: prefix 0
					first[c] = 0;
					if (fin[c]) continue;
					if (LB_WIN < 64 && lane >= LB_WIN) w[c] = lb_pack(ep, 1, 0);      // (not looked at: a total of nothing)
					else if (idx >= 0) w[c] = __hip_atomic_load(&lb.status[c][idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
#pragma unroll
				for (int c = 0; c < NV; c++) {
					if (fin[c]) continue;
					const bool rdy = (u32)(w[c] >> 34) == ep && ((w[c] >> 32) & 3) != 0;
					const unsigned long long p2 = __ballot(rdy && ((w[c] >> 32) & 3) == 2), nr = __ballot(!rdy);
					first[c] = p2 ? __ffsll((long long)p2) - 1 : 64;     // nearest predecessor that already knows its inclusive prefix
					const unsigned long long need = first[c] >= 63 ? ~0ull : ((2ull << first[c]) - 1ull);      // lanes 0 .. first
					if (nr & need) settled = false;
				}
				if (settled) break;
				// (the bound is wall-clock time -- LB_WAIT_TICKS of the 100 MHz clock = 5 s -- not a number of looks: a predecessor is always a tile that RUNS, but since round 6 kernels of
				//  up to eight contexts share a CU and a running wave can be kept from issuing for a long time; a count of 2^22 looks was ~0.8 s and tripped once in ~60 runs of ten contexts)
				if ((++spins & 1023u) == 0 && (t_spin0 == 0 ? (t_spin0 = wall_clock64(), false) : wall_clock64() - t_spin0 > LB_WAIT_TICKS)) {
					if (lane == 0) *lb.err = 1;
#pragma unroll
					for (int c = 0; c < NV; c++) { w[c] = lb_pack(ep, 2, 0); first[c] = 0; }
					break;
				}
				__builtin_amdgcn_s_sleep(1);
			}
			bool all_fin = true;
#pragma unroll
			for (int c = 0; c < NV; c++) {
				if (fin[c]) continue;
				i32 v = lane <= first[c] ? (i32)(u32)w[c] : 0;
				for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
				excl[c] += v;
				if (first[c] < 64) fin[c] = true; else all_fin = false;
			}
			if (all_fin) break;
		}
		if (lane == 0) {
#pragma unroll
			for (int c = 0; c < NV; c++) { __hip_atomic_store(&lb.status[c][tile], lb_pack(ep, 2, (u32)(excl[c] + agg[c])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_bcast[c] = excl[c]; }
		}
	}
	__syncthreads();
#pragma unroll
	for (int c = 0; c < NV; c++) pre[c] = s_bcast[c];
}

// a store that the finish() hook of a pass (run by whichever workgroup is through last, possibly on another XCD) will read
__device__ __forceinline__ void lb_pub(i32 *p, i32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// for finish() hooks: n words that other workgroups published with lb_pub() -> pinned host memory, by the whole workgroup.  Eight agent-scope loads in flight per
// thread before the first store (one load, one store, one load ... was a round trip past the L2 per 256 words: 48 of them for the first 4 096 large gaps of a contig)
__device__ __forceinline__ void lb_copy_out(i32 *dst, const i32 *src, i32 n, int tid)
{
	for (i32 t0 = 0; t0 < n; t0 += 8 * LB_TPB) {
		i32 v[8];
#pragma unroll
		for (int u = 0; u < 8; u++) { const i32 t = t0 + u * LB_TPB + tid; v[u] = t < n ? __hip_atomic_load(&src[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0; }
#pragma unroll
		for (int u = 0; u < 8; u++) { const i32 t = t0 + u * LB_TPB + tid; if (t < n) dst[t] = v[u]; }
	}
}

// Generic fused pass over i in [0, n):  v = op.value(i, k)  (NV components, k < NV),
// ex = exclusive prefix sums, then op.emit(i, v, ex);  op.done(totals) once, by the thread
// that owns the last element (or thread 0 of tile 0 when n == 0).
#ifndef LB_MIN_WAVES
#define LB_MIN_WAVES 1      // (experiment: waves per SIMD the register allocation of a fused pass must allow -- 8: at most 64 VGPRs, 10: 48)
#endif
#ifdef LB_BISECT      // (diagnosis build: the register bound on ONE Op -- the one whose lb_id is LB_BISECT -- to find which pass miscomputes under it)
template <class T, class = void> struct lb_id_of { static constexpr int value = -1; };
template <class T> struct lb_id_of<T, std::void_t<decltype(T::lb_id)>> { static constexpr int value = T::lb_id; };
#define LB_BOUND_OF(Op_) (lb_id_of<Op_>::value == (LB_BISECT) ? 4 : 1)
#else
#define LB_BOUND_OF(Op_) LB_MIN_WAVES
#endif
template <int NV, class Op, int ITEMS>
__global__ void __launch_bounds__(LB_TPB, LB_BOUND_OF(Op)) k_lb_pass(i64 n, Op op, LbArgs lb)
{
	constexpr int LB_TILE = LB_TPB * ITEMS;
	__shared__ i32 s_tile, s_bcast[2], s_wsum[NV][ITEMS][LB_TPB / 64];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	// PERSISTENT since round 4: the grid is at most LB_GRID_PER_CU workgroups per CU and every workgroup draws tile after tile.  A full
	// grid put ~2000 tiles on the chip at once, none of which had its inclusive prefix yet, so a tile walked back through up to 30 windows
	// of 64 aggregates -- a dependent agent-scope load each -- before it met one (58 us mean tile lifetime in OpDpJobs: r03_sq_human.txt,
	// 2.2 x 10^9 wave-cycles per contig, more than the striped DP).  With a few hundred tiles in flight, all of them consecutive, the
	// predecessors of a tile are mostly through and the first window answers.
#ifdef LB_TIMING
	unsigned long long t_last = wall_clock64();
#endif
	for (bool first = true;; first = false) {
	if (!first && (i32)gridDim.x >= lb.n_tiles) break;      // (a launch with a workgroup per tile: one draw each, no failing second one -- a round trip per pass on small contigs)
	__syncthreads();
	// (tiles by workgroup number instead of the ticket -- round 5 experiment -- hang as soon as other kernels share the chip: a workgroup waits for a tile whose
	//  workgroup has not been dispatched yet; profiles/r05_lb_pass_experiments.txt)
	if (tid == 0) s_tile = (i32)(atomicAdd(lb.ticket, 1u) - lb.base);
	__syncthreads();
	const int tile = s_tile;
	LB_T(0)      // ticket (+ the barriers around it)
	if (tile >= lb.n_tiles) break;
	// STRIPED since round 4: item k of thread t is element tile * LB_TILE + k * LB_TPB + t, so the 64 lanes of a wave read 64 consecutive
	// elements of every array an Op touches (with consecutive items per thread a wave's load touched 64 lines at 16 items per thread and ran
	// at the address unit's pace: chain 2.0 -> 2.6 ms when tried).  The tile is then ITEMS rows of LB_TPB elements: a row is scanned across
	// the workgroup (wave scan + wave totals through LDS, one barrier for all rows), rows follow each other.
	const i64 i0 = (i64)tile * LB_TILE + tid;
	i32 v[NV][ITEMS], inc[NV][ITEMS];
	typename lb_item_of<Op>::type item[ITEMS];
	if constexpr (lb_has_item<Op>::value) {
		if constexpr (lb_is_clamped<Op>::value) {
			// LB_BATCH elements at a time: their loads together, pinned, then their prep() -- the raw values of a batch are dead before the next one is loaded
			// (all ITEMS = 8 at once: 180 - 250 VGPRs, most of them the 64-bit addresses of ~56 loads in flight)
			if (n > 0) {
#pragma unroll
				for (int k0 = 0; k0 < ITEMS; k0 += LB_BATCH) {
#pragma unroll
					for (int k = k0; k < k0 + LB_BATCH && k < ITEMS; k++) { const i64 i = i0 + (i64)k * LB_TPB; item[k] = op.load(i < n ? i : n - 1); }
#pragma unroll
					for (int k = k0; k < k0 + LB_BATCH && k < ITEMS; k++) lb_pin(item[k]);
					if constexpr (lb_has_prep<Op>::value) {
#pragma unroll
						for (int k = k0; k < k0 + LB_BATCH && k < ITEMS; k++) { const i64 i = i0 + (i64)k * LB_TPB; if (i < n) op.prep(item[k], i); }
					}
				}
			}
		} else {
#pragma unroll
			for (int k = 0; k < ITEMS; k++) { const i64 i = i0 + (i64)k * LB_TPB; if (i < n) item[k] = op.load(i); }
			if constexpr (lb_has_prep<Op>::value) {
#pragma unroll
				for (int k = 0; k < ITEMS; k++) { const i64 i = i0 + (i64)k * LB_TPB; if (i < n) op.prep(item[k], i); }
			}
		}
	}
#pragma unroll
	for (int k = 0; k < ITEMS; k++) {
		const i64 i = i0 + (i64)k * LB_TPB;
#pragma unroll
		for (int c = 0; c < NV; c++) {
			if constexpr (lb_has_item<Op>::value) v[c][k] = i < n ? op.value(item[k], i, c) : 0;
			else v[c][k] = i < n ? op.value(i, c) : 0;
		}
	}
#pragma unroll
	for (int c = 0; c < NV; c++)
#pragma unroll
		for (int k = 0; k < ITEMS; k++) {
			i32 x = v[c][k];
			for (int o = 1; o < 64; o <<= 1) { const i32 t = __shfl_up(x, o); if (lane >= o) x += t; }
			inc[c][k] = x;
			if (lane == 63) s_wsum[c][k][wv] = x;
		}
	__syncthreads();
	i32 agg[NV];
#pragma unroll
	for (int c = 0; c < NV; c++) {
		i32 run = 0;                      // elements of the rows in front of row k
#pragma unroll
		for (int k = 0; k < ITEMS; k++) {
			i32 wo = 0, tot = 0;
#pragma unroll
			for (int w = 0; w < LB_TPB / 64; w++) { const i32 sw = s_wsum[c][k][w]; if (w < wv) wo += sw; tot += sw; }
			inc[c][k] += run + wo - v[c][k];      // exclusive prefix of my element inside the tile
			run += tot;
		}
		agg[c] = run;
	}
	i32 pre[NV];
	LB_T(1)      // loads + values + the tile's own scan
	lb_tile_prefix<NV>(lb, tile, agg, pre, s_bcast);
	LB_T(2)      // look-back
#ifdef LB_TIMING
	if (tid == 0) atomicAdd(&g_lb_prof[4], 1ull);
#endif
	i32 vv[NV], ee[NV];
	if constexpr (lb_has_emit_all<Op>::value) {
		// (single-component Ops whose emit() has round trips of its own -- atomics with a result: the rows of a thread go through them together)
		static_assert(NV == 1, "emit_all: one component");
		i32 ex0[ITEMS]; bool in[ITEMS];
#pragma unroll
		for (int k = 0; k < ITEMS; k++) { ex0[k] = pre[0] + inc[0][k]; in[k] = i0 + (i64)k * LB_TPB < n; }
		op.emit_all(item, i0, in, v[0], ex0);
#pragma unroll
		for (int k = 0; k < ITEMS; k++) if (i0 + (i64)k * LB_TPB == n - 1) { i32 tt[1] = { ex0[k] + v[0][k] }; op.done(tt); }
	} else {
#pragma unroll
	for (int k = 0; k < ITEMS; k++) {
		const i64 i = i0 + (i64)k * LB_TPB;
#pragma unroll
		for (int c = 0; c < NV; c++) { vv[c] = v[c][k]; ee[c] = pre[c] + inc[c][k]; }
		if (i < n) {
			if constexpr (lb_has_item<Op>::value) op.emit(item[k], i, vv, ee);
			else op.emit(i, vv, ee);
			if (i == n - 1) { i32 tt[NV]; for (int c = 0; c < NV; c++) tt[c] = ee[c] + vv[c]; op.done(tt); }
		}
	}
	}
	if (n == 0 && tile == 0 && tid == 0) { i32 tt[NV]; for (int c = 0; c < NV; c++) tt[c] = 0; op.done(tt); }
#ifdef LB_TIMING
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
	LB_T(3)      // emit (stores drained)
	}
	// Ops with a finish(tid) hook: the workgroup that is through LAST runs it (all 256 threads) -- everything every tile
	// emitted is visible to it.  Used to put counts and list heads into pinned memory for the host without another launch.
	if constexpr (lb_has_finish<Op>::value) {
		__shared__ int s_last;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (every wave: its lb_pub stores are done before the tile counts itself)
		__syncthreads();
		if (tid == 0) {
			// No agent-scope fence here: on this part that is a write-back of the XCD's whole L2 per TILE (thousands per pass,
			// while the DP kernels beside it keep the L2s full of dirty direction bytes).  Instead, what finish() reads is
			// written with lb_pub() -- agent-scope atomic stores, coherent across the XCDs' L2s by themselves -- and is
			// complete (vmcnt) before this tile counts itself: the barrier above made every wave wait for its stores.
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			const u32 t = __hip_atomic_fetch_add(lb.finished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			s_last = t == gridDim.x - 1 ? 1 : 0;
			if (s_last) __hip_atomic_store(lb.finished, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		__syncthreads();
		if (s_last) op.finish(tid);
	}
}

// host side: one fused pass on the context's stream
template <int NV, int ITEMS = LB_ITEMS, class Op>
static inline int lb_launch(gsa_ctx *c, i64 n, const Op &op, hipStream_t stream = nullptr, int grid_per_cu = LB_GRID_PER_CU)
{
	if (!stream) stream = c->stream;      // (only one stream may run fused passes at a time: they share the status words)
	if (c->n_cus <= 0) { hipDeviceProp_t pr; GSA_CHECK(c, hipGetDeviceProperties(&pr, c->device)); c->n_cus = pr.multiProcessorCount; }
	// a short input (a bacterial contig's 75 000 seeds) keeps tiles of four elements per thread: while every tile finds a workgroup of its own,
	// smaller tiles are more of them at work
	if constexpr (ITEMS > 4) { if (n <= (i64)LB_TPB * 4 * grid_per_cu * c->n_cus) return lb_launch<NV, 4>(c, n, op, stream, grid_per_cu); }
	constexpr i64 LB_TILE = (i64)LB_TPB * ITEMS;
	const size_t tiles = n > 0 ? (size_t)((n + LB_TILE - 1) / LB_TILE) : 1;
	for (int k = 0; k < 2; k++) {
		if (c->d_lb_status[k].cap < tiles * 8) {
			if (!dev_ensure<unsigned long long>(c, c->d_lb_status[k], tiles + 1024)) return GSA_ERR_NOMEM;
			GSA_CHECK(c, hipMemsetAsync(c->d_lb_status[k].p, 0, c->d_lb_status[k].cap, stream));       // fresh memory: no word may look current
		}
	}
	c->lb_epoch++;
	if ((c->lb_epoch & 0x3fffffffu) == 0) {      // epoch wrapped: forget every old word
		for (int k = 0; k < 2; k++) GSA_CHECK(c, hipMemsetAsync(c->d_lb_status[k].p, 0, c->d_lb_status[k].cap, stream));
		c->lb_epoch = 1;
	}
	LbArgs lb;
	lb.status[0] = c->d_lb_status[0].as<unsigned long long>(); lb.status[1] = c->d_lb_status[1].as<unsigned long long>();
	lb.ticket = c->d_mail.as<u32>() + M_TICKET; lb.base = c->lb_base; lb.epoch = c->lb_epoch; lb.err = c->d_mail.as<i32>() + M_LBERR; lb.finished = c->d_mail.as<u32>() + M_LBDONE;
	lb.n_tiles = (i32)tiles;
	const size_t grid = std::min<size_t>(tiles, (size_t)grid_per_cu * (size_t)c->n_cus);
	hipLaunchKernelGGL((k_lb_pass<NV, Op, ITEMS>), dim3((unsigned)grid), dim3(LB_TPB), 0, stream, n, op, lb);
	GSA_CHECK(c, hipGetLastError());
#ifdef LB_SYNC_DEBUG      // (diagnosis build: every fused pass is waited for, the first one that faults names itself)
	{ const hipError_t e_ = hipStreamSynchronize(stream); if (e_ != hipSuccess) { fprintf(stderr, "[LB_SYNC_DEBUG] %s: n = %lld, tiles = %zu, grid = %zu: %s\n", __PRETTY_FUNCTION__, (long long)n, tiles, grid, hipGetErrorString(e_)); return gsa_fail(c, GSA_ERR_HIP, "fused pass failed (LB_SYNC_DEBUG)"); } }
#endif
	c->lb_base += (u32)(tiles + (grid < tiles ? grid : 0));      // (persistent: every workgroup's last draw is the one that fails)
	return GSA_OK;
}

#endif
