// gsalign_amd/csrc/gsa_prim.hip -- the three library primitives the pipeline
// uses from rocPRIM: LSD radix sort of (u64 key, u32 value) pairs and exclusive
// prefix sums.  Everything else in the pipeline is hand-written kernels.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "gsa_ctx.h"

static int ensure_tmp(gsa_ctx *c, size_t bytes)
{
	if (bytes <= c->tmp.cap) return GSA_OK;
	if (c->tmp.p) { ctx_quiesce(c); hipFree(c->tmp.p); c->tmp.p = nullptr; c->tmp.cap = 0; }
	size_t want = bytes + bytes / 4 + 4096;
	if (hipMalloc(&c->tmp.p, want) != hipSuccess) return gsa_fail(c, GSA_ERR_NOMEM, "hipMalloc(prim temp)");
	c->tmp.cap = want;
	return GSA_OK;
}

int prim_sort_pairs_u64_u32(gsa_ctx *c, const u64 *kin, u64 *kout, const u32 *vin, u32 *vout, size_t n, int begin_bit, int end_bit)
{
	if (n == 0) return GSA_OK;
	if (end_bit <= begin_bit) end_bit = begin_bit + 1;
	size_t bytes = 0;
	// At a bacterial contig's 75 k pairs rocPRIM runs its merge sort (Onesweep forced by config is 100 us slower there):
	// 2048-item block sorts save two merge launches against the default tuning (measured: 512 x 4 best of five shapes)
	using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<512, 512, 4>, rocprim::default_config>;
	if (n <= (1u << 18)) {
		GSA_CHECK(c, rocprim::radix_sort_pairs<cfg>(nullptr, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, c->stream));
		int rc = ensure_tmp(c, bytes); if (rc) return rc;
		GSA_CHECK(c, rocprim::radix_sort_pairs<cfg>(c->tmp.p, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, c->stream));
		return GSA_OK;
	}
	// (larger inputs: the library's own tuning)
	GSA_CHECK(c, rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, c->stream));
	int rc = ensure_tmp(c, bytes); if (rc) return rc;
	GSA_CHECK(c, rocprim::radix_sort_pairs(c->tmp.p, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, c->stream));
	return GSA_OK;
}

int prim_exscan_i32(gsa_ctx *c, const i32 *in, i32 *out, size_t n)
{
	if (n == 0) return GSA_OK;
	size_t bytes = 0;
	GSA_CHECK(c, rocprim::exclusive_scan(nullptr, bytes, in, out, (i32)0, n, rocprim::plus<i32>(), c->stream));
	int rc = ensure_tmp(c, bytes); if (rc) return rc;
	GSA_CHECK(c, rocprim::exclusive_scan(c->tmp.p, bytes, in, out, (i32)0, n, rocprim::plus<i32>(), c->stream));
	return GSA_OK;
}

int prim_exscan_i32_i64(gsa_ctx *c, const i32 *in, i64 *out, size_t n)
{
	if (n == 0) return GSA_OK;
	size_t bytes = 0;
	GSA_CHECK(c, rocprim::exclusive_scan(nullptr, bytes, in, out, (i64)0, n, rocprim::plus<i64>(), c->stream));
	int rc = ensure_tmp(c, bytes); if (rc) return rc;
	GSA_CHECK(c, rocprim::exclusive_scan(c->tmp.p, bytes, in, out, (i64)0, n, rocprim::plus<i64>(), c->stream));
	return GSA_OK;
}
