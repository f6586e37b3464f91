// tools/rand_probe.hip -- how many RANDOM reads per second does an MI355X serve?  The seed kernel's work is random 16-64-byte
// reads over tables of 0.5 - 17 GB (presence table, k-mer table, packed text, BWT blocks, dense SA); this measures the ceiling
// such a kernel can reach, by table size and by bytes per read, so that its rate can be put against something.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/rand_probe tools/rand_probe.hip && /tmp/rand_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>

template <int BYTES>
__global__ void __launch_bounds__(256) k_rand(const uint4 *__restrict__ tab, uint64_t lines, int rounds, int chain, uint32_t *sink)
{
	uint64_t x = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
	uint32_t acc = 0;
	for (int r = 0; r < rounds; r++) {
		// `chain` dependent reads (the next address needs the data: what a walk does), independent across lanes and waves
		uint64_t a = x;
		for (int c = 0; c < chain; c++) {
			a ^= a >> 29; a *= 0xBF58476D1CE4E5B9ull; a ^= a >> 32;
			const uint4 *p = BYTES > 64 ? tab + (a % (lines / 2)) * 8 : tab + (a % lines) * 4;           // a 64-byte line (128-byte aligned pair for BYTES = 128)
			uint4 v = p[0];
			if (BYTES >= 32) { const uint4 w = p[1]; v.x ^= w.x; v.y ^= w.y; }
			if (BYTES >= 64) { const uint4 w = p[2], z = p[3]; v.x ^= w.x ^ z.x; v.y ^= w.y ^ z.y; }
			if (BYTES >= 128) { const uint4 w = p[4], z = p[5], w2 = p[6], z2 = p[7]; v.x ^= w.x ^ z.x ^ w2.x ^ z2.x; v.y ^= w.y ^ z.y ^ w2.y ^ z2.y; }
			acc += v.x ^ v.y ^ v.z ^ v.w;
			a += v.x;                                         // (the table holds zeros: the dependency is real, the address is not disturbed)
		}
		x += 0x632BE59BD9B4E019ull;
	}
	if (acc == 0x12345678u) sink[0] = acc;
}

template <int BYTES>
static void run(const uint4 *tab, uint64_t bytes, int wg, int rounds, int chain, uint32_t *sink)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const uint64_t lines = bytes / 64;
	hipLaunchKernelGGL(k_rand<BYTES>, dim3(wg), dim3(256), 0, 0, tab, lines, 2, chain, sink);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k_rand<BYTES>, dim3(wg), dim3(256), 0, 0, tab, lines, rounds, chain, sink);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	const double reads = (double)wg * 256 * rounds * chain;
	printf("  table %7.2f GB  %2d B per read  %5d workgroups x 256 lanes, chain %2d: %7.2f G reads/s  %7.1f GB/s useful  (%.2f ms)\n",
	       bytes / 1e9, BYTES, wg, chain, reads / ms / 1e6, reads * BYTES / ms / 1e6, ms);
}

int main(int argc, char **argv)
{
	uint32_t *sink; hipMalloc(&sink, 64);
	if (argc > 1 && !strcmp(argv[1], "cal")) {
		// calibration of the TCC_EA0_RDREQ counters (round 4): reads of KNOWN number and width on a table far beyond the caches, one launch
		// per width, so that `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum` (tools/pmc_cal.sh) tells how many requests, of which
		// kind, a 16 / 32 / 64 / 128-byte random read costs: wg x 256 lanes x rounds reads per launch
		const uint64_t bytes = 18ull << 30; uint4 *tab = nullptr;
		if (hipMalloc(&tab, bytes) != hipSuccess) { printf("no memory\n"); return 1; }
		hipMemset(tab, 0, bytes); hipDeviceSynchronize();
		const int wg = 4096, rounds = 16;
		printf("calibration: %llu reads per launch\n", (unsigned long long)wg * 256 * rounds);
		hipLaunchKernelGGL(k_rand<16>, dim3(wg), dim3(256), 0, 0, tab, bytes / 64, rounds, 1, sink);
		hipLaunchKernelGGL(k_rand<32>, dim3(wg), dim3(256), 0, 0, tab, bytes / 64, rounds, 1, sink);
		hipLaunchKernelGGL(k_rand<64>, dim3(wg), dim3(256), 0, 0, tab, bytes / 64, rounds, 1, sink);
		hipLaunchKernelGGL(k_rand<128>, dim3(wg), dim3(256), 0, 0, tab, bytes / 64, rounds, 1, sink);
		hipDeviceSynchronize();
		return 0;
	}
	const uint64_t sizes[] = {64ull << 20, 512ull << 20, 4ull << 30, 17ull << 30, 48ull << 30};      // (argv[1] = one size in MB)
	for (uint64_t bytes : sizes) {
		if (argc > 1 && bytes != ((uint64_t)atoll(argv[1]) << 20)) continue;
		uint4 *tab = nullptr;
		if (hipMalloc(&tab, bytes) != hipSuccess) { printf("  (no %llu MB)\n", (unsigned long long)(bytes >> 20)); continue; }
		hipMemset(tab, 0, bytes);
		hipDeviceSynchronize();
		for (int wg : {1024, 4096, 16384}) {
			run<16>(tab, bytes, wg, 64, 1, sink);
			run<64>(tab, bytes, wg, 64, 1, sink);
		}
		run<16>(tab, bytes, 4096, 8, 8, sink);
		run<64>(tab, bytes, 4096, 8, 8, sink);
		// few lanes in flight (the seed kernel holds ~2 500 waves x 15 active lanes = 150 workgroups' worth): latency-bound rates
		for (int wg : {40, 150, 300, 600}) { run<16>(tab, bytes, wg, 64, 8, sink); run<32>(tab, bytes, wg, 64, 8, sink); }
		hipFree(tab);
	}
	return 0;
}
