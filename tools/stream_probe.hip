// tools/stream_probe.hip -- what does the access shape of the fused passes get from HBM?  NA arrays of n 4-byte elements are read and NW written by a persistent grid,
// tile after tile (tile = 256 threads x ITEMS elements), (a) striped: element k * 256 + t of the tile per load -- a wave's load is 256 contiguous bytes, one
// 4-byte load instruction per element and array, as k_lb_pass does -- (b) blocked: a thread reads 4 consecutive elements with one 16-byte load.
//   hipcc --offload-arch=gfx950 -O2 -o tools/stream_probe.bin tools/stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define ITEMS 8
struct Arr { const int *in[12]; int *out[6]; };
template <int NA, int NW, bool BLOCKED>
__global__ void __launch_bounds__(256) k_stream(Arr a, long n, long n_tiles)
{
	for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
		const long base = tile * 256 * ITEMS;
		int acc[ITEMS];
#pragma unroll
		for (int k = 0; k < ITEMS; k++) acc[k] = 0;
		if (BLOCKED) {
#pragma unroll
			for (int j = 0; j < NA; j++)
#pragma unroll
				for (int k = 0; k < ITEMS / 4; k++) { const int4 v = *(const int4 *)(a.in[j] + base + (long)k * 1024 + threadIdx.x * 4); acc[4 * k] += v.x; acc[4 * k + 1] += v.y; acc[4 * k + 2] += v.z; acc[4 * k + 3] += v.w; }
#pragma unroll
			for (int j = 0; j < NW; j++)
#pragma unroll
				for (int k = 0; k < ITEMS / 4; k++) *(int4 *)(a.out[j] + base + (long)k * 1024 + threadIdx.x * 4) = make_int4(acc[4 * k] + j, acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]);
		} else {
#pragma unroll
			for (int j = 0; j < NA; j++)
#pragma unroll
				for (int k = 0; k < ITEMS; k++) acc[k] += a.in[j][base + (long)k * 256 + threadIdx.x];
#pragma unroll
			for (int j = 0; j < NW; j++)
#pragma unroll
				for (int k = 0; k < ITEMS; k++) a.out[j][base + (long)k * 256 + threadIdx.x] = acc[k] + j;
		}
	}
}
template <int NA, int NW, bool B> static void run(const char *tag, Arr a, long n, int grid)
{
	const long n_tiles = n / (256 * ITEMS);
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	float best = 1e9;
	for (int rep = 0; rep < 4; rep++) {
		CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL((k_stream<NA, NW, B>), dim3(grid), dim3(256), 0, 0, a, n, n_tiles); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
	}
	const double gb = (double)n * 4 * (NA + NW) / 1e9;
	printf("%-10s %2d read + %d written arrays, grid %5d: %7.3f ms  %6.2f TB/s  %5.1f ps per element\n", tag, NA, NW, grid, best, gb / best, best * 1e9 / (double)n);
}
int main()
{
	const long n = 16l << 20;      // 16 M elements (64 MB per array): the seeds of a 60 Mb -sen bundle
	Arr a;
	for (int j = 0; j < 12; j++) { int *p; CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 1, n * 4)); a.in[j] = p; }
	for (int j = 0; j < 6; j++) { CK(hipMalloc(&a.out[j], n * 4)); }
	for (int grid : { 512, 2048 }) {
		run<1, 0, false>("striped", a, n, grid); run<1, 0, true>("blocked", a, n, grid);
		run<4, 2, false>("striped", a, n, grid); run<4, 2, true>("blocked", a, n, grid);
		run<10, 5, false>("striped", a, n, grid); run<10, 5, true>("blocked", a, n, grid);
		run<12, 0, false>("striped", a, n, grid); run<12, 0, true>("blocked", a, n, grid);
	}
	return 0;
}
