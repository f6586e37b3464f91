// tools/l2_boundary_probe.hip -- does a kernel boundary on one stream cost the kernels running beside it their L2 contents?
// Kernel A: every workgroup re-reads its own 64 KB slice (L2-resident, larger than the L1) `iters` times.  Measured alone, then while a second
// stream launches empty one-workgroup kernels back to back, then while the second stream runs ONE long empty-ish kernel (control).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/l2probe tools/l2_boundary_probe.hip && /tmp/l2probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <thread>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_reread(const uint4 *tab, int slice_vec, int iters, unsigned *out)
{
	const uint4 *p = tab + (size_t)blockIdx.x * slice_vec;
	unsigned acc = 0;
	for (int it = 0; it < iters; it++) {
		for (int i = threadIdx.x; i < slice_vec; i += 256) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
		asm volatile("" ::: "memory");
	}
	if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_empty(unsigned *out) { if (out && threadIdx.x == 1000) out[1] = 1; }
__global__ void k_spin(unsigned *out, long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) { } if (out && threadIdx.x == 1000) out[2] = 1; }

int main()
{
	const int wgs = 1024, slice = 64 << 10, slice_vec = slice / 16, iters = 400;
	uint4 *tab; unsigned *out;
	CK(hipMalloc(&tab, (size_t)wgs * slice)); CK(hipMemset(tab, 1, (size_t)wgs * slice)); CK(hipMalloc(&out, 64));
	hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto run_a = [&]() { CK(hipEventRecord(e0, s1)); hipLaunchKernelGGL(k_reread, dim3(wgs), dim3(256), 0, s1, tab, slice_vec, iters, out); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms; };
	run_a();
	const double gb = (double)wgs * slice * iters / 1e9;
	for (int rep = 0; rep < 3; rep++) { const float ms = run_a(); printf("A alone:                          %8.3f ms  %7.1f GB/s from L2\n", ms, gb / ms * 1e3); }
	for (int mode = 0; mode < 3; mode++) {
		std::atomic<bool> stop{false}; std::atomic<long> n{0};
		std::thread th([&] {
			CK(hipSetDevice(0));
			while (!stop.load()) {
				if (mode == 0) { for (int k = 0; k < 64; k++) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s2, out); n += 64; CK(hipStreamSynchronize(s2)); }
				else if (mode == 1) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s2, out, 100000000ll / 50); n += 1; CK(hipStreamSynchronize(s2)); }      // 20 ms per kernel (100 MHz clock)
				else { for (int k = 0; k < 64; k++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s2, out, 2000ll); n += 64; CK(hipStreamSynchronize(s2)); }      // 20 us kernels back to back
			}
		});
		std::this_thread::sleep_for(std::chrono::milliseconds(50));
		for (int rep = 0; rep < 3; rep++) { const long n0 = n.load(); const float ms = run_a(); printf("A beside %-24s %8.3f ms  %7.1f GB/s   (%ld kernels on the other stream meanwhile)\n", mode == 0 ? "empty kernels:" : mode == 1 ? "one long kernel:" : "20-us kernels:", ms, gb / ms * 1e3, n.load() - n0); }
		stop = true; th.join();
	}
	return 0;
}
