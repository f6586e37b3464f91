// tools/dispatch_probe.hip -- how fast does an MI355X start workgroups?  One-wave workgroups holding L bytes of LDS and R registers that do
// nothing (or spin for a given time): is a launch of 25 000 of them bound by the dispatcher?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dispatch_probe tools/dispatch_probe.hip && /tmp/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int LDS_BYTES>
__global__ void __launch_bounds__(64) k_empty(uint32_t *sink, int spin_us)
{
	__shared__ uint32_t buf[LDS_BYTES / 4];
	buf[threadIdx.x] = threadIdx.x;
	if (spin_us > 0) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < (unsigned long long)spin_us * 100) __builtin_amdgcn_s_sleep(8); }
	if (buf[(threadIdx.x + 1) & 63] == 0xdeadbeefu) sink[0] = 1;
}

template <int LDS_BYTES>
static void run(uint32_t *sink, int wgs, int spin_us)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k_empty<LDS_BYTES>, dim3(wgs), dim3(64), 0, 0, sink, 0);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k_empty<LDS_BYTES>, dim3(wgs), dim3(64), 0, 0, sink, spin_us);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	printf("  %6d one-wave workgroups, %5d B of LDS each, %3d us of sleep each: %8.3f ms = %6.2f workgroups per us\n", wgs, LDS_BYTES, spin_us, ms, wgs / ms / 1e3);
}

int main()
{
	uint32_t *sink; hipMalloc(&sink, 64);
	for (int wgs : {25000, 100000}) {
		run<256>(sink, wgs, 0); run<15616>(sink, wgs, 0); run<32768>(sink, wgs, 0);
		run<15616>(sink, wgs, 30); run<15616>(sink, wgs, 100); run<15616>(sink, wgs, 250);
	}
	return 0;
}
