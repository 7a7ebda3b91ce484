// tools/bw_probe2.hip -- second look at the copy ceiling of an MI355X (tools/bw_probe reached only
// 4.7-5.4 TB/s where /opt/skills/guides/MI355X_MICROARCH.md records 6.29 TB/s for a float4 copy).
// What differs from bw_probe.hip: U independent 16-byte loads in flight per lane before the first store, a
// persistent grid (k workgroups per CU, grid-stride), a warm-up that lasts >= 60 ms before every timed run
// (the clock ramp after an idle gap takes ~25 ms), 20 timed launches, read:write ratios
// of the decode kernels (1:1 int16 planes, 0.54:1 byte planes) besides the plain copy.
// hipcc --offload-arch=gfx950 -O3 -o tools/bw_probe2 tools/bw_probe2.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// MODE 0 plain, 1 nt loads + nt stores.  RW: 0 copy n -> n, 1 read only, 2 write only, 3 read n/2 + write n (byte-plane shape)
template <int U, int MODE, int RW>
__global__ __launch_bounds__(256) void k_copy(const u4 *__restrict__ src, u4 *__restrict__ dst, size_t n, uint32_t *sink)
{
	const size_t stride = (size_t)gridDim.x * 256;
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	u4 acc = {0, 0, 0, 0};
	for (; i + (U - 1) * stride < n; i += U * stride) {
		u4 v[U];
		if (RW != 2) {
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const size_t j = RW == 3 ? ((i + u * stride) >> 1) : (i + u * stride);
				if (RW == 3 && (u & 1))
					v[u] = v[u - 1];
				else
					v[u] = MODE ? __builtin_nontemporal_load(src + j) : src[j];
			}
		} else {
#pragma unroll
			for (int u = 0; u < U; ++u)
				v[u] = (u4){(uint32_t)i, 1, 2, (uint32_t)u};
		}
		if (RW == 1) {
#pragma unroll
			for (int u = 0; u < U; ++u)
				acc ^= v[u];
		} else {
#pragma unroll
			for (int u = 0; u < U; ++u) {
				if (MODE)
					__builtin_nontemporal_store(v[u], dst + i + u * stride);
				else
					dst[i + u * stride] = v[u];
			}
		}
	}
	if (RW == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u)
		sink[0] = 1;
}

template <int U, int MODE, int RW>
static int run(const u4 *src, u4 *dst, size_t n, uint32_t *sink, int grid)
{
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	const auto t0 = std::chrono::steady_clock::now();
	int warm = 0;
	do {
		for (int w = 0; w < 4; ++w, ++warm)
			hipLaunchKernelGGL((k_copy<U, MODE, RW>), dim3(grid), dim3(256), 0, 0, src, dst, n, sink);
		CK(hipDeviceSynchronize());
	} while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.06);
	const int reps = 20;
	CK(hipEventRecord(e0));
	for (int r = 0; r < reps; ++r)
		hipLaunchKernelGGL((k_copy<U, MODE, RW>), dim3(grid), dim3(256), 0, 0, src, dst, n, sink);
	CK(hipEventRecord(e1));
	CK(hipEventSynchronize(e1));
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	ms /= reps;
	const double bytes = (double)n * 16 * (RW == 0 ? 2.0 : (RW == 3 ? 1.5 : 1.0));
	static const char *rw[4] = {"copy 1:1", "read", "write", "copy 0.5:1"};
	printf("%-10s %-5s U=%d grid %6d (%4.1f wg/CU)  %.3f ms  %6.0f GB/s  (%d warm-up launches)\n", rw[RW], MODE ? "nt" : "plain", U, grid, grid / 256.0, ms,
			 bytes / ms / 1e6, warm);
	fflush(stdout);
	CK(hipEventDestroy(e0));
	CK(hipEventDestroy(e1));
	return 0;
}

int main()
{
	const size_t bytes = 6400ull << 20; /* 6.4 GiB each way, as the 1024-image decode launch */
	const size_t n = bytes / 16;
	u4 *src, *dst;
	uint32_t *sink;
	CK(hipMalloc(&src, bytes));
	CK(hipMalloc(&dst, bytes));
	CK(hipMalloc(&sink, 64));
	CK(hipMemset(src, 1, bytes));
	CK(hipMemset(dst, 2, bytes));
	CK(hipDeviceSynchronize());
	const int grids[4] = {256 * 4, 256 * 8, 256 * 16, 256 * 64};
	for (int g = 0; g < 4; ++g) {
		run<1, 0, 0>(src, dst, n, sink, grids[g]);
		run<4, 0, 0>(src, dst, n, sink, grids[g]);
		run<8, 0, 0>(src, dst, n, sink, grids[g]);
		run<4, 1, 0>(src, dst, n, sink, grids[g]);
		run<8, 1, 0>(src, dst, n, sink, grids[g]);
	}
	for (int g = 1; g < 3; ++g) {
		run<8, 0, 1>(src, dst, n, sink, grids[g]);
		run<8, 1, 1>(src, dst, n, sink, grids[g]);
		run<8, 0, 2>(src, dst, n, sink, grids[g]);
		run<8, 1, 2>(src, dst, n, sink, grids[g]);
		run<8, 0, 3>(src, dst, n, sink, grids[g]);
		run<8, 1, 3>(src, dst, n, sink, grids[g]);
	}
	// hipMemcpyAsync device-to-device of the same bytes, for reference
	{
		hipEvent_t e0, e1;
		CK(hipEventCreate(&e0));
		CK(hipEventCreate(&e1));
		for (int w = 0; w < 12; ++w)
			CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0));
		CK(hipDeviceSynchronize());
		CK(hipEventRecord(e0));
		for (int r = 0; r < 10; ++r)
			CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0));
		CK(hipEventRecord(e1));
		CK(hipEventSynchronize(e1));
		float ms = 0;
		CK(hipEventElapsedTime(&ms, e0, e1));
		printf("hipMemcpyAsync D2D  %.3f ms  %6.0f GB/s\n", ms / 10, 2.0 * bytes / (ms / 10) / 1e6);
	}
	return 0;
}
