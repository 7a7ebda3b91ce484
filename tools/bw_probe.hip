// tools/bw_probe.hip -- practical HBM bandwidth of an MI355X for the traffic shape of the decode kernel:
// R bytes read + W bytes written per launch (uint4 per lane, fully coalesced), plus read-only and write-only.
// hipcc --offload-arch=gfx950 -O2 -o tools/bw_probe tools/bw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int MODE> // 0 copy, 1 read, 2 write, 3 copy nt load+store, 4 copy nt store, 5 write nt
__global__ __launch_bounds__(256) void k_bw(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n, uint32_t *sink)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * 256;
	uint4 acc = make_uint4(0, 0, 0, 0);
	for (; i < n; i += stride) {
		if (MODE == 0)
			dst[i] = src[i];
		else if (MODE == 3) {
			const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(src) + i);
			__builtin_nontemporal_store(v, reinterpret_cast<u4 *>(dst) + i);
		} else if (MODE == 4) {
			const u4 v = reinterpret_cast<const u4 *>(src)[i];
			__builtin_nontemporal_store(v, reinterpret_cast<u4 *>(dst) + i);
		} else if (MODE == 5) {
			const u4 v = {(uint32_t)i, 1, 2, 3};
			__builtin_nontemporal_store(v, reinterpret_cast<u4 *>(dst) + i);
		}
		else if (MODE == 1) {
			const uint4 v = src[i];
			acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
		} else
			dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
	}
	if (MODE == 1 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u)
		sink[0] = 1;
}

template <int MODE>
static int run(const char *name, const uint4 *src, uint4 *dst, size_t n, uint32_t *sink, int grid)
{
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	for (int w = 0; w < 2; ++w)
		hipLaunchKernelGGL(k_bw<MODE>, dim3(grid), dim3(256), 0, 0, src, dst, n, sink);
	CK(hipDeviceSynchronize());
	const int reps = 10;
	CK(hipEventRecord(e0));
	for (int r = 0; r < reps; ++r)
		hipLaunchKernelGGL(k_bw<MODE>, dim3(grid), dim3(256), 0, 0, src, dst, n, sink);
	CK(hipEventRecord(e1));
	CK(hipEventSynchronize(e1));
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	ms /= reps;
	const double bytes = (double)n * 16 * ((MODE == 0 || MODE == 3 || MODE == 4) ? 2 : 1);
	printf("%-12s grid %6d  %.3f ms  %.0f GB/s\n", name, grid, ms, bytes / ms / 1e6);
	return 0;
}

int main()
{
	const size_t bytes = 6400ull << 20; /* 6.4 GiB each way: the 1024-image launch moves 6.5 GB in + 6.4 GB out */
	const size_t n = bytes / 16;
	uint4 *src, *dst;
	uint32_t *sink;
	CK(hipMalloc(&src, bytes));
	CK(hipMalloc(&dst, bytes));
	CK(hipMalloc(&sink, 64));
	CK(hipMemset(src, 1, bytes));
	CK(hipMemset(dst, 2, bytes));
	const int grids[3] = {256 * 8, 256 * 32, 256 * 128};
	for (int g = 0; g < 3; ++g) {
		run<0>("copy (r+w)", src, dst, n, sink, grids[g]);
		run<1>("read only", src, dst, n, sink, grids[g]);
		run<2>("write only", src, dst, n, sink, grids[g]);
		run<3>("copy nt ld+st", src, dst, n, sink, grids[g]);
		run<4>("copy nt st", src, dst, n, sink, grids[g]);
		run<5>("write nt", src, dst, n, sink, grids[g]);
	}
	return 0;
}
