// tools/isa_probe4.hip -- bring-up probe (not product code), round 2: issue rates of the 4-byte (VOP1/VOP2)
// float and integer forms against their packed / VOP3 twins, to decide whether the encoder's packed-f32
// arithmetic and the decoders' VOP3 address arithmetic can be spelled cheaper.
// Build: hipcc --offload-arch=gfx950 -O3 tools/isa_probe4.hip -o tools/isa_probe4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

typedef uint32_t u2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k_rate(uint32_t *out, uint32_t seed, int iters)
{
	uint32_t x0 = seed + threadIdx.x, x1 = x0 * 3u, x2 = x0 * 5u, x3 = x0 * 7u, x4 = x0 * 11u, x5 = x0 * 13u, x6 = x0 * 17u, x7 = x0 * 19u;
	const uint32_t k = seed | 0x10001u;
	for (int i = 0; i < iters; ++i) {
#define REP8(stmt) { uint32_t &x = x0; stmt } { uint32_t &x = x1; stmt } { uint32_t &x = x2; stmt } { uint32_t &x = x3; stmt } { uint32_t &x = x4; stmt } { uint32_t &x = x5; stmt } { uint32_t &x = x6; stmt } { uint32_t &x = x7; stmt }
		if (OP == 0) { REP8(asm volatile("v_add_f32 %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 1) { REP8(asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 2) { REP8(asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 3) { REP8(asm volatile("v_fmac_f32 %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 4) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 5) { REP8(asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x));) }
		if (OP == 6) { REP8(asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x));) }
		if (OP == 7) { REP8(asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(x));) }
		if (OP == 8) { REP8(asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x));) }
		if (OP == 9) { REP8(asm volatile("v_or_b32 %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 10) { REP8(asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 11) { REP8(asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(x));) }
		if (OP == 12) { REP8(asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x) : "v"(k));) }
		if (OP == 13) { REP8(asm volatile("v_max_i32 %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 14) { REP8(asm volatile("v_mul_i32_i24 %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 15) { REP8(asm volatile("v_mul_i32_i24_sdwa %0, sext(%0), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(x) : "v"(k));) }
		if (OP == 16) { REP8(asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(x) : "v"(k));) }
		if (OP == 17) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(k));) }
		if (OP == 18) { REP8(asm volatile("v_bfi_b32 %0, %1, %0, %0" : "+v"(x) : "v"(k));) }
		if (OP == 19) { REP8(asm volatile("v_add_f32 %0, %1, %0" : "+v"(x) : "s"(k));) }
		if (OP == 20) { REP8(asm volatile("v_mul_f32 %0, 0x3fb504f3, %0" : "+v"(x));) }   /* 8-byte: literal constant */
		if (OP == 21) { REP8(asm volatile("v_add_f32_e64 %0, %1, %0" : "+v"(x) : "v"(k));) } /* VOP3 encoding of the same op */
		if (OP == 22) { REP8(asm volatile("v_bfrev_b32 %0, %0" : "+v"(x));) }
		if (OP == 23) { REP8(asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 24) { REP8(asm volatile("v_add_f32 %0, |%1|, %0" : "+v"(x) : "v"(k));) }   /* modifiers force VOP3 */
		if (OP == 25) { REP8(asm volatile("v_mul_legacy_f32 %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 26) { REP8(asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(x) : "v"(k) : "vcc");) }
		if (OP == 27) { REP8(asm volatile("v_min_f32 %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 28) { REP8(asm volatile("v_cvt_pk_u8_f32 %0, %0, 1, %0" : "+v"(x));) }
		if (OP == 29) { REP8(asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(x));) }
		if (OP == 30) { REP8(asm volatile("v_bitop3_b32 %0, %0, %1, %0 bitop3:0x96" : "+v"(x) : "v"(k));) }
		if (OP == 31) { REP8(asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 32) { REP8(asm volatile("v_sub_u32 %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 33) { REP8(asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 34) { REP8(asm volatile("v_ashrrev_i32 %0, 10, %0" : "+v"(x));) }
		if (OP == 35) { REP8(asm volatile("v_add_f32 %0, 0.5, %0" : "+v"(x));) }             /* inline constant: 4-byte */
	}
	out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}

/* packed f32: 8 independent register pairs */
template <int OP>
__global__ __launch_bounds__(256) void k_rate_pk(uint32_t *out, uint32_t seed, int iters)
{
	u2 x0 = {seed + threadIdx.x, seed}, x1 = x0 * 3u, x2 = x0 * 5u, x3 = x0 * 7u, x4 = x0 * 11u, x5 = x0 * 13u, x6 = x0 * 17u, x7 = x0 * 19u;
	const u2 k = {seed | 0x10001u, seed | 0x30003u};
	for (int i = 0; i < iters; ++i) {
#define REP8P(stmt) { u2 &x = x0; stmt } { u2 &x = x1; stmt } { u2 &x = x2; stmt } { u2 &x = x3; stmt } { u2 &x = x4; stmt } { u2 &x = x5; stmt } { u2 &x = x6; stmt } { u2 &x = x7; stmt }
		if (OP == 0) { REP8P(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(k));) }
		if (OP == 1) { REP8P(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(k));) }
		if (OP == 2) { REP8P(asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 3) { REP8P(asm volatile("v_pk_mov_b32 %0, %0, %1" : "+v"(x) : "v"(k));) }
		if (OP == 4) { REP8P(asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(x));) }
	}
	out[blockIdx.x * 256 + threadIdx.x] = x0.x ^ x1.x ^ x2.x ^ x3.x ^ x4.x ^ x5.x ^ x6.x ^ x7.x ^ x0.y ^ x1.y ^ x2.y ^ x3.y ^ x4.y ^ x5.y ^ x6.y ^ x7.y;
}

/* how many waves per SIMD a 4-byte op needs to reach its rate: occupancy limited by LDS per workgroup */
template <int OP>
__global__ __launch_bounds__(256) void k_rate_occ(uint32_t *out, uint32_t seed, int iters)
{
	extern __shared__ uint32_t pad[];
	uint32_t x0 = seed + threadIdx.x, x1 = x0 * 3u, x2 = x0 * 5u, x3 = x0 * 7u, x4 = x0 * 11u, x5 = x0 * 13u, x6 = x0 * 17u, x7 = x0 * 19u;
	const uint32_t k = seed | 0x10001u;
	if (seed == 1) pad[threadIdx.x] = x0;
	for (int i = 0; i < iters; ++i) {
		if (OP == 0) { REP8(asm volatile("v_add_f32 %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 1) { REP8(asm volatile("v_dot2_i32_i16 %0, %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 2) { REP8(asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(k));) }
	}
	out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}

template <typename K>
static int run(const char *name, K kern, uint32_t *d_out, double ops_per_instr, size_t lds = 0, int blocks = 256 * 8)
{
	const int iters = 4096;
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	for (int w = 0; w < 3; ++w)
		hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_out, 12345u, iters);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0));
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, d_out, 12345u, iters);
	CK(hipEventRecord(e1));
	CK(hipEventSynchronize(e1));
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	double instr = (double)blocks * 256 * iters * 8;
	printf("%-34s %8.2f T lane-instr/s  %8.2f T lane-ops/s  (%.3f ms)\n", name, instr / ms / 1e9, instr * ops_per_instr / ms / 1e9, ms);
	return 0;
}

int main()
{
	uint32_t *d_out;
	CK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
#define R(op, name) run(name, k_rate<op>, d_out, 1.0)
	R(17 + 0 * 0, "v_cndmask_b32 (vop2)");
	R(0, "v_add_f32 (vop2)");
	R(1, "v_mul_f32 (vop2)");
	R(2, "v_sub_f32 (vop2)");
	R(3, "v_fmac_f32 (vop2)");
	R(4, "v_fma_f32 (vop3)");
	R(35, "v_add_f32 inline const (vop2)");
	R(19, "v_add_f32 v,S,v (vop2)");
	R(20, "v_mul_f32 literal (vop2+lit, 8B)");
	R(21, "v_add_f32_e64 (vop3)");
	R(24, "v_add_f32 |src| (vop3)");
	R(25, "v_mul_legacy_f32 (vop2)");
	R(27, "v_min_f32 (vop2)");
	R(5, "v_cvt_i32_f32 (vop1)");
	R(6, "v_cvt_f32_i32 (vop1)");
	R(7, "v_cvt_f32_ubyte0 (vop1)");
	R(28, "v_cvt_pk_u8_f32 (vop3)");
	R(8, "v_lshlrev_b32 (vop2)");
	R(29, "v_lshrrev_b32 (vop2)");
	R(34, "v_ashrrev_i32 10 (vop2)");
	R(9, "v_or_b32 (vop2)");
	R(10, "v_xor_b32 (vop2)");
	R(23, "v_add_u32 (vop2)");
	R(32, "v_sub_u32 (vop2)");
	R(33, "v_subrev_u32 (vop2)");
	R(26, "v_add_co_u32 (vop2)");
	R(13, "v_max_i32 (vop2)");
	R(14, "v_mul_i32_i24 (vop2)");
	R(31, "v_mul_u32_u24 (vop2)");
	R(15, "v_mul_i32_i24_sdwa");
	R(16, "v_add_u32_sdwa");
	R(11, "v_bfe_u32 (vop3)");
	R(12, "v_lshl_or_b32 (vop3)");
	R(18, "v_bfi_b32 (vop3)");
	R(30, "v_bitop3_b32 (vop3)");
	R(22, "v_bfrev_b32 (vop1)");
	run("v_pk_add_f32 (2 ops)", k_rate_pk<0>, d_out, 2.0);
	run("v_pk_mul_f32 (2 ops)", k_rate_pk<1>, d_out, 2.0);
	run("v_pk_fma_f32 (2 ops)", k_rate_pk<2>, d_out, 2.0);
	run("v_pk_mov_b32 (2 ops)", k_rate_pk<3>, d_out, 2.0);
	run("v_lshlrev_b64", k_rate_pk<4>, d_out, 1.0);
	/* occupancy: 1, 2, 3, 4, 8 workgroups (x4 waves) per CU via the LDS request */
	const size_t lds_for[5] = {160 * 1024, 80 * 1024, 53 * 1024, 40 * 1024, 16 * 1024};
	const char *occ[5] = {"1 wave/SIMD", "2 waves/SIMD", "3 waves/SIMD", "4 waves/SIMD", "8 waves/SIMD"};
	for (int o = 0; o < 5; ++o) {
		char nm[64];
		hipFuncSetAttribute((const void *)k_rate_occ<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		hipFuncSetAttribute((const void *)k_rate_occ<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		hipFuncSetAttribute((const void *)k_rate_occ<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		snprintf(nm, sizeof nm, "v_add_f32, %s", occ[o]);
		run(nm, k_rate_occ<0>, d_out, 1.0, lds_for[o]);
		snprintf(nm, sizeof nm, "v_dot2_i32_i16, %s", occ[o]);
		run(nm, k_rate_occ<1>, d_out, 1.0, lds_for[o]);
		snprintf(nm, sizeof nm, "v_mov_b32, %s", occ[o]);
		run(nm, k_rate_occ<2>, d_out, 1.0, lds_for[o]);
	}
	return 0;
}
