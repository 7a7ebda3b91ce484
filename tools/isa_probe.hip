// tools/isa_probe.hip -- bring-up probe (not product code): checks the semantics of gfx950
// instructions the kernels rely on and measures their issue rate, so design decisions rest on
// measurements.  Build: hipcc --offload-arch=gfx950 -O3 tools/isa_probe.hip -o tools/isa_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__global__ void k_sem(const int *in, uint32_t *out)
{
	int a = in[0], b = in[1], c = in[2], d = in[3];
	uint32_t r0 = 0xdeadbeefu, r1 = 0xdeadbeefu, r2;
	asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 17" : "+v"(r0) : "v"(a), "v"(b));
	asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 17 op_sel:[0,0,0,1]" : "+v"(r1) : "v"(c), "v"(d));
	r2 = 0xdeadbeefu;
	asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 17\n\tv_ashr_pk_u8_i32 %0, %3, %4, 17 op_sel:[0,0,0,1]" : "+v"(r2) : "v"(a), "v"(b), "v"(c), "v"(d));
	out[0] = r0;
	out[1] = r1;
	out[2] = r2;
	uint32_t s = 0xdeadbeefu;
	asm volatile("v_sat_pk_u8_i16 %0, %1" : "+v"(s) : "v"(in[4]));
	out[3] = s;
	// SDWA shift into the upper half, preserving the lower half
	uint32_t p = 0xdeadbeefu;
	asm volatile("v_ashrrev_i32 %0, 17, %1\n\tv_ashrrev_i32_sdwa %0, 17, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\ts_nop 0" : "+v"(p) : "v"(c), "v"(d));
	out[4] = p; // expect 004dfffd  (c>>17 = -3 -> fffd, d>>17 = 77 -> 004d)
	uint32_t q = 0xdeadbeefu;
	asm volatile("v_sat_pk_u8_i16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n\ts_nop 0" : "+v"(q) : "v"(in[4]));
	out[5] = q; // expect 00ffbeef
	uint32_t r3;
	asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r3) : "v"(in[4]), "v"(0x00020003), "v"(1000));
	out[6] = r3; // (300*3 + (-5)*2) + 1000 = 1890
	uint32_t al;
	int sh = in[5];
	asm volatile("v_alignbyte_b32 %0, %1, %2, %3" : "=v"(al) : "v"(0x44332211), "v"(0xddccbbaa), "v"(sh));
	out[7] = al; // sh=3: bytes (dd,11,22,33) -> 0x332211dd
}

template <int OP>
__global__ __launch_bounds__(256) void k_rate(uint32_t *out, uint32_t seed, int iters)
{
	uint32_t x0 = seed + threadIdx.x, x1 = x0 * 3u, x2 = x0 * 5u, x3 = x0 * 7u, x4 = x0 * 11u, x5 = x0 * 13u, x6 = x0 * 17u, x7 = x0 * 19u;
	const uint32_t k = seed | 0x10001u;
	for (int i = 0; i < iters; ++i) {
#define REP8(stmt) { uint32_t &x = x0; stmt } { uint32_t &x = x1; stmt } { uint32_t &x = x2; stmt } { uint32_t &x = x3; stmt } { uint32_t &x = x4; stmt } { uint32_t &x = x5; stmt } { uint32_t &x = x6; stmt } { uint32_t &x = x7; stmt }
		if (OP == 0) { REP8(asm volatile("v_dot2c_i32_i16 %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 1) { REP8(asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 2) { REP8(asm volatile("v_perm_b32 %0, %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 3) { REP8(asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(k));) }
		if (OP == 4) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(k));) }
		if (OP == 5) { REP8(asm volatile("v_ashr_pk_u8_i32 %0, %0, %1, 3" : "+v"(x) : "v"(k));) }
		if (OP == 6) { REP8(asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(x) : "v"(k));) }
		if (OP == 7) { REP8(asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(k));) }
		if (OP == 8) { REP8(asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 9) { REP8(asm volatile("v_med3_i32 %0, %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 10) { REP8(asm volatile("v_alignbyte_b32 %0, %0, %1, 3" : "+v"(x) : "v"(k));) }
		if (OP == 11) { REP8(asm volatile("v_pk_mad_u16 %0, %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 12) { REP8(asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(x));) }
		if (OP == 13) { REP8(asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 14) { REP8(asm volatile("v_dot2_i32_i16 %0, %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 15) { REP8(asm volatile("v_ashrrev_i32_sdwa %0, 3, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(x));) }
		if (OP == 16) { REP8(asm volatile("v_sat_pk_u8_i16 %0, %0" : "+v"(x));) }
		if (OP == 17) { REP8(asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(k));) }
		if (OP == 18) { REP8(asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(x) : "v"(k));) }
		if (OP == 19) { REP8(asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(k));) }
		if (OP == 20) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(k));) }
		if (OP == 21) { REP8(asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(k));) }
		if (OP == 22) { REP8(asm volatile("v_pk_ashrrev_i16 %0, 3, %0" : "+v"(x));) }
		if (OP == 23) { REP8(asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(x) : "v"(k));) }
		if (OP == 24) { REP8(asm volatile("v_dot2_i32_i16 %0, %0, %1, %0" : "+v"(x) : "s"(k));) }
		if (OP == 25) { REP8(asm volatile("v_dot2_i32_i16 %0, %0, %1, 0" : "+v"(x) : "s"(k));) }
		if (OP == 26) { REP8(asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(x) : "s"(k));) }
		if (OP == 27) { REP8(asm volatile("v_dot4_u32_u8 %0, %0, %1, 0" : "+v"(x) : "s"(k));) }
		if (OP == 28) { REP8(asm volatile("v_dot2_i32_i16 %0, %0, %1, 0" : "+v"(x) : "v"(k));) }
		if (OP == 29) { REP8(asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "s"(k));) }
		if (OP == 30) { REP8(asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "s"(k));) }
	}
	out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}

/* sustained shader clock under a VALU-bound load: s_memtime (core clock) against s_memrealtime (100 MHz) */
__global__ __launch_bounds__(256) void k_clock(unsigned long long *out, uint32_t *sink, uint32_t seed, int iters)
{
	uint32_t x0 = seed + threadIdx.x, x1 = x0 * 3u, x2 = x0 * 5u, x3 = x0 * 7u, x4 = x0 * 11u, x5 = x0 * 13u, x6 = x0 * 17u, x7 = x0 * 19u;
	const uint32_t k = seed | 0x10001u;
	const unsigned long long c0 = clock64(), w0 = wall_clock64();
	for (int i = 0; i < iters; ++i) {
		REP8(asm volatile("v_dot2_i32_i16 %0, %0, %1, %0" : "+v"(x) : "v"(k));)
	}
	const unsigned long long c1 = clock64(), w1 = wall_clock64();
	sink[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
	if (threadIdx.x == 0) {
		out[2 * blockIdx.x] = c1 - c0;
		out[2 * blockIdx.x + 1] = w1 - w0;
	}
}

static int clock_probe(uint32_t *d_out)
{
	const int blocks = 256 * 8;
	unsigned long long *d_t, *h_t = (unsigned long long *)malloc(sizeof(unsigned long long) * 2 * blocks);
	CK(hipMalloc(&d_t, sizeof(unsigned long long) * 2 * blocks));
	for (int round = 0; round < 4; ++round) {
		const int iters = 4096 << (2 * round > 6 ? 6 : 2 * round); /* 4096, 16384, 65536, 262144 */
		hipEvent_t e0, e1;
		CK(hipEventCreate(&e0));
		CK(hipEventCreate(&e1));
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(256), 0, 0, d_t, d_out, 12345u, iters);
		CK(hipEventRecord(e1));
		CK(hipEventSynchronize(e1));
		float ms = 0;
		CK(hipEventElapsedTime(&ms, e0, e1));
		CK(hipMemcpy(h_t, d_t, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
		double sc = 0, sw = 0;
		for (int i = 0; i < blocks; ++i) {
			sc += (double)h_t[2 * i];
			sw += (double)h_t[2 * i + 1];
		}
		const double ops = (double)blocks * 256 * iters * 8;
		printf("clock probe: %7d iters  %8.3f ms  %6.2f T lane-ops/s  s_memtime/s_memrealtime = %.3f  (x100 MHz = %.0f MHz if s_memtime is the core clock)\n", iters, ms,
				 ops / ms / 1e9, sc / sw, sc / sw * 100.0);
	}
	free(h_t);
	CK(hipFree(d_t));
	return 0;
}

template <int OP>
static int rate(const char *name, uint32_t *d_out)
{
	const int blocks = 256 * 8, iters = 4096;
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, 16);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0));
	hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u, iters);
	CK(hipEventRecord(e1));
	CK(hipEventSynchronize(e1));
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	double ops = (double)blocks * 256 * iters * 8;
	printf("%-22s %8.2f T lane-ops/s  (%.3f ms)\n", name, ops / ms / 1e9, ms);
	return 0;
}

int main()
{
	int *d_in;
	uint32_t *d_out;
	CK(hipMalloc(&d_in, 64));
	CK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
	// a>>17 = 5, b>>17 = 300 (sat 255), c>>17 = -3 (sat 0), d>>17 = 77
	int h_in[6] = {5 << 17, 300 << 17, -(3 << 17), 77 << 17, (int)(((uint32_t)(uint16_t)-5 << 16) | 300u), 3};
	CK(hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice));
	hipLaunchKernelGGL(k_sem, dim3(1), dim3(1), 0, 0, d_in, d_out);
	uint32_t h_out[8];
	CK(hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost));
	printf("ashr_pk_u8_i32 lo  (dst preset deadbeef): %08x  (expect ....ff05)\n", h_out[0]);
	printf("ashr_pk_u8_i32 hi  (dst preset deadbeef): %08x  (expect 4d00....)\n", h_out[1]);
	printf("ashr_pk_u8_i32 lo then hi               : %08x  (expect 4d00ff05)\n", h_out[2]);
	printf("sat_pk_u8_i16 of (300, -5)              : %08x  (expect ....00ff)\n", h_out[3]);
	printf("ashrrev + ashrrev_sdwa WORD_1 preserve  : %08x  (expect 004dfffd)\n", h_out[4]);
	printf("sat_pk_u8_i16_sdwa WORD_1 preserve      : %08x  (expect 00ffbeef)\n", h_out[5]);
	printf("v_dot2_i32_i16 VOP3P vgpr operands      : %u  (expect 1890)\n", h_out[6]);
	printf("v_alignbyte_b32 with VGPR shift 3       : %08x  (expect 332211dd)\n", h_out[7]);
	if (getenv("PROBE_CLOCK_ONLY")) {
		clock_probe(d_out);
		return 0;
	}
	rate<7>("v_add_u32", d_out);
	rate<0>("v_dot2c_i32_i16", d_out);
	rate<1>("v_dot4_u32_u8", d_out);
	rate<2>("v_perm_b32", d_out);
	rate<3>("v_pk_mul_lo_u16", d_out);
	rate<4>("v_mul_lo_u32", d_out);
	rate<5>("v_ashr_pk_u8_i32", d_out);
	rate<6>("v_cvt_pk_i16_i32", d_out);
	rate<8>("v_mad_i32_i24", d_out);
	rate<9>("v_med3_i32", d_out);
	rate<10>("v_alignbyte_b32", d_out);
	rate<11>("v_pk_mad_u16", d_out);
	rate<12>("v_ashrrev_i32", d_out);
	rate<13>("v_and_or_b32", d_out);
	rate<14>("v_dot2_i32_i16 (VOP3P)", d_out);
	rate<15>("v_ashrrev_i32_sdwa", d_out);
	rate<16>("v_sat_pk_u8_i16", d_out);
	rate<17>("v_mov_b32", d_out);
	rate<18>("v_pk_add_i16", d_out);
	rate<19>("v_and_b32", d_out);
	rate<20>("v_lshl_add_u32", d_out);
	rate<21>("v_sub_u32", d_out);
	rate<22>("v_pk_ashrrev_i16", d_out);
	rate<23>("v_add3_u32", d_out);
	rate<24>("v_dot2 v,v,S,v", d_out);
	rate<25>("v_dot2 v,v,S,0", d_out);
	rate<26>("v_perm v,v,v,S", d_out);
	rate<27>("v_dot4 v,v,S,0", d_out);
	rate<28>("v_dot2 v,v,v,0", d_out);
	rate<29>("v_add_u32 v,S,v", d_out);
	rate<30>("v_pk_mul_lo_u16 v,v,S", d_out);
	clock_probe(d_out);
	return 0;
}
