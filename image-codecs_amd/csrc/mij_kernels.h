/*
 * mij_kernels.h -- gfx950 (CDNA4, wave64) device code of the JPEG back end.
 *
 * Everything the reference does after the Huffman walk, as integer HIP kernels:
 *   de-quantisation        codec/jpeg.c:325,345,365 / :1319-1324     (int16 wrap, v_pk_mul_lo_u16)
 *   8x8 integer IDCT       codec/jpeg.c:578-679                      (v_dot2_i32_i16 form, see idct_*)
 *   chroma upsampling      codec/jpeg.c:1765-1840,1962-1971, rows :2301-2319
 *   YCbCr->RGB + the other colour branches  codec/jpeg.c:1976-2018, :2320-2431
 *
 * Two families:
 *   k_fused420   one pass, coefficients -> interleaved pixels, for 3-component h2v2 YCbCr
 *                (the north-star configuration).  A 256-thread workgroup owns a band of MCU rows
 *                of one image and marches down it; per MCU row: phase A = one 8x8 block per lane
 *                (coefficients straight from HBM in the tile layout of mij.h, fully coalesced),
 *                IDCT result to LDS planes; phase B = 4-pixel strips x row pairs, h2v2 filter as
 *                one v_dot4_u32_u8 per sample, colour as v_dot2_i32_i16, 12-byte stores that are
 *                contiguous across the wave.
 *   k_idct_planes + k_resample_color   the general two-pass path for every other sampling /
 *                colour layout the reference accepts (planar u8 intermediate in HBM).
 *
 * No MFMA (the butterflies are not a dense contraction), no CUDA-compat headers, no fallbacks.
 */
#ifndef MIJ_KERNELS_H
#define MIJ_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mij.h"

#ifndef MIJ_VARIANT
#define MIJ_VARIANT 0
#endif

namespace mij {

/* ------------------------------------------------------------------ device-side descriptors */

struct DevComp {
	int32_t h, v, x, y;   /* sampling factors, effective size in samples */
	int32_t bw, bh;       /* padded size in blocks */
	int32_t hs, vs;       /* h_max / h, v_max / v (integer division, codec/jpeg.c:2273-2274) */
	uint64_t coef_off;    /* byte offset of the component's tile-layout plane in the coefficient arena */
	uint64_t plane_off;   /* byte offset of its u8 sample plane in the scratch arena (two-pass path) */
	uint64_t dc_off;      /* compact planes (MIJ_DEV_COEF_BYTES): offset of the int16 DC array, one entry per block */
	uint64_t hi_off;      /* compact planes: offset of the escape bytes, 64 per block in in-block position order P */
};

/* DevImage.flags: the image's coefficients sit in HBM as COMPACT planes (mij.h, "compact coefficient planes"):
 * per 64-block tile 4 KiB of low bytes in the chunk order of the int16 tile layout, the DC terms in an int16
 * array, and for blocks holding a coefficient outside -128..127 ("escaped", bit 0 of the block's byte at
 * in-block position 0) 64 high bytes h with  coefficient == sext8(low) + 256*h  (mod 2^16). */
#define MIJ_DEV_COEF_BYTES 0x100

struct DevImage {
	int32_t width, height, n_out, color;
	int32_t ncomp, flags, mcu_x, mcu_y;
	uint64_t out_off; /* byte offset of the pixels in the output arena */
	uint64_t src16_off; /* k_pack_c8 only: byte offset of the image's int16 tile-layout planes (back to back) in the upload scratch */
	uint32_t es_blk_off; /* GPU entropy stage: index of the image's first block in the per-block arrays (scan order) */
	uint8_t es_bpm, es_j0[4], es_pad[3]; /* blocks per MCU; first block-in-MCU of every component */
	uint64_t plane_bytes_total;
	DevComp comp[4];
	uint32_t dq[4][32]; /* per component: quantisation table as u16 pairs in in-block position order P */
};

struct WorkBand { /* one workgroup of the band kernels: MCU rows [m0, m1) of an image */
	uint32_t img, m0, m1;
	uint32_t cols; /* column-segmented forms (k_fused420c / k_fused440c) only: first MCU column | one past the last << 16 */
};
struct WorkIdct { /* one workgroup of k_idct_planes: 256 consecutive blocks of one component */
	uint32_t img, comp, first, pad;
};

/* ------------------------------------------------------------------ small helpers */

typedef short v2s __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef uint32_t u3v __attribute__((ext_vector_type(3)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ constexpr uint32_t pk16(int lo, int hi) { return (uint32_t)(uint16_t)lo | ((uint32_t)(uint16_t)hi << 16); }

/*
 * Measured on MI355X (profiles/r01_isa_probe*.txt): every 8-byte VALU encoding (VOP3, VOP3P, SDWA:
 * v_dot2, v_dot4, v_perm, v_pk_mul, v_and_or ...) issues at one rate, 4-byte VOP1/VOP2 forms a
 * bit faster, v_ashr_pk_u8_i32 at half rate -- so the kernels are tuned for instruction COUNT.
 */

/* a.lo*b.lo + a.hi*b.hi + acc in wrapping 32-bit: the three-address VOP3P form of v_dot2_i32_i16
 * (the compiler only picks the tied v_dot2c form, which costs a v_mov per accumulator seed).
 * The constant operand lives in a VGPR (see vreg). */
/* MIJ_KSGPR: the constant operand (matrix entries, weights, selectors) is a scalar register the compiler materialises where
 * it needs it (s_mov is not a VALU instruction) instead of one of ~30 vector registers pinned for the whole kernel; one
 * scalar source per VOP3 instruction, so accumulator constants stay in vector registers.  A/B knob, see DESIGN.md 3.1. */
#ifndef MIJ_KSGPR
#define MIJ_KSGPR 0
#endif
#if MIJ_KSGPR
#define MIJ_KC "s"
#else
#define MIJ_KC "v"
#endif
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int acc)
{
	int d;
	asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), MIJ_KC(b), "v"(acc));
	return d;
}
__device__ __forceinline__ int dot2z(uint32_t a, uint32_t b)
{
	int d;
	asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), MIJ_KC(b));
	return d;
}
/* a loop-invariant constant pinned in a vector register */
__device__ __forceinline__ uint32_t vreg(uint32_t c)
{
	asm("" : "+v"(c));
	return c;
}
/* the constant operand of a v_dot2 / v_dot4 / v_perm / v_and_or (not an accumulator) */
__device__ __forceinline__ uint32_t kconst(uint32_t c)
{
#if MIJ_KSGPR
	return c;
#else
	return vreg(c);
#endif
}
/* sum of four u8*u8 products + acc (v_dot4_u32_u8) */
__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_udot4(a, b, acc, false); }
/* two int16 wrapping multiplies (v_pk_mul_lo_u16): the reference's (short)(coef * dequant) */
__device__ __forceinline__ uint32_t pkmul(uint32_t a, uint32_t b)
{
	return __builtin_bit_cast(uint32_t, (v2u)(__builtin_bit_cast(v2u, a) * __builtin_bit_cast(v2u, b)));
}
/* (x & m) | o in one instruction */
__device__ __forceinline__ uint32_t and_or(uint32_t x, uint32_t m, uint32_t o)
{
	uint32_t d;
	asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(m), MIJ_KC(o));
	return d;
}

__device__ __forceinline__ int clamp255(int x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }

/*
 * v_ashr_pk_u8_i32 (new on gfx950): sat_u8(a >> S) | sat_u8(b >> S) << 8 into ONE half of the
 * destination (op_sel picks the half, the other half is preserved -- measured).  ROCm 7.2's own
 * pattern for it assumes the other half is zeroed and mis-compiles "clamp(x>>n) | clamp(y>>n) << 8
 * | ..." chains, so it is only ever emitted here, through asm.  Two of them give four clamped
 * samples in a dword: the reference's ">> 17, stbi__clamp" (codec/jpeg.c:670-677) and
 * ">> 20, clamp" (:1988-2011).
 */
/* No trailing wait state: every consumer of a sat4 result in this file is a memory instruction
 * (ds_write / global_store); the gfx940-family dst_sel forwarding hazard only concerns a VALU that
 * reads the half-written register in the very next issue slot. */
template <int S>
__device__ __forceinline__ uint32_t sat4(int a, int b, int c, int d)
{
	uint32_t r;
	asm("v_ashr_pk_u8_i32 %0, %1, %2, %5\n\t"
		 "v_ashr_pk_u8_i32 %0, %3, %4, %5 op_sel:[0,0,0,1]"
		 : "=&v"(r)
		 : "v"(a), "v"(b), "v"(c), "v"(d), "n"(S));
	return r;
}
/* eight int16 pairs ((lo[i] >> S) | (hi[i] >> S) << 16): a plain shift, then an SDWA shift into the
 * upper half.  The results feed v_dot2 (VALU), so one wait state closes the block (dst_sel hazard). */
template <int S>
__device__ __forceinline__ void shr_pack8_i16(const int (&lo)[8], const int (&hi)[8], uint32_t (&o)[8])
{
#define MIJ_SP(i, l, h) "v_ashrrev_i32 %" #i ", %24, %" #l "\n\tv_ashrrev_i32_sdwa %" #i ", %24, %" #h " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
	asm(MIJ_SP(0, 8, 16) MIJ_SP(1, 9, 17) MIJ_SP(2, 10, 18) MIJ_SP(3, 11, 19) MIJ_SP(4, 12, 20) MIJ_SP(5, 13, 21) MIJ_SP(6, 14, 22) MIJ_SP(7, 15, 23) "s_nop 0"
		 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
		 : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(lo[4]), "v"(lo[5]), "v"(lo[6]), "v"(lo[7]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]),
			"v"(hi[4]), "v"(hi[5]), "v"(hi[6]), "v"(hi[7]), "n"(S));
#undef MIJ_SP
}
/* keeps the compiler from fusing a preceding shift with a following clamp into the v_ashr_pk pattern */
__device__ __forceinline__ int opaque(int x)
{
	asm("" : "+v"(x));
	return x;
}

/* ------------------------------------------------------------------ IDCT (codec/jpeg.c:578-679)
 *
 * STBI__IDCT_1D is linear over the ring Z/2^32, so its eight outputs are fixed integer
 * combinations of the inputs (derived from the macro, verified exhaustively in tests):
 *     x0 = 4096 s0 + 5352 s2 + 4096 s4 + 2217 s6          t3 = 5683 s1 + 4816 s3 + 3219 s5 + 1131 s7
 *     x1 = 4096 s0 + 2217 s2 - 4096 s4 - 5350 s6          t2 = 4816 s1 - 1129 s3 - 5681 s5 - 3218 s7
 *     x2 = 4096 s0 - 2217 s2 - 4096 s4 + 5350 s6          t1 = 3219 s1 - 5681 s3 + 1132 s5 + 4816 s7
 *     x3 = 4096 s0 - 5352 s2 + 4096 s4 - 2217 s6          t0 = 1131 s1 - 3218 s3 + 4816 s5 - 5680 s7
 * out = (x0+t3, x1+t2, x2+t1, x3+t0, x3-t0, x2-t1, x1-t2, x0-t3) + bias; the callers shift (>>10, >>17).
 * With the inputs paired as (s0,s4) (s2,s6) (s1,s3) (s5,s7) -- exactly how the tile layout stores
 * a column -- that is 14 v_dot2 + 8 add/sub per 1-D transform.
 */
struct Idct1D {
	int o[8];
};

/* the matrix entries as int16 pairs, pinned in vector registers once per kernel */
struct IdctK {
	uint32_t e0, e1, x0, x3, x1, x2, t3a, t3b, t2a, t2b, t1a, t1b, t0a, t0b;
	int bias1, bias2;
	__device__ __forceinline__ void init()
	{
		e0 = kconst(pk16(4096, 4096));
		e1 = kconst(pk16(4096, -4096));
		x0 = kconst(pk16(5352, 2217));
		x3 = kconst(pk16(-5352, -2217));
		x1 = kconst(pk16(2217, -5350));
		x2 = kconst(pk16(-2217, 5350));
		t3a = kconst(pk16(5683, 4816));
		t3b = kconst(pk16(3219, 1131));
		t2a = kconst(pk16(4816, -1129));
		t2b = kconst(pk16(-5681, -3218));
		t1a = kconst(pk16(3219, -5681));
		t1b = kconst(pk16(1132, 4816));
		t0a = kconst(pk16(1131, -3218));
		t0b = kconst(pk16(4816, -5680));
		bias1 = (int)vreg(512);                       /* codec/jpeg.c:639 */
		bias2 = (int)vreg(65536 + (128 << 17));       /* codec/jpeg.c:664 */
	}
};

__device__ __forceinline__ Idct1D idct1d_packed(const IdctK &K, int bias, uint32_t d04, uint32_t d26, uint32_t d13, uint32_t d57)
{
	int e0 = dot2(d04, K.e0, bias);
	int e1 = dot2(d04, K.e1, bias);
	uint32_t x0 = (uint32_t)dot2(d26, K.x0, e0);
	uint32_t x3 = (uint32_t)dot2(d26, K.x3, e0);
	uint32_t x1 = (uint32_t)dot2(d26, K.x1, e1);
	uint32_t x2 = (uint32_t)dot2(d26, K.x2, e1);
	uint32_t t3 = (uint32_t)dot2(d13, K.t3a, dot2z(d57, K.t3b));
	uint32_t t2 = (uint32_t)dot2(d13, K.t2a, dot2z(d57, K.t2b));
	uint32_t t1 = (uint32_t)dot2(d13, K.t1a, dot2z(d57, K.t1b));
	uint32_t t0 = (uint32_t)dot2(d13, K.t0a, dot2z(d57, K.t0b));
	Idct1D r;
	r.o[0] = (int)(x0 + t3);
	r.o[7] = (int)(x0 - t3);
	r.o[1] = (int)(x1 + t2);
	r.o[6] = (int)(x1 - t2);
	r.o[2] = (int)(x2 + t1);
	r.o[5] = (int)(x2 - t1);
	r.o[3] = (int)(x3 + t0);
	r.o[4] = (int)(x3 - t0);
	return r;
}

/* the same transform on full 32-bit inputs, all arithmetic wrapping (unsigned) like the
 * reference's int math on out-of-range streams */
template <int BIAS>
__device__ __forceinline__ Idct1D idct1d_wide(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7)
{
	uint32_t u0 = s0, u1 = s1, u2 = s2, u3 = s3, u4 = s4, u5 = s5, u6 = s6, u7 = s7;
	uint32_t x0 = 4096u * u0 + 5352u * u2 + 4096u * u4 + 2217u * u6 + (uint32_t)BIAS;
	uint32_t x1 = 4096u * u0 + 2217u * u2 - 4096u * u4 - 5350u * u6 + (uint32_t)BIAS;
	uint32_t x2 = 4096u * u0 - 2217u * u2 - 4096u * u4 + 5350u * u6 + (uint32_t)BIAS;
	uint32_t x3 = 4096u * u0 - 5352u * u2 + 4096u * u4 - 2217u * u6 + (uint32_t)BIAS;
	uint32_t t3 = 5683u * u1 + 4816u * u3 + 3219u * u5 + 1131u * u7;
	uint32_t t2 = 4816u * u1 - 1129u * u3 - 5681u * u5 - 3218u * u7;
	uint32_t t1 = 3219u * u1 - 5681u * u3 + 1132u * u5 + 4816u * u7;
	uint32_t t0 = 1131u * u1 - 3218u * u3 + 4816u * u5 - 5680u * u7;
	Idct1D r;
	r.o[0] = (int)(x0 + t3);
	r.o[7] = (int)(x0 - t3);
	r.o[1] = (int)(x1 + t2);
	r.o[6] = (int)(x1 - t2);
	r.o[2] = (int)(x2 + t1);
	r.o[5] = (int)(x2 - t1);
	r.o[3] = (int)(x3 + t0);
	r.o[4] = (int)(x3 - t0);
	return r;
}

#define MIJ_PASS2_BIAS (65536 + (128 << 17)) /* codec/jpeg.c:664 */

/* second-pass sums -> eight clamped samples: (x >> 17) saturated to 0..255 (codec/jpeg.c:670-677) */
__device__ __forceinline__ void pack_row(const Idct1D &r, uint32_t &lo, uint32_t &hi)
{
	lo = sat4<17>(r.o[0], r.o[1], r.o[2], r.o[3]);
	hi = sat4<17>(r.o[4], r.o[5], r.o[6], r.o[7]);
}

/*
 * One 8x8 block per lane.  c[k] = column k of the quantised block as stored in the tile layout
 * (.x=(r0,r4) .y=(r2,r6) .z=(r1,r3) .w=(r5,r7)); dq = the component's table in the same order
 * (32 dwords, wave-uniform -> scalar registers).  rows[i] = the 8 output samples of row i.
 * WIDE = false requires every first-pass output to fit int16 (host guarantee, mij.h).
 */
/* PREDEQ: the inputs are de-quantised already (byte-coefficient planes, load_block_b8) */
template <bool WIDE, bool PREDEQ = false>
__device__ __forceinline__ void idct_block(const IdctK &K, const uint4 (&c)[8], const uint32_t *__restrict__ dq, uint2 (&rows)[8])
{
	auto deq = [&](uint32_t v, uint32_t q) { return PREDEQ ? v : pkmul(v, q); };
	if constexpr (!WIDE) {
		/* pk[i][g]: row i, column pair g = (0,4) (2,6) (1,3) (5,7) */
		uint32_t pk[8][4];
		const int ca[4] = {0, 2, 1, 5}, cb[4] = {4, 6, 3, 7};
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			const int a = ca[g], b = cb[g];
			Idct1D va = idct1d_packed(K, K.bias1, deq(c[a].x, dq[4 * a + 0]), deq(c[a].y, dq[4 * a + 1]), deq(c[a].z, dq[4 * a + 2]),
											  deq(c[a].w, dq[4 * a + 3]));
			Idct1D vb = idct1d_packed(K, K.bias1, deq(c[b].x, dq[4 * b + 0]), deq(c[b].y, dq[4 * b + 1]), deq(c[b].z, dq[4 * b + 2]),
											  deq(c[b].w, dq[4 * b + 3]));
			uint32_t pg[8];
			shr_pack8_i16<10>(va.o, vb.o, pg);
#pragma unroll
			for (int i = 0; i < 8; ++i)
				pk[i][g] = pg[i];
		}
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			Idct1D r = idct1d_packed(K, K.bias2, pk[i][0], pk[i][1], pk[i][2], pk[i][3]);
			pack_row(r, rows[i].x, rows[i].y);
		}
	} else {
		int v[8][8]; /* v[row][col] */
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			Idct1D col = idct1d_packed(K, K.bias1, deq(c[k].x, dq[4 * k + 0]), deq(c[k].y, dq[4 * k + 1]), deq(c[k].z, dq[4 * k + 2]),
												deq(c[k].w, dq[4 * k + 3]));
#pragma unroll
			for (int i = 0; i < 8; ++i)
				v[i][k] = col.o[i] >> 10;
		}
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			Idct1D r = idct1d_wide<MIJ_PASS2_BIAS>(v[i][0], v[i][1], v[i][2], v[i][3], v[i][4], v[i][5], v[i][6], v[i][7]);
			pack_row(r, rows[i].x, rows[i].y);
		}
	}
}

/* coefficient chunk address of block L, chunk k, in a tile-layout plane (bytes) */
__device__ __forceinline__ size_t tile_chunk_off(uint32_t L, int k) { return ((size_t)(L >> 6) << 13) + ((size_t)k << 10) + ((size_t)(L & 63u) << 4); }

__device__ __forceinline__ void load_block(const uint8_t *__restrict__ plane, uint32_t L, uint4 (&c)[8])
{
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const uint8_t *p = plane + tile_chunk_off((MIJ_VARIANT & 2) ? (L & 63u) : L, k); /* ablation bit 2: one cache-resident tile */
		if (MIJ_VARIANT & 16) { /* ablation: what byte-sized AC coefficients would cost and save -- half the bytes, one v_perm per pair */
			const uint8_t *p8 = plane + tile_chunk_off(L, k) / 2;
			const uint2 h = *reinterpret_cast<const uint2 *>(p8);
			c[k] = make_uint4(__builtin_amdgcn_perm(0, h.x, 0x0c010c00u), __builtin_amdgcn_perm(0, h.x, 0x0c030c02u), __builtin_amdgcn_perm(0, h.y, 0x0c010c00u),
									__builtin_amdgcn_perm(0, h.y, 0x0c030c02u));
			continue;
		}
		/* coefficients are read once, pixels written once: streaming (nt) accesses, measured -2.4 % kernel time */
		const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(p));
		c[k] = make_uint4(v.x, v.y, v.z, v.w);
	}
}

/* where one component's coefficients are, for either format */
struct CoefView {
	const uint8_t *plane, *dc, *hi;
};
__device__ __forceinline__ CoefView coef_view(const uint8_t *__restrict__ coef, const DevComp &cp)
{
	CoefView v;
	v.plane = coef + cp.coef_off;
	v.dc = coef + cp.dc_off;
	v.hi = coef + cp.hi_off;
	return v;
}

/* The same block from COMPACT planes (the default format in HBM; written by the GPU entropy stage and by
 * k_pack_c8): tiles of 4 KiB, chunk k = column k as eight low bytes in the row order of the int16 layout, DC
 * (row 0 of column 0) in its own int16 array, the byte in its place holds the block's flags.  Unpacking and
 * de-quantising are one step: v_mul_i32_i24 with an SDWA byte select (sign-extended) times the 16-bit quantiser,
 * its low half written into one half of the destination -- two instructions per pair, the pair comes out as
 * (short)(coef * dequant) (codec/jpeg.c:325-365).  dq = the component's table in the in-block pair order
 * (wave-uniform).  Half the bytes of the int16 plane.
 * Escapes: the reference's coefficients are 16-bit (codec/jpeg.c:250-265: up to 15 magnitude bits), so a block
 * may hold values a byte cannot.  Such a block has bit 0 of its flags byte set and 64 bytes h[P] at hi + 64*L
 * with coef == sext8(low) + 256*h (mod 2^16); (short)(coef*q) == (short)(sext8(low)*q) + ((h*q & 255) << 8), so
 * the fix is, per pair, two SDWA byte multiplies into bytes 1 and 3 of a zeroed register and one v_pk_add_u16.
 * The branch is wave-uniform (any lane escaped); lanes without escapes add zero. */
/* the unpack + de-quantise step on a block's loaded low bytes h[] and DC term (escape bytes are fetched here when a lane needs them) */
__device__ __forceinline__ void dequant_block_b8(const CoefView &cv, uint32_t L, const uint32_t *__restrict__ dq, const uint2 (&h)[8], uint32_t dc, uint4 (&c)[8])
{
	const bool esc = (h[0].x & 1u) != 0;
	/* the low halves of the four pairs first, then the high halves: the instruction that preserves a register's
	 * other half never directly follows the one that wrote it (dst_sel forwarding), and the s_nop covers the first
	 * reader behind the block */
#define MIJ_DQLO(dst, src, q, b) "v_mul_i32_i24_sdwa " dst ", sext(" src "), " q " dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:" b " src1_sel:WORD_0\n\t"
#define MIJ_DQHI(dst, src, q, b) "v_mul_i32_i24_sdwa " dst ", sext(" src "), " q " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:" b " src1_sel:WORD_1\n\t"
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		uint32_t x, y, z, w;
		asm(MIJ_DQLO("%0", "%4", "%6", "BYTE_0") MIJ_DQLO("%1", "%4", "%7", "BYTE_2") MIJ_DQLO("%2", "%5", "%8", "BYTE_0") MIJ_DQLO("%3", "%5", "%9", "BYTE_2")
				 MIJ_DQHI("%0", "%4", "%6", "BYTE_1") MIJ_DQHI("%1", "%4", "%7", "BYTE_3") MIJ_DQHI("%2", "%5", "%8", "BYTE_1") MIJ_DQHI("%3", "%5", "%9", "BYTE_3") "s_nop 0"
			 : "=&v"(x), "=&v"(y), "=&v"(z), "=&v"(w)
			 : "v"(h[k].x), "v"(h[k].y), "s"(dq[4 * k + 0]), "s"(dq[4 * k + 1]), "s"(dq[4 * k + 2]), "s"(dq[4 * k + 3]));
		c[k] = make_uint4(x, y, z, w);
	}
#undef MIJ_DQLO
#undef MIJ_DQHI
	if (__builtin_amdgcn_ballot_w64(esc) != 0ull) { /* wave-uniform */
		uint4 e[4] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
		if (esc) {
			const uint4 *hp = reinterpret_cast<const uint4 *>(cv.hi + ((size_t)L << 6));
			e[0] = hp[0];
			e[1] = hp[1];
			e[2] = hp[2];
			e[3] = hp[3];
		}
		/* byte 1 <- low byte of h_a * q_lo (rest zero), byte 3 <- low byte of h_b * q_hi, then the packed add */
#define MIJ_EX1(dst, src, q, b) "v_mul_u32_u24_sdwa " dst ", " src ", " q " dst_sel:BYTE_1 dst_unused:UNUSED_PAD src0_sel:" b " src1_sel:WORD_0\n\t"
#define MIJ_EX3(dst, src, q, b) "v_mul_u32_u24_sdwa " dst ", " src ", " q " dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:" b " src1_sel:WORD_1\n\t"
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const uint32_t ex = (k & 1) ? e[k >> 1].z : e[k >> 1].x, ey = (k & 1) ? e[k >> 1].w : e[k >> 1].y;
			uint32_t fx, fy, fz, fw;
			asm(MIJ_EX1("%0", "%4", "%6", "BYTE_0") MIJ_EX1("%1", "%4", "%7", "BYTE_2") MIJ_EX1("%2", "%5", "%8", "BYTE_0") MIJ_EX1("%3", "%5", "%9", "BYTE_2")
					 MIJ_EX3("%0", "%4", "%6", "BYTE_1") MIJ_EX3("%1", "%4", "%7", "BYTE_3") MIJ_EX3("%2", "%5", "%8", "BYTE_1") MIJ_EX3("%3", "%5", "%9", "BYTE_3") "s_nop 0"
				 : "=&v"(fx), "=&v"(fy), "=&v"(fz), "=&v"(fw)
				 : "v"(ex), "v"(ey), "s"(dq[4 * k + 0]), "s"(dq[4 * k + 1]), "s"(dq[4 * k + 2]), "s"(dq[4 * k + 3]));
			c[k].x = __builtin_bit_cast(uint32_t, (v2u)(__builtin_bit_cast(v2u, c[k].x) + __builtin_bit_cast(v2u, fx)));
			c[k].y = __builtin_bit_cast(uint32_t, (v2u)(__builtin_bit_cast(v2u, c[k].y) + __builtin_bit_cast(v2u, fy)));
			c[k].z = __builtin_bit_cast(uint32_t, (v2u)(__builtin_bit_cast(v2u, c[k].z) + __builtin_bit_cast(v2u, fz)));
			c[k].w = __builtin_bit_cast(uint32_t, (v2u)(__builtin_bit_cast(v2u, c[k].w) + __builtin_bit_cast(v2u, fw)));
		}
#undef MIJ_EX1
#undef MIJ_EX3
	}
	/* (short)(DC * dequant[0]) into the low half of the (r0, r4) pair of column 0 (whose low byte held the flags) */
	asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n\ts_nop 0"
		 : "+v"(c[0].x)
		 : "v"(dc), "s"(dq[0]));
}

__device__ __forceinline__ void load_block_b8(const CoefView &cv, uint32_t L, const uint32_t *__restrict__ dq, uint4 (&c)[8])
{
	const uint8_t *base = cv.plane + ((size_t)(L >> 6) << 12) + ((size_t)(L & 63u) << 3);
	uint2 h[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		typedef uint32_t u2v __attribute__((ext_vector_type(2)));
		const u2v v = __builtin_nontemporal_load(reinterpret_cast<const u2v *>(base + (k << 9)));
		h[k] = make_uint2(v.x, v.y);
	}
	const uint32_t dc = *reinterpret_cast<const uint16_t *>(cv.dc + 2u * (size_t)L);
	dequant_block_b8(cv, L, dq, h, dc, c);
}

/* one block in either format: the coefficients of c[] come out de-quantised for B8, quantised otherwise */
template <bool B8>
__device__ __forceinline__ void load_block_fmt(const CoefView &cv, uint32_t L, const uint32_t *__restrict__ dq, uint4 (&c)[8])
{
	if constexpr (B8)
		load_block_b8(cv, L, dq, c);
	else
		load_block(cv.plane, L, c);
}

/* ------------------------------------------------------------------ sparse blocks (round 3)
 *
 * The reference skips work on sparse blocks one column at a time (codec/jpeg.c:625-633: a column whose rows 1..7 are zero is
 * "dcterm = d[0]*4").  A lane cannot branch per column, but a WAVE can branch per block class: most chroma blocks of an ordinary
 * picture hold a DC term and a handful of low frequencies (the benchmark's 1080p batch: 89 % of the Cb wavefronts are DC-only, every
 * Cr wavefront lies inside the top-left 2x2, luma is dense), so the 64 blocks of a wavefront are classified by the extent of their
 * non-zero coefficients and the wave takes the cheapest transform that covers all of them:
 *     class 0  DC only          sample = clamp(((short)(dc*q0) * 16384 + 65536 + (128<<17)) >> 17) for all 64 positions
 *     class 1  inside the 2x2   column pass on columns 0-1 with s2..s7 = 0, row pass with inputs 2..7 = 0
 *     class 2  inside the 4x4   column pass on columns 0-3 with s4..s7 = 0, row pass with inputs 4..7 = 0
 *     class 3  anything         the full transform (idct_block)
 * Exact by construction: the dropped terms are products with zero (the transform is linear over Z/2^32, see idct1d_packed), the
 * dropped columns give (0 + 512) >> 10 = 0, and the packed second pass needs the same int16 guarantee as the full one.  The class
 * is computed from the loaded coefficients themselves, widest class first (a dense wavefront leaves after five OR instructions and
 * one ballot; a sparse one spends about twenty) -- not from a flag a producer wrote -- so no plane a producer gets wrong can change
 * a pixel; an escaped block (a coefficient beyond a byte) is class 3.  Streams that need the WIDE second pass take class 3
 * throughout.  Measured and dropped (round 3, tools/ab3.sh, three builds interleaved on one device): the class carried in a 32-bit
 * DC word per block, read one wave-task ahead so that only the chunks the class needs are fetched at all (0 / 2 / 4 / 8 of eight) --
 * 10 % fewer bytes read, and slower: 2.04 against 2.01-2.02 ms on the headline batch, 2.32 against 2.28 ms on the harsh one.  The
 * kernel is bound by instruction issue, not by bytes, and the dependent load costs more than the classification it saves.
 * MIJ_DEV_COUNT_CLASSES in DevImage.flags: lane 0 of every wavefront adds one to g_idct_class[class] (measurement only). */
#define MIJ_DEV_COUNT_CLASSES 0x200
__device__ unsigned long long g_idct_class[4];

__device__ __forceinline__ uint32_t or3(uint32_t a, uint32_t b, uint32_t c) { return a | b | c; }

/* compact planes: h[k] = column k as the bytes (r0 r4 r2 r6 | r1 r3 r5 r7); byte 0 of h[0].x is the block's flags byte (bit 0: escaped).
 * Widest class first, so that a wavefront of dense blocks leaves after five instructions and a ballot. */
__device__ __forceinline__ int block_class_b8(const uint2 (&h)[8])
{
	/* columns 4-7 (and an escaped block) */
	const uint32_t z47 = or3(or3(h[4].x, h[4].y, h[5].x), or3(h[5].y, h[6].x, h[6].y), or3(h[7].x, h[7].y, h[0].x & 1u));
	if (__builtin_amdgcn_ballot_w64(z47 != 0u) != 0ull)
		return 3;
	const uint32_t x01 = h[0].x | h[1].x, y01 = h[0].y | h[1].y, x23 = h[2].x | h[3].x, y23 = h[2].y | h[3].y;
	/* rows 4-7 of columns 0-3 (r4, r6: bytes 1, 3 of .x; r5, r7: bytes 2, 3 of .y) */
	const uint32_t bad4 = ((x01 | x23) & 0xff00ff00u) | ((y01 | y23) & 0xffff0000u);
	if (__builtin_amdgcn_ballot_w64(bad4 != 0u) != 0ull)
		return 3;
	/* outside the 2x2: columns 2-3 and rows 2-3 of columns 0-1 (r2: byte 2 of .x; r3: byte 1 of .y) */
	const uint32_t bad2 = or3(x23 | y23, x01 & 0x00ff0000u, y01 & 0x0000ff00u);
	if (__builtin_amdgcn_ballot_w64(bad2 != 0u) != 0ull)
		return 2;
	/* any AC term: column 1 and row 1 of column 0 (byte 0 of h[0].x is not a coefficient) */
	const uint32_t bad1 = or3(h[1].x, h[1].y, h[0].y & 0xffu);
	return __builtin_amdgcn_ballot_w64(bad1 != 0u) != 0ull ? 1 : 0;
}

/* int16 tile layout: c[k] = column k as the pairs .x = (r0, r4) .y = (r2, r6) .z = (r1, r3) .w = (r5, r7), quantised */
__device__ __forceinline__ int block_class_i16(const uint4 (&c)[8])
{
	uint32_t z47 = 0;
#pragma unroll
	for (int k = 4; k < 8; ++k)
		z47 = or3(z47, c[k].x | c[k].y, c[k].z | c[k].w);
	if (__builtin_amdgcn_ballot_w64(z47 != 0u) != 0ull)
		return 3;
	const uint32_t xy01 = or3(c[0].x, c[0].y, c[1].x | c[1].y), xy23 = or3(c[2].x, c[2].y, c[3].x | c[3].y);
	const uint32_t bad4 = or3(or3(c[0].w, c[1].w, c[2].w | c[3].w), (xy01 | xy23) & 0xffff0000u, 0u);
	if (__builtin_amdgcn_ballot_w64(bad4 != 0u) != 0ull)
		return 3;
	const uint32_t bad2 = or3(or3(xy23, c[2].z, c[3].z), or3(c[0].y, c[1].y, (c[0].z | c[1].z) & 0xffff0000u), 0u);
	if (__builtin_amdgcn_ballot_w64(bad2 != 0u) != 0ull)
		return 2;
	const uint32_t bad1 = or3(or3(c[1].x, c[1].z, c[0].z), c[0].x & 0xffff0000u, 0u);
	return __builtin_amdgcn_ballot_w64(bad1 != 0u) != 0ull ? 1 : 0;
}

/* (v >> S) as an int16 in the low half, zero above (so that the untouched high input of a v_dot2 contributes nothing) */
template <int S>
__device__ __forceinline__ uint32_t shr_lo16(int v) { return __builtin_amdgcn_ubfe((uint32_t)v, S, 16); }
/* ((lo >> S) | (hi >> S) << 16) for one pair: see shr_pack8_i16 */
template <int S>
__device__ __forceinline__ uint32_t shr_pack_i16(int lo, int hi)
{
	uint32_t o;
	asm("v_ashrrev_i32 %0, %3, %1\n\tv_ashrrev_i32_sdwa %0, %3, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\ts_nop 0"
		 : "=&v"(o)
		 : "v"(lo), "v"(hi), "n"(S));
	return o;
}

/* STBI__IDCT_1D with s4..s7 = 0 (N = 4: d04 = (s0, 0), d26 = (s2, 0), d13 = (s1, s3)) or s2..s7 = 0 (N = 2: d04 = (s0, 0),
 * d13 = (s1, 0)): idct1d_packed minus the products with zero */
template <int N>
__device__ __forceinline__ Idct1D idct1d_low(const IdctK &K, int bias, uint32_t d04, uint32_t d26, uint32_t d13)
{
	const int e = dot2(d04, K.e0, bias);
	uint32_t x0 = (uint32_t)e, x1 = (uint32_t)e, x2 = (uint32_t)e, x3 = (uint32_t)e;
	if (N == 4) {
		x0 = (uint32_t)dot2(d26, K.x0, e);
		x3 = (uint32_t)dot2(d26, K.x3, e);
		x1 = (uint32_t)dot2(d26, K.x1, e);
		x2 = (uint32_t)dot2(d26, K.x2, e);
	}
	const uint32_t t3 = (uint32_t)dot2z(d13, K.t3a), t2 = (uint32_t)dot2z(d13, K.t2a), t1 = (uint32_t)dot2z(d13, K.t1a), t0 = (uint32_t)dot2z(d13, K.t0a);
	Idct1D r;
	r.o[0] = (int)(x0 + t3);
	r.o[7] = (int)(x0 - t3);
	r.o[1] = (int)(x1 + t2);
	r.o[6] = (int)(x1 - t2);
	r.o[2] = (int)(x2 + t1);
	r.o[5] = (int)(x2 - t1);
	r.o[3] = (int)(x3 + t0);
	r.o[4] = (int)(x3 - t0);
	return r;
}

/* classes 1 and 2 on de-quantised pairs: q[k] = column k as (s0, 0) (s2, 0) (s1, s3) -- N = 2: (s0, 0) - (s1, 0) */
template <int N>
__device__ __forceinline__ void idct_block_low(const IdctK &K, const uint32_t (&q04)[4], const uint32_t (&q26)[4], const uint32_t (&q13)[4], uint2 (&rows)[8])
{
	Idct1D v[N];
#pragma unroll
	for (int k = 0; k < N; ++k)
		v[k] = idct1d_low<N>(K, K.bias1, q04[k], q26[k], q13[k]);
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		const uint32_t p04 = shr_lo16<10>(v[0].o[i]);
		const uint32_t p26 = N == 4 ? shr_lo16<10>(v[2].o[i]) : 0u;
		const uint32_t p13 = N == 4 ? shr_pack_i16<10>(v[1].o[i], v[3].o[i]) : shr_lo16<10>(v[1].o[i]);
		const Idct1D r = idct1d_low<N>(K, K.bias2, p04, p26, p13);
		pack_row(r, rows[i].x, rows[i].y);
	}
}

/* class 0: every sample of the block is clamp((dcq * 16384 + 65536 + (128 << 17)) >> 17), dcq = (short)(dc * q0) sign-extended */
__device__ __forceinline__ void idct_block_dc(const IdctK &K, int dcq, uint2 (&rows)[8])
{
	const int x = dcq * 16384 + K.bias2;
	const uint32_t p = sat4<17>(x, x, x, x);
#pragma unroll
	for (int i = 0; i < 8; ++i)
		rows[i] = make_uint2(p, p);
}

/* a block's loaded low bytes and DC term (compact planes): the loads of several blocks can be in flight before the first is transformed */
struct RawB8 {
	uint2 h[8];
	uint32_t dc;
};
__device__ __forceinline__ void load_raw_b8(const CoefView &cv, uint32_t L, RawB8 &r)
{
	const uint8_t *base = cv.plane + ((size_t)(L >> 6) << 12) + ((size_t)(L & 63u) << 3);
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		typedef uint32_t u2v __attribute__((ext_vector_type(2)));
		const u2v v = __builtin_nontemporal_load(reinterpret_cast<const u2v *>(base + (k << 9)));
		r.h[k] = make_uint2(v.x, v.y);
	}
	r.dc = *reinterpret_cast<const uint16_t *>(cv.dc + 2u * (size_t)L);
}
/* classify the wavefront's loaded blocks and run the cheapest transform that covers them (not WIDE); returns the class */
__device__ __forceinline__ int idct_raw_b8(const IdctK &K, const CoefView &cv, uint32_t L, const uint32_t *__restrict__ dq, const RawB8 &raw, uint2 (&rows)[8])
{
	const uint2 (&h)[8] = raw.h;
	const uint32_t dc = raw.dc;
	uint4 c[8];
	int cls;
		cls = block_class_b8(h);
		if (cls == 3) {
			dequant_block_b8(cv, L, dq, h, dc, c);
			idct_block<false, true>(K, c, dq, rows);
		} else if (cls == 0) {
			int dcq;
			asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0\n\tv_bfe_i32 %0, %0, 0, 16" : "=&v"(dcq) : "v"(dc), "s"(dq[0]));
			idct_block_dc(K, dcq, rows);
		} else {
			/* (short)(coef * q) of the terms the class keeps: low halves with a zero above them, (s1, s3) as a pair for the 4x4 */
			uint32_t q04[4], q26[4], q13[4];
#define MIJ_DQP(dst, src, q, b) "v_mul_i32_i24_sdwa " dst ", sext(" src "), " q " dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:" b " src1_sel:WORD_0\n\t"
#define MIJ_DQH(dst, src, q, b) "v_mul_i32_i24_sdwa " dst ", sext(" src "), " q " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:" b " src1_sel:WORD_1\n\t"
			if (cls == 2) {
#pragma unroll
				for (int k = 0; k < 4; ++k)
					asm(MIJ_DQP("%0", "%3", "%5", "BYTE_0") MIJ_DQP("%1", "%3", "%6", "BYTE_2") MIJ_DQP("%2", "%4", "%7", "BYTE_0") MIJ_DQH("%2", "%4", "%7", "BYTE_1") "s_nop 0"
						 : "=&v"(q04[k]), "=&v"(q26[k]), "=&v"(q13[k])
						 : "v"(h[k].x), "v"(h[k].y), "s"(dq[4 * k + 0]), "s"(dq[4 * k + 1]), "s"(dq[4 * k + 2]));
			} else {
#pragma unroll
				for (int k = 0; k < 2; ++k) {
					asm(MIJ_DQP("%0", "%2", "%4", "BYTE_0") MIJ_DQP("%1", "%3", "%5", "BYTE_0") "s_nop 0"
						 : "=&v"(q04[k]), "=&v"(q13[k])
						 : "v"(h[k].x), "v"(h[k].y), "s"(dq[4 * k + 0]), "s"(dq[4 * k + 2]));
					q26[k] = 0;
				}
				q04[2] = q04[3] = q26[2] = q26[3] = q13[2] = q13[3] = 0;
			}
#undef MIJ_DQP
#undef MIJ_DQH
			/* the DC term replaces the flags byte's product */
			asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0\n\ts_nop 0" : "=v"(q04[0]) : "v"(dc), "s"(dq[0]));
			if (cls == 2)
				idct_block_low<4>(K, q04, q26, q13, rows);
			else
				idct_block_low<2>(K, q04, q26, q13, rows);
		}
	return cls;
}

/* One block per lane from either plane format through the cheapest transform that covers the wavefront's blocks.
 * Returns the class taken (wave-uniform).  count != 0 (wave-uniform): measurement, see MIJ_DEV_COUNT_CLASSES. */
template <bool WIDE, bool B8>
__device__ __forceinline__ int load_idct_block(const IdctK &K, const CoefView &cv, uint32_t L, const uint32_t *__restrict__ dq, uint2 (&rows)[8], int count)
{
	uint4 c[8];
	int cls = 3;
	if constexpr (WIDE) {
		load_block_fmt<B8>(cv, L, dq, c);
		idct_block<WIDE, B8>(K, c, dq, rows);
	} else if constexpr (B8) {
		RawB8 raw;
		load_raw_b8(cv, L, raw);
		cls = idct_raw_b8(K, cv, L, dq, raw, rows);
	} else {
		load_block(cv.plane, L, c);
		cls = block_class_i16(c);
		if (cls == 3) {
			idct_block<WIDE, B8>(K, c, dq, rows);
		} else if (cls == 0) {
			idct_block_dc(K, (int)(short)(pkmul(c[0].x, dq[0]) & 0xffffu), rows);
		} else {
			uint32_t q04[4], q26[4], q13[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) { /* the class guarantees zero high halves where a pair is (s, 0) */
				q04[k] = pkmul(c[k].x, dq[4 * k + 0]);
				q26[k] = pkmul(c[k].y, dq[4 * k + 1]);
				q13[k] = pkmul(c[k].z, dq[4 * k + 2]);
			}
			if (cls == 2)
				idct_block_low<4>(K, q04, q26, q13, rows);
			else
				idct_block_low<2>(K, q04, q26, q13, rows);
		}
	}
	if (count) {
		if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
			atomicAdd(&g_idct_class[cls], 1ull);
	}
	return cls;
}

/* ------------------------------------------------------------------ colour (codec/jpeg.c:1976-2018)
 * All four stbi__float2fixed constants are multiples of 256, so the >>20 of the reference equals
 * >>12 of the same sum divided by 256:  r = (4096 y + 2048 + 5743 cr') >> 12, etc. with
 * cr' = cr - 128; the green channel's "& 0xffff0000" becomes "& ~255" on the cb term.
 */
__device__ __forceinline__ void ycbcr_to_rgb(int y, int cb, int cr, int &r, int &g, int &b)
{
	int yf = (y << 12) + 2048;
	int crc = cr - 128, cbc = cb - 128;
	r = clamp255(opaque((yf + crc * 5743) >> 12));
	g = clamp255(opaque((yf + crc * -2925 + ((cbc * -1410) & ~255)) >> 12));
	b = clamp255(opaque((yf + cbc * 7258) >> 12));
}

__device__ __forceinline__ int compute_y(int r, int g, int b) { return ((r * 77) + (g * 150) + (29 * b)) >> 8 & 255; } /* common.c:173 */
__device__ __forceinline__ int blinn8(int x, int y)                                                                  /* codec/jpeg.c:2218 */
{
	unsigned t = (unsigned)(x * y + 128);
	return (int)((t + (t >> 8)) >> 8) & 255;
}

/* ------------------------------------------------------------------ generic sampling (two-pass path)
 *
 * Row scheduling of load_jpeg_image (codec/jpeg.c:2273-2318) in closed form, for any vs in 1..4:
 *   g = (r + (vs>>1)) / vs, phase = (r + (vs>>1)) % vs, y_bot = phase >= (vs>>1)
 *   line1 = min(g, y-1), line0 = g ? min(g-1, y-1) : 0;  near = y_bot ? line1 : line0, far = the other.
 */
struct RowSel {
	int near, far;
};
__device__ __forceinline__ RowSel select_rows(int r, int vs, int ycomp)
{
	int half = vs >> 1;
	int q = r + half;
	int g = vs == 1 ? q : (vs == 2 ? q >> 1 : (vs == 4 ? q >> 2 : q / vs)), phase = q - g * vs;
	int l1 = min(g, ycomp - 1);
	int l0 = g ? min(g - 1, ycomp - 1) : 0;
	RowSel s;
	if (phase >= half) {
		s.near = l1;
		s.far = l0;
	} else {
		s.near = l0;
		s.far = l1;
	}
	return s;
}

/* the colour branches of load_jpeg_image (codec/jpeg.c:2320-2431); s[] = up-sampled components */
__device__ __forceinline__ void store_pixel(uint8_t *__restrict__ out, int n, int color, const int (&s)[4])
{
	int r, g, b;
	if (n >= 3) {
		switch (color) {
		case MIJ_COLOR_YCBCR:
		case MIJ_COLOR_YCBCRA:
			ycbcr_to_rgb(s[0], s[1], s[2], r, g, b);
			break;
		case MIJ_COLOR_RGB:
			r = s[0], g = s[1], b = s[2];
			break;
		case MIJ_COLOR_CMYK:
			r = blinn8(s[0], s[3]), g = blinn8(s[1], s[3]), b = blinn8(s[2], s[3]);
			break;
		case MIJ_COLOR_YCCK:
			ycbcr_to_rgb(s[0], s[1], s[2], r, g, b);
			r = blinn8(255 - r, s[3]), g = blinn8(255 - g, s[3]), b = blinn8(255 - b, s[3]);
			break;
		default: /* grey */
			r = g = b = s[0];
			break;
		}
		out[0] = (uint8_t)r;
		out[1] = (uint8_t)g;
		out[2] = (uint8_t)b;
		if (n == 4)
			out[3] = 255;
	} else {
		int y;
		switch (color) {
		case MIJ_COLOR_RGB:
			y = compute_y(s[0], s[1], s[2]);
			break;
		case MIJ_COLOR_CMYK:
			y = compute_y(blinn8(s[0], s[3]), blinn8(s[1], s[3]), blinn8(s[2], s[3]));
			break;
		case MIJ_COLOR_YCCK:
			y = blinn8(255 - s[0], s[3]);
			break;
		default: /* grey, YCbCr luma-only, YCbCr+A */
			y = s[0];
			break;
		}
		out[0] = (uint8_t)y;
		if (n == 2)
			out[1] = 255;
	}
}

/* which components a colour mode reads */
__device__ __forceinline__ int comps_needed(int color, int n, int ncomp)
{
	if (color == MIJ_COLOR_GREY)
		return 1;
	if (n < 3 && color == MIJ_COLOR_YCBCRA)
		return 1;
	if (color == MIJ_COLOR_YCBCRA && n >= 3)
		return 3;
	return ncomp;
}

/* ------------------------------------------------------------------ two-pass path, pass 1 */

template <bool WIDE, bool B8 = false>
__global__ __launch_bounds__(256) void k_idct_planes(const DevImage *__restrict__ imgs, const WorkIdct *__restrict__ work, const uint8_t *__restrict__ coef,
																	  uint8_t *__restrict__ planes)
{
	const WorkIdct wk = work[blockIdx.x];
	const DevImage &im = imgs[wk.img];
	const DevComp &cp = im.comp[wk.comp];
	const uint32_t nblk = (uint32_t)(cp.bw * cp.bh);
	const uint32_t L = wk.first + threadIdx.x;
	if (L >= nblk)
		return;
	IdctK K;
	K.init();
	uint2 rows[8];
	load_idct_block<WIDE, B8>(K, coef_view(coef, cp), L, im.dq[wk.comp], rows, im.flags & MIJ_DEV_COUNT_CLASSES);
	const uint32_t by = L / (uint32_t)cp.bw, bx = L - by * (uint32_t)cp.bw;
	const size_t w2 = (size_t)cp.bw * 8;
	uint8_t *dst = planes + cp.plane_off + (size_t)by * 8 * w2 + (size_t)bx * 8;
#pragma unroll
	for (int i = 0; i < 8; ++i)
		*reinterpret_cast<uint2 *>(dst + (size_t)i * w2) = rows[i];
}

/* ------------------------------------------------------------------ int16 planes -> compact planes
 * Every producer that stages int16 coefficients on the host (the host Huffman walk; the progressive scans, which
 * read-modify-write their planes, codec/jpeg.c:372-558) uploads them into a scratch buffer and this kernel packs
 * them into the compact format the decode kernels read (MIJ_DEV_COEF_BYTES above), so an image with coefficients
 * beyond a byte never leaves the GPU path: its blocks get escape bytes.  One block per lane, whole tiles (the
 * padding blocks of the last tile are zero in the source and come out as zero).
 * work.first = first block, work.comp = component. */
/* MIJ_DEV_L1_MAX in DevImage.flags (progressive files, MIJ_FLAG_L1_ON_DEVICE): every block's sum of |(short)(coef * q)| -- the bound behind
 * MIJ_FLAG_WIDE_IDCT, which the host computes for baseline files while it walks them and would need one more pass over 100 MB of planes for a
 * 4096 x 4096 progressive file -- is taken here, where every coefficient passes through registers anyway, and its maximum over the image
 * lands in l1max[image] (one atomic per wavefront). */
#define MIJ_DEV_L1_MAX 0x400
__global__ __launch_bounds__(256) void k_pack_c8(const DevImage *__restrict__ imgs, const WorkIdct *__restrict__ work, const uint8_t *__restrict__ src16,
																 uint8_t *__restrict__ coef, uint32_t *__restrict__ l1max)
{
	const WorkIdct wk = work[blockIdx.x];
	const DevImage &im = imgs[wk.img];
	const DevComp &cp = im.comp[wk.comp];
	const uint32_t ntile = ((uint32_t)(cp.bw * cp.bh) + 63u) >> 6;
	const uint32_t L = wk.first + threadIdx.x;
	if (L >= ntile * 64u)
		return;
	size_t soff = im.src16_off;
	for (uint32_t k = 0; k < wk.comp; ++k)
		soff += (size_t)(((uint32_t)(im.comp[k].bw * im.comp[k].bh) + 63u) >> 6) << 13;
	uint4 c[8];
	load_block(src16 + soff, L, c);
	uint32_t lo[16], hi[16], any = 0;
	const uint32_t bias = 0x00800080u;
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const uint32_t w[4] = {c[k].x, c[k].y, c[k].z, c[k].w};
		uint32_t t[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			/* x + 128 per 16-bit half: the high byte is h = (x + 128) >> 8, non-zero exactly when x is outside -128..127 */
			t[j] = __builtin_bit_cast(uint32_t, (v2u)(__builtin_bit_cast(v2u, w[j]) + __builtin_bit_cast(v2u, bias)));
			any |= t[j] & ((k == 0 && j == 0) ? 0xff000000u : 0xff00ff00u); /* the DC term has its own array */
		}
		lo[2 * k] = __builtin_amdgcn_perm(w[1], w[0], 0x06040200u);
		lo[2 * k + 1] = __builtin_amdgcn_perm(w[3], w[2], 0x06040200u);
		hi[2 * k] = __builtin_amdgcn_perm(t[1], t[0], 0x07050301u);
		hi[2 * k + 1] = __builtin_amdgcn_perm(t[3], t[2], 0x07050301u);
	}
	if (im.flags & MIJ_DEV_L1_MAX) { /* wave-uniform */
		const uint32_t *dq = im.dq[wk.comp];
		uint32_t l1 = 0;
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const uint32_t w[4] = {c[k].x, c[k].y, c[k].z, c[k].w};
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const v2s m = __builtin_bit_cast(v2s, pkmul(w[j], dq[4 * k + j]));
				const v2s a = __builtin_elementwise_max(m, (v2s)(-m)); /* |-32768| stays 0x8000 = 32768 as an unsigned half */
				const v2u au = __builtin_bit_cast(v2u, a);
				l1 += (uint32_t)au.x + (uint32_t)au.y;
			}
		}
#pragma unroll
		for (int off = 32; off >= 1; off >>= 1) {
			const uint32_t o = (uint32_t)__shfl_xor((int)l1, off, 64);
			l1 = o > l1 ? o : l1;
		}
		if ((threadIdx.x & 63u) == 0u)
			atomicMax(&l1max[wk.img], l1);
	}
	const uint32_t dc = c[0].x & 0xffffu;
	lo[0] = (lo[0] & 0xffffff00u) | (any ? 1u : 0u); /* flags byte in the DC's place */
	hi[0] &= 0xffffff00u;
	uint8_t *dst = coef + cp.coef_off + ((size_t)(L >> 6) << 12) + ((size_t)(L & 63u) << 3);
#pragma unroll
	for (int k = 0; k < 8; ++k)
		*reinterpret_cast<uint2 *>(dst + (k << 9)) = make_uint2(lo[2 * k], lo[2 * k + 1]);
	*reinterpret_cast<uint16_t *>(coef + cp.dc_off + 2u * (size_t)L) = (uint16_t)dc;
	if (any) {
		uint4 *hp = reinterpret_cast<uint4 *>(coef + cp.hi_off + ((size_t)L << 6));
#pragma unroll
		for (int k = 0; k < 4; ++k)
			hp[k] = make_uint4(hi[4 * k], hi[4 * k + 1], hi[4 * k + 2], hi[4 * k + 3]);
	}
}

/* ------------------------------------------------------------------ two-pass path, pass 2 */

/* ------------------------------------------------------------------ fused single-component (grey) kernel
 * One block per lane: IDCT, then the 8x8 samples go straight to the pixel buffer, replicated to n_out
 * channels (codec/jpeg.c:2373-2378, :2380-2430: grey -> y | y,255 | y,y,y | y,y,y,255).  No sample plane. */
template <bool WIDE, bool B8 = false>
__global__ __launch_bounds__(256) void k_fused_grey(const DevImage *__restrict__ imgs, const WorkIdct *__restrict__ work, const uint8_t *__restrict__ coef,
																	 uint8_t *__restrict__ outbase)
{
	const WorkIdct wk = work[blockIdx.x];
	const DevImage &im = imgs[wk.img];
	const DevComp &cp = im.comp[0];
	const uint32_t nblk = (uint32_t)(cp.bw * cp.bh);
	const uint32_t L = wk.first + threadIdx.x;
	if (L >= nblk)
		return;
	IdctK K;
	K.init();
	uint2 rows[8];
	load_idct_block<WIDE, B8>(K, coef_view(coef, cp), L, im.dq[0], rows, im.flags & MIJ_DEV_COUNT_CLASSES);
	const int by = (int)(L / (uint32_t)cp.bw), bx = (int)(L - (uint32_t)by * (uint32_t)cp.bw);
	const int W = im.width, H = im.height, n = im.n_out;
	const int x0 = 8 * bx, y0 = 8 * by;
	if (x0 >= W || y0 >= H)
		return;
	uint8_t *const out = outbase + im.out_off;
	const bool whole = x0 + 8 <= W && ((uint32_t)(W * n) & 3u) == 0; /* dword-aligned rows, all eight columns inside */
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		if (y0 + i >= H)
			break;
		const uint32_t lo = rows[i].x, hi = rows[i].y;
		uint8_t *dst = out + ((size_t)(y0 + i) * W + x0) * n;
		if (whole) {
			uint32_t *q = reinterpret_cast<uint32_t *>(dst);
			if (n == 1) {
				q[0] = lo;
				q[1] = hi;
			} else if (n == 2) { /* selector 0x0d..: constant 0xff */
				q[0] = __builtin_amdgcn_perm(0, lo, 0x0d010d00u);
				q[1] = __builtin_amdgcn_perm(0, lo, 0x0d030d02u);
				q[2] = __builtin_amdgcn_perm(0, hi, 0x0d010d00u);
				q[3] = __builtin_amdgcn_perm(0, hi, 0x0d030d02u);
			} else if (n == 3) {
				q[0] = __builtin_amdgcn_perm(0, lo, 0x01000000u);
				q[1] = __builtin_amdgcn_perm(0, lo, 0x02020101u);
				q[2] = __builtin_amdgcn_perm(0, lo, 0x03030302u);
				q[3] = __builtin_amdgcn_perm(0, hi, 0x01000000u);
				q[4] = __builtin_amdgcn_perm(0, hi, 0x02020101u);
				q[5] = __builtin_amdgcn_perm(0, hi, 0x03030302u);
			} else {
				q[0] = __builtin_amdgcn_perm(0, lo, 0x0d000000u);
				q[1] = __builtin_amdgcn_perm(0, lo, 0x0d010101u);
				q[2] = __builtin_amdgcn_perm(0, lo, 0x0d020202u);
				q[3] = __builtin_amdgcn_perm(0, lo, 0x0d030303u);
				q[4] = __builtin_amdgcn_perm(0, hi, 0x0d000000u);
				q[5] = __builtin_amdgcn_perm(0, hi, 0x0d010101u);
				q[6] = __builtin_amdgcn_perm(0, hi, 0x0d020202u);
				q[7] = __builtin_amdgcn_perm(0, hi, 0x0d030303u);
			}
		} else {
			const int cnt = min(8, W - x0);
			for (int j = 0; j < cnt; ++j) {
				const uint8_t v = (uint8_t)((j < 4 ? lo >> (8 * j) : hi >> (8 * (j - 4))) & 255u);
				dst[j * n] = v;
				if (n >= 3) {
					dst[j * n + 1] = v;
					dst[j * n + 2] = v;
				}
				if (n == 2 || n == 4)
					dst[j * n + n - 1] = 255;
			}
		}
	}
}

/* Four up-sampled samples of one component: output row r, columns x0 .. x0+3 (x0 % 4 == 0, columns past the
 * image clamp to W-1).  Follows resample_row_1 / _v_2 / _h_2 / _hv_2 / _generic
 * (codec/jpeg.c:1765-1840, 1962-1971); row selection and shared neighbour loads once per strip, 32-bit indices. */
struct CompView {
	const uint8_t *p;
	int w2, limit, hs, vs, y;
	__device__ __forceinline__ int at(int row, int col) const
	{
		const int idx = row * w2 + col; /* hs == 1 planes narrower than the image are read past the row end, as the reference does */
		return p[idx > limit ? limit : idx];
	}
};

__device__ __forceinline__ void upsample_strip(const CompView &c, int W, int r, int x0, int (&o)[4])
{
	const RowSel rs = select_rows(r, c.vs, c.y);
	if (c.hs == 2 && (c.vs == 1 || c.vs == 2)) {
		const int w = (W + 1) >> 1; /* w_lores, codec/jpeg.c:2276 */
		const int i0 = x0 >> 1;
		const int ca = max(i0 - 1, 0), cb = min(i0, w - 1), cc = min(i0 + 1, w - 1), cd = min(i0 + 2, w - 1);
		if (c.vs == 2) { /* hv_2 :1816-1840; the end cases are the general form with the neighbour clamped */
			const int na = c.at(rs.near, ca), nb = c.at(rs.near, cb), nc = c.at(rs.near, cc), nd = c.at(rs.near, cd);
			const int ta = 3 * na + c.at(rs.far, ca), tb = 3 * nb + c.at(rs.far, cb), tc = 3 * nc + c.at(rs.far, cc), td = 3 * nd + c.at(rs.far, cd);
			o[0] = (3 * tb + ta + 8) >> 4;
			o[1] = (3 * tb + tc + 8) >> 4;
			o[2] = (3 * tc + tb + 8) >> 4;
			o[3] = (3 * tc + td + 8) >> 4;
		} else { /* h_2 :1784-1812, with its first / last / last-but-one column forms */
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const int col = min(x0 + j, W - 1), i = col >> 1;
				int v;
				if (w == 1 || col == 0)
					v = c.at(rs.near, 0);
				else if (col == 2 * w - 1)
					v = c.at(rs.near, w - 1);
				else if (col == 2 * w - 2) /* the reference's right-edge form, :1805 */
					v = (3 * c.at(rs.near, w - 2) + c.at(rs.near, w - 1) + 2) >> 2;
				else if (col & 1)
					v = (3 * c.at(rs.near, i) + c.at(rs.near, i + 1) + 2) >> 2;
				else
					v = (3 * c.at(rs.near, i) + c.at(rs.near, i - 1) + 2) >> 2;
				o[j] = v;
			}
		}
		return;
	}
	if (c.hs == 1 && c.vs == 1 && x0 + 3 < W && rs.near * c.w2 + x0 + 3 <= c.limit) { /* row_1 on a whole strip: one dword */
		const uint32_t v = *reinterpret_cast<const uint32_t *>(c.p + rs.near * c.w2 + x0);
		o[0] = v & 255;
		o[1] = (v >> 8) & 255;
		o[2] = (v >> 16) & 255;
		o[3] = v >> 24;
		return;
	}
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int col = min(x0 + j, W - 1);
		if (c.hs == 1 && c.vs == 2)
			o[j] = (3 * c.at(rs.near, col) + c.at(rs.far, col) + 2) >> 2; /* v_2 :1774-1782 */
		else if (c.hs == 1)
			o[j] = c.at(rs.near, col); /* row_1, or the generic resampler with hs == 1 */
		else
			o[j] = c.at(rs.near, col / c.hs); /* generic :1962-1971 */
	}
}

/* Pass 2: work item = MIJ_RESAMPLE_ROWS output rows of one image; a thread takes 4-pixel strips, so that
 * the usual case (whole strip, dword-aligned address) leaves as n_out dwords instead of 4*n_out byte stores. */
#define MIJ_RESAMPLE_ROWS 4
__global__ __launch_bounds__(256) void k_resample_color(const DevImage *__restrict__ imgs, const WorkIdct *__restrict__ work, const uint8_t *__restrict__ planes,
																		  uint8_t *__restrict__ outbase)
{
	const WorkIdct wk = work[blockIdx.x];
	const DevImage &im = imgs[wk.img];
	const int W = im.width, n = im.n_out;
	const int r0 = (int)wk.first, r1 = min(r0 + MIJ_RESAMPLE_ROWS, im.height);
	const int spr = (W + 3) >> 2; /* strips per row */
	const int need = comps_needed(im.color, n, im.ncomp);
	CompView C[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const DevComp &cp = im.comp[k < need ? k : 0];
		C[k].p = planes + cp.plane_off;
		C[k].w2 = cp.bw * 8;
		C[k].limit = cp.bw * 8 * cp.bh * 8 - 1;
		C[k].hs = cp.hs;
		C[k].vs = cp.vs;
		C[k].y = cp.y;
	}
	uint8_t *const out = outbase + im.out_off;
	for (int r = r0; r < r1; ++r)
	for (int t = threadIdx.x; t < spr; t += 256) {
		const int x0 = 4 * t, cnt = min(4, W - x0);
		int smp[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}; /* [component][pixel] */
#pragma unroll
		for (int k = 0; k < 4; ++k) /* unrolled with static indices: a run-time bound would put C[] and smp[] in scratch */
			if (k < need)
				upsample_strip(C[k], W, r, x0, smp[k]);
		uint8_t px[4][4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const int s[4] = {smp[0][j], smp[1][j], smp[2][j], smp[3][j]};
			store_pixel(px[j], n, im.color, s);
		}
		const size_t off = ((size_t)r * W + x0) * n;
		uint8_t *dst = out + off;
		if (cnt == 4 && ((im.out_off + off) & 3) == 0) {
			uint32_t *q = reinterpret_cast<uint32_t *>(dst);
			if (n == 4) {
				for (int j = 0; j < 4; ++j)
					q[j] = px[j][0] | px[j][1] << 8 | px[j][2] << 16 | (uint32_t)px[j][3] << 24;
			} else if (n == 3) {
				q[0] = px[0][0] | px[0][1] << 8 | px[0][2] << 16 | (uint32_t)px[1][0] << 24;
				q[1] = px[1][1] | px[1][2] << 8 | px[2][0] << 16 | (uint32_t)px[2][1] << 24;
				q[2] = px[2][2] | px[3][0] << 8 | px[3][1] << 16 | (uint32_t)px[3][2] << 24;
			} else if (n == 2) {
				q[0] = px[0][0] | px[0][1] << 8 | px[1][0] << 16 | (uint32_t)px[1][1] << 24;
				q[1] = px[2][0] | px[2][1] << 8 | px[3][0] << 16 | (uint32_t)px[3][1] << 24;
			} else {
				q[0] = px[0][0] | px[1][0] << 8 | px[2][0] << 16 | (uint32_t)px[3][0] << 24;
			}
		} else {
			for (int j = 0; j < cnt; ++j)
				for (int c = 0; c < n; ++c)
					dst[j * n + c] = px[j][c];
		}
	}
}

/* ------------------------------------------------------------------ fused h2v2 YCbCr kernel
 *
 * LDS (dynamic, bytes; YP = 16*mcu_x, CP = 8*mcu_x):
 *   Y[16][YP]  Cb[8][CP]  Cr[8][CP]            samples of the current MCU row
 *   saveY[2][YP] saveCb[2][CP] saveCr[2][CP]   last Y / chroma row of the previous MCU row (ping-pong)
 * Chroma rows are addressed by absolute row C: rows of the current MCU row come from the planes,
 * row 8m-1 from the save buffer; row pairs (2C-1, 2C) only ever need chroma rows C-1 and C.
 */
struct Fused420Lds {
	uint8_t *Y, *Cb, *Cr, *saveY[2], *saveCb[2], *saveCr[2];
	int YP, CP;
};

__device__ __forceinline__ size_t fused420_lds_bytes(int mcu_x) { return (size_t)mcu_x * (16 * 16 + 2 * 8 * 8 + 2 * 16 + 4 * 8); } /* 4:4:0: 16 * 8 + 128 + 2 * 8 + 32 */

template <int NOUT>
__device__ __forceinline__ void store_rgb_px(uint8_t *__restrict__ p, int r, int g, int b)
{
	p[0] = (uint8_t)r;
	p[1] = (uint8_t)g;
	p[2] = (uint8_t)b;
	if (NOUT == 4)
		p[3] = 255;
}

/* careful per-pixel path for the image's left/right edge strips and unaligned widths:
 * the same closed form as upsample_strip's hv_2 case with the chroma rows already resolved */
template <int NOUT, bool H2 = true>
__device__ __forceinline__ void fused420_pixel(const uint8_t *yrow, const uint8_t *cbA, const uint8_t *cbB, const uint8_t *crA, const uint8_t *crB, int nearIsB,
															  int wc, int x, uint8_t *__restrict__ dst)
{
	if (!H2) { /* v_2: (3 near + far + 2) >> 2 on the pixel's own column (codec/jpeg.c:1774-1782) */
		const int cb = (3 * (nearIsB ? cbB : cbA)[x] + (nearIsB ? cbA : cbB)[x] + 2) >> 2, cr = (3 * (nearIsB ? crB : crA)[x] + (nearIsB ? crA : crB)[x] + 2) >> 2;
		int r, g, b;
		ycbcr_to_rgb(yrow[x], cb, cr, r, g, b);
		store_rgb_px<NOUT>(dst + (size_t)x * NOUT, r, g, b);
		return;
	}
	int i = x >> 1;
	int j = (x & 1) ? min(i + 1, wc - 1) : max(i - 1, 0);
	const uint8_t *cbN = nearIsB ? cbB : cbA, *cbF = nearIsB ? cbA : cbB;
	const uint8_t *crN = nearIsB ? crB : crA, *crF = nearIsB ? crA : crB;
	int tbi = 3 * cbN[i] + cbF[i], tbj = 3 * cbN[j] + cbF[j];
	int tri = 3 * crN[i] + crF[i], trj = 3 * crN[j] + crF[j];
	int cb = (3 * tbi + tbj + 8) >> 4, cr = (3 * tri + trj + 8) >> 4;
	int r, g, b;
	ycbcr_to_rgb(yrow[x], cb, cr, r, g, b);
	store_rgb_px<NOUT>(dst + (size_t)x * NOUT, r, g, b);
}

/* colour constants pinned in vector registers (see color_px) */
struct ColorK {
	uint32_t r, b, t, g, mask, o80, w128;
	int kr, kb, kt;
	uint32_t wBk, wBk1, wAk, wAk1;
	uint32_t v0, v1, v2, p0, p1, p2, p3;
	__device__ __forceinline__ void init()
	{
		r = kconst(pk16(5743, 4096));
		b = kconst(pk16(7258, 4096));
		t = kconst(pk16(-1410, 0));
		g = kconst(pk16(-2925, 4096));
		mask = vreg(0xffffff00u);
		o80 = kconst(0x80u);
		w128 = vreg(128u);
		kr = (int)vreg((uint32_t)(2048 - 128 * 5743));
		kb = (int)vreg((uint32_t)(2048 - 128 * 7258));
		kt = (int)vreg((uint32_t)(128 * 1410 + 0x5BE00));
		/* h2v2 weights x16, byte order of the operand is (B_k, A_k, B_k+1, A_k+1) */
		wBk = kconst(0x10303090u);  /* near = B, centre = k   : 9B_k + 3A_k + 3B_k1 +  A_k1 */
		wBk1 = kconst(0x30901030u); /* near = B, centre = k+1 : 3B_k +  A_k + 9B_k1 + 3A_k1 */
		wAk = kconst(0x30109030u);  /* near = A, centre = k   : 3B_k + 9A_k +  B_k1 + 3A_k1 */
		wAk1 = kconst(0x90303010u); /* near = A, centre = k+1 :  B_k + 3A_k + 3B_k1 + 9A_k1 */
		/* v_perm selectors: V_k = (B[k], A[k], B[k+1], A[k+1]); (chroma byte 1 | luma byte j << 16) */
		v0 = kconst(0x05010400u);
		v1 = kconst(0x06020501u);
		v2 = kconst(0x07030602u);
		p0 = kconst(0x0c000c05u);
		p1 = kconst(0x0c010c05u);
		p2 = kconst(0x0c020c05u);
		p3 = kconst(0x0c030c05u);
	}
};

/* colour sums (before the >> 12) for one pixel from packed operands pcr = (cr | y << 16), pcb = (cb | y << 16):
 * r = 4096 y + 2048 + 5743 (cr-128), etc., the constants folded into the accumulators
 * (codec/jpeg.c:1981-1990 with every fixed-point constant divided by 256, see ycbcr_to_rgb) */
struct Rgb12 {
	int r, g, b;
};
__device__ __forceinline__ Rgb12 color_px(const ColorK &K, uint32_t pcr, uint32_t pcb)
{
	Rgb12 c;
	c.r = dot2(pcr, K.r, K.kr);
	c.b = dot2(pcb, K.b, K.kb);
	/* ((cb-128) * -1410) & ~255, plus 2048 + 128*2925 = 0x5BE80 split as 0x5BE00 (commutes with the mask) | 0x80 */
	uint32_t t = (uint32_t)dot2(pcb, K.t, K.kt);
	c.g = dot2(pcr, K.g, (int)and_or(t, K.mask, K.o80));
	return c;
}

/* four pixels -> 12 (RGB) or 16 (RGBA) bytes: ">> 12 then clamp" for two samples per instruction */
template <int NOUT>
__device__ __forceinline__ void store_px4(uint8_t *__restrict__ dst, const Rgb12 &p0, const Rgb12 &p1, const Rgb12 &p2, const Rgb12 &p3)
{
	if ((MIJ_VARIANT & 1) && (p0.r ^ p0.g ^ p0.b ^ p1.r ^ p1.g ^ p1.b ^ p2.r ^ p2.g ^ p2.b ^ p3.r ^ p3.g ^ p3.b) != 0x12345678) /* ablation: no stores (and xors instead of the packing) */
		return;
	if (NOUT == 4) {
		const int a = 0x7fffffff; /* saturates to 255 */
		uint4 v;
		v.x = sat4<12>(p0.r, p0.g, p0.b, a);
		v.y = sat4<12>(p1.r, p1.g, p1.b, a);
		v.z = sat4<12>(p2.r, p2.g, p2.b, a);
		v.w = sat4<12>(p3.r, p3.g, p3.b, a);
		__builtin_nontemporal_store((u4v){v.x, v.y, v.z, v.w}, reinterpret_cast<u4v *>(dst));
	} else {
		const uint32_t q0 = sat4<12>(p0.r, p0.g, p0.b, p1.r);
		const uint32_t q1 = sat4<12>(p1.g, p1.b, p2.r, p2.g);
		const uint32_t q2 = sat4<12>(p2.b, p3.r, p3.g, p3.b);
		__builtin_nontemporal_store((u3v){q0, q1, q2}, reinterpret_cast<u3v *>(dst));
	}
}

/* one output row of a 4-pixel strip: chroma operands V0..V2 (see ColorK), luma dword yv */
template <int NOUT>
__device__ __forceinline__ void strip_row(const ColorK &K, uint32_t wk, uint32_t wk1, uint32_t vb0, uint32_t vb1, uint32_t vb2, uint32_t vr0, uint32_t vr1,
														uint32_t vr2, uint32_t yv, uint8_t *__restrict__ dst)
{
	/* weights x16 so that the filtered sample is byte 1 of the dot product:
	 * (3*(3n+f) + (3n'+f') + 8) >> 4  ==  (16*(9n+3f+3n'+f') + 128) >> 8   (codec/jpeg.c:1826-1835)
	 * pixel x0 uses chroma columns (i0, i0-1), x0+1: (i0, i0+1), x0+2: (i0+1, i0), x0+3: (i0+1, i0+2) */
	const uint32_t cb0 = dot4(vb0, wk1, K.w128), cb1 = dot4(vb1, wk, K.w128), cb2 = dot4(vb1, wk1, K.w128), cb3 = dot4(vb2, wk, K.w128);
	const uint32_t cr0 = dot4(vr0, wk1, K.w128), cr1 = dot4(vr1, wk, K.w128), cr2 = dot4(vr1, wk1, K.w128), cr3 = dot4(vr2, wk, K.w128);
	const Rgb12 p0 = color_px(K, __builtin_amdgcn_perm(cr0, yv, K.p0), __builtin_amdgcn_perm(cb0, yv, K.p0));
	const Rgb12 p1 = color_px(K, __builtin_amdgcn_perm(cr1, yv, K.p1), __builtin_amdgcn_perm(cb1, yv, K.p1));
	const Rgb12 p2 = color_px(K, __builtin_amdgcn_perm(cr2, yv, K.p2), __builtin_amdgcn_perm(cb2, yv, K.p2));
	const Rgb12 p3 = color_px(K, __builtin_amdgcn_perm(cr3, yv, K.p3), __builtin_amdgcn_perm(cb3, yv, K.p3));
	store_px4<NOUT>(dst, p0, p1, p2, p3);
}

/* (3 n + f + 2) >> 2 on the four bytes of n and f (even and odd bytes in 16-bit lanes: 3*255 + 255 + 2 < 2^16) */
__device__ __forceinline__ uint32_t rs_v2(uint32_t n, uint32_t f)
{
	const uint32_t m = 0x00ff00ffu;
	const uint32_t te = (n & m) * 3u + (f & m) + 0x00020002u, to = ((n >> 8) & m) * 3u + ((f >> 8) & m) + 0x00020002u;
	return ((te >> 2) & m) | (((to >> 2) & m) << 8);
}

/* one output row of a 4-pixel strip whose chroma is already up-sampled and packed: four samples per dword */
template <int NOUT>
__device__ __forceinline__ void strip_row_packed(const ColorK &K, uint32_t cb4, uint32_t cr4, uint32_t yv, uint8_t *__restrict__ dst)
{
	Rgb12 p[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const uint32_t sel = 0x0c000c04u + (uint32_t)j * 0x00010001u; /* (chroma byte j | luma byte j << 16) */
		p[j] = color_px(K, __builtin_amdgcn_perm(cr4, yv, sel), __builtin_amdgcn_perm(cb4, yv, sel));
	}
	store_px4<NOUT>(dst, p[0], p[1], p[2], p[3]);
}

/* threads per workgroup of k_fused420 (A/B knob: 256 = three workgroups of four waves per CU at 1080p, 512 = two of eight) */
#ifndef MIJ_F420_NT
#define MIJ_F420_NT 256
#endif
#ifndef MIJ_F420_ATTR
#define MIJ_F420_ATTR
#endif
#ifndef MIJ_F420_WAVES /* waves per SIMD the kernel's register count allows (the band planner counts co-resident workgroups with it) */
#define MIJ_F420_WAVES 4
#endif
/* H2 = false: the same band kernel for h1v2 (4:4:0) files -- MCU = 8 x 16 pixels, two luma blocks one above the other, chroma at
 * full width and half height (resample_row_v_2, codec/jpeg.c:1774-1782).  Vertically nothing changes (row pairs (2C-1, 2C) on chroma
 * rows C-1 and C, saved rows, halo block rows at the band edges); horizontally there is no neighbourhood: a strip's chroma is one
 * dword per row and (3 near + far + 2) >> 2 runs on four samples at once in 16-bit lanes (rs_v2). */
/* SEG = true: the workgroup emits MCU columns [xm0, xm1) only -- pictures whose row of MCUs does not fit the LDS of a CU (4:2:0 beyond
 * 5840 pixels, 4:4:0 beyond 4300) are cut into column segments.  A segment transforms one MCU column more on either side (the horizontal
 * chroma filter reads c[i-1] and c[i+1], codec/jpeg.c:1784-1835); everything inside the kernel is in LOCAL columns (the segment plus its
 * halo), and only the plane addresses, the output columns and the picture-edge tests are global.  SEG = false compiles to the code it was. */
template <int NOUT, bool WIDE, bool B8, bool H2, int NT, bool SEG = false>
__device__ __forceinline__ void fused_band(const DevImage *__restrict__ imgs, const WorkBand *__restrict__ work, const uint8_t *__restrict__ coef,
														 uint8_t *__restrict__ outbase)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const WorkBand wk = work[blockIdx.x];
	const DevImage &im = imgs[wk.img];
	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;

	const int gmx = im.mcu_x; /* MCU columns of the picture */
	const int xm0 = SEG ? (int)(wk.cols & 0xffffu) : 0, xm1 = SEG ? (int)(wk.cols >> 16) : gmx; /* columns emitted */
	const int lm0 = SEG ? max(xm0 - 1, 0) : 0, lm1 = SEG ? min(xm1 + 1, gmx) : gmx;              /* columns transformed */
	const int mcu_x = lm1 - lm0;
	const int W = im.width, H = im.height;
	const int YP = (H2 ? 16 : 8) * mcu_x, CP = 8 * mcu_x;
	const int wc = H2 ? (W + 1) >> 1 : W; /* w_lores of the chroma planes, codec/jpeg.c:2276 */
	const int hc = im.comp[1].y;    /* effective chroma rows */
	const int bwY = (H2 ? 2 : 1) * mcu_x, bwC = mcu_x;        /* blocks per row held in LDS */
	const int gbwY = (H2 ? 2 : 1) * gmx, gbwC = gmx;          /* blocks per row of the planes */
	const int bx0Y = (H2 ? 2 : 1) * lm0, bx0C = lm0;          /* the planes' block column of local column 0 */
	const int spm = H2 ? 4 : 2;                               /* 4-pixel strips per MCU column */
	const int soff = spm * lm0;                               /* global strip index of local strip 0 */

	uint8_t *const sY = lds;
	uint8_t *const sCb = sY + 16 * YP;
	uint8_t *const sCr = sCb + 8 * CP;
	uint8_t *const saveY = sCr + 8 * CP;  /* [2][YP] */
	uint8_t *const saveCb = saveY + 2 * YP; /* [2][CP] */
	uint8_t *const saveCr = saveCb + 2 * CP;

	const CoefView cvY = coef_view(coef, im.comp[0]), cvCb = coef_view(coef, im.comp[1]), cvCr = coef_view(coef, im.comp[2]);
	uint8_t *const out = outbase + im.out_off;
	const uint32_t opitch = (uint32_t)W * NOUT;

	const int m0 = (int)wk.m0, m1 = (int)wk.m1;
	const int row_lo = 16 * m0, row_hi = min(16 * m1, H); /* rows this band emits */
	const int nYw = (2 * bwY + 63) >> 6, nCw = (mcu_x + 63) >> 6;
	const int nstrip = (W + 3) >> 2;
	/* fast strips need dword-aligned rows (RGB: W % 4 == 0) and an output that fits 32-bit offsets */
	const bool aligned = ((NOUT == 4) || ((W & 3) == 0)) && ((uint64_t)opitch * (uint32_t)H < 0xfffffff0ull);
	int sv = 0; /* which save buffer holds the previous MCU row's last rows */

	IdctK KI;
	KI.init();
	ColorK KC;
	KC.init();
	const int count_classes = im.flags & MIJ_DEV_COUNT_CLASSES;

	/* ---- chroma-only IDCT of block row mc, keeping sample row 7 (keep != 0) or 0 in dstCb/dstCr (halo rows) */
	auto chroma_halo = [&](int mc, int keep, uint8_t *dstCb, uint8_t *dstCr) {
		for (int ww = wave; ww < 2 * nCw; ww += NT / 64) {
			const int comp = ww < nCw ? 1 : 2;
			const int bx = (comp == 1 ? ww : ww - nCw) * 64 + lane;
			if (bx < bwC) {
				uint2 rows[8];
				load_idct_block<WIDE, B8>(KI, comp == 1 ? cvCb : cvCr, (uint32_t)(mc * gbwC + bx0C + bx), im.dq[comp], rows, 0);
				*reinterpret_cast<uint2 *>((comp == 1 ? dstCb : dstCr) + 8 * bx) = keep ? rows[7] : rows[0];
			}
		}
	};

	/* ---- phase B for chroma row C: emits output rows 2C-1 and 2C (those inside the band) */
	auto emit_pair = [&](int C, const uint8_t *cbA, const uint8_t *cbB, const uint8_t *crA, const uint8_t *crB, const uint8_t *yA, const uint8_t *yB) {
		const int ra = 2 * C - 1, rb = 2 * C;
		const bool doA = ra >= row_lo && ra < row_hi, doB = rb >= row_lo && rb < row_hi;
		if (!doA && !doB)
			return;
		const uint32_t offA = (uint32_t)ra * opitch, offB = (uint32_t)rb * opitch;
		/* fast strips: 4 whole pixels, dword-aligned rows.  Two strips per thread and iteration, all LDS
		 * reads of both issued before either is processed (hides LDS latency, halves the loop overhead) */
		const int nfast = aligned ? (W >> 2) : 0;
		const int s_lo = SEG ? spm * xm0 : 0, s_hi = SEG ? min(spm * xm1, nfast) : nfast; /* the fast strips of this workgroup (global indices) */
		struct StripIn {
			uint32_t bA0, bA1, bB0, bB1, rA0, rA1, rB0, rB1, yA, yB;
		};
		auto load_strip = [&](int sg, StripIn &in) {
			const int s = sg - soff; /* the strip inside the LDS rows */
			if (!H2) { /* 4:4:0: the strip's four chroma samples are one dword of each row */
				in.bA0 = reinterpret_cast<const uint32_t *>(cbA)[s];
				in.bB0 = reinterpret_cast<const uint32_t *>(cbB)[s];
				in.rA0 = reinterpret_cast<const uint32_t *>(crA)[s];
				in.rB0 = reinterpret_cast<const uint32_t *>(crB)[s];
				in.bA1 = in.bB1 = in.rA1 = in.rB1 = 0;
				in.yA = *reinterpret_cast<const uint32_t *>(yA + 4 * s);
				in.yB = *reinterpret_cast<const uint32_t *>(yB + 4 * s);
				return;
			}
			const int d0 = (2 * s - 1) >> 2; /* strip 0 reads the dword in front of the row (inside LDS); the edge fix discards it */
			const uint32_t *pbA = reinterpret_cast<const uint32_t *>(cbA) + d0, *pbB = reinterpret_cast<const uint32_t *>(cbB) + d0;
			const uint32_t *prA = reinterpret_cast<const uint32_t *>(crA) + d0, *prB = reinterpret_cast<const uint32_t *>(crB) + d0;
			in.bA0 = pbA[0];
			in.bA1 = pbA[1];
			in.bB0 = pbB[0];
			in.bB1 = pbB[1];
			in.rA0 = prA[0];
			in.rA1 = prA[1];
			in.rB0 = prB[0];
			in.rB1 = prB[1];
			in.yA = *reinterpret_cast<const uint32_t *>(yA + 4 * s);
			in.yB = *reinterpret_cast<const uint32_t *>(yB + 4 * s);
		};
		auto do_strip = [&](int s, const StripIn &in) {
			if (!H2) {
				const uint32_t xo = (uint32_t)(4 * s) * NOUT;
				if (doB)
					strip_row_packed<NOUT>(KC, rs_v2(in.bB0, in.bA0), rs_v2(in.rB0, in.rA0), in.yB, out + (offB + xo));
				if (doA)
					strip_row_packed<NOUT>(KC, rs_v2(in.bA0, in.bB0), rs_v2(in.rA0, in.rB0), in.yA, out + (offA + xo));
				return;
			}
			const int x0 = 4 * s, i0 = 2 * s;
			/* bytes (c[i0-1], c[i0], c[i0+1], c[i0+2]) of each chroma row: two dwords + a byte funnel shift */
			const uint32_t sh = (uint32_t)(i0 - 1) & 3u;
			uint32_t bA = __builtin_amdgcn_alignbyte(in.bA1, in.bA0, sh);
			uint32_t bB = __builtin_amdgcn_alignbyte(in.bB1, in.bB0, sh);
			uint32_t rA = __builtin_amdgcn_alignbyte(in.rA1, in.rA0, sh);
			uint32_t rB = __builtin_amdgcn_alignbyte(in.rB1, in.rB0, sh);
			/* image edges: column -1 -> 0 and column wc -> wc-1 (the reference's (t+2)>>2 end cases are
			 * the general form with the neighbour clamped, codec/jpeg.c:1820-1835) */
			if (i0 == 0 || i0 + 2 > wc - 1) {
				uint32_t sel = 0x03020100u;
				if (i0 == 0)
					sel = (sel & 0xffffff00u) | 0x01u;
				if (i0 + 2 > wc - 1)
					sel = (sel & 0x00ffffffu) | 0x02000000u;
				bA = __builtin_amdgcn_perm(0, bA, sel);
				bB = __builtin_amdgcn_perm(0, bB, sel);
				rA = __builtin_amdgcn_perm(0, rA, sel);
				rB = __builtin_amdgcn_perm(0, rB, sel);
			}
			const uint32_t vb0 = __builtin_amdgcn_perm(bA, bB, KC.v0), vb1 = __builtin_amdgcn_perm(bA, bB, KC.v1), vb2 = __builtin_amdgcn_perm(bA, bB, KC.v2);
			const uint32_t vr0 = __builtin_amdgcn_perm(rA, rB, KC.v0), vr1 = __builtin_amdgcn_perm(rA, rB, KC.v1), vr2 = __builtin_amdgcn_perm(rA, rB, KC.v2);
			const uint32_t xo = (uint32_t)x0 * NOUT;
			if (doB)
				strip_row<NOUT>(KC, KC.wBk, KC.wBk1, vb0, vb1, vb2, vr0, vr1, vr2, in.yB, out + (offB + xo));
			if (doA)
				strip_row<NOUT>(KC, KC.wAk, KC.wAk1, vb0, vb1, vb2, vr0, vr1, vr2, in.yA, out + (offA + xo));
		};
		if (NT >= 512) { /* one strip per thread and iteration: a 1080p row is 480 strips */
			for (int sa = s_lo + tid; sa < s_hi; sa += NT) {
				StripIn ia;
				load_strip(sa, ia);
				do_strip(sa, ia);
			}
		} else
		for (int base = s_lo; base < s_hi; base += 2 * NT) {
			const int sa = base + tid, sb = sa + NT;
			StripIn ia, ib;
			load_strip(min(sa, s_hi - 1), ia);
			load_strip(min(sb, s_hi - 1), ib);
			if (sa < s_hi)
				do_strip(sa, ia);
			if (sb < s_hi)
				do_strip(sb, ib);
		}
		/* the rest (partial last strip, unaligned widths): careful per-pixel path, on row pointers moved back to the picture's column 0 */
		const int ypx0 = (H2 ? 16 : 8) * lm0, cpx0 = 8 * lm0;
		for (int s = (SEG ? max(nfast, s_lo) : nfast) + tid; s < (SEG ? min(nstrip, spm * xm1) : nstrip); s += NT) {
			const int x0 = 4 * s, xe = min(x0 + 4, W);
			for (int x = x0; x < xe; ++x) {
				if (doA)
					fused420_pixel<NOUT, H2>(yA - ypx0, cbA - cpx0, cbB - cpx0, crA - cpx0, crB - cpx0, 0, wc, x, out + (size_t)ra * opitch);
				if (doB)
					fused420_pixel<NOUT, H2>(yB - ypx0, cbA - cpx0, cbB - cpx0, crA - cpx0, crB - cpx0, 1, wc, x, out + (size_t)rb * opitch);
			}
		}
	};

	/* ---- prologue: chroma row 8*m0-1 from the block row above the band */
	if (m0 > 0) {
		chroma_halo(m0 - 1, 1, saveCb + sv * CP, saveCr + sv * CP);
	}

	for (int m = m0; m < m1; ++m) {
		__syncthreads(); /* previous phase B (and the prologue) done with the planes / save buffers */
		/* ---- phase A: IDCT of MCU row m, one block per lane, component uniform per wave; one call site for the three components
		 * (the sparse-class transforms of load_idct_block are instantiated once per kernel) */
		for (int ww = wave; ww < nYw + 2 * nCw; ww += NT / 64) {
			uint2 rows[8];
			const int comp = ww < nYw ? 0 : ((ww - nYw) < nCw ? 1 : 2); /* wave-uniform */
			const int i = (ww - (comp == 0 ? 0 : (comp == 1 ? nYw : nYw + nCw))) * 64 + lane; /* luma: block of the two block rows 2m, 2m+1 (contiguous in L) */
			const int nblk = comp == 0 ? 2 * bwY : bwC;
			if (i < nblk) {
				const int by = (comp == 0 && i >= bwY) ? 1 : 0, bx = i - by * bwY;
				const int pitch = comp == 0 ? YP : CP;
				const uint32_t L = SEG ? (uint32_t)(comp == 0 ? (2 * m + by) * gbwY + bx0Y + bx : m * gbwC + bx0C + bx) : (uint32_t)((comp == 0 ? 2 * m * bwY : m * bwC) + i);
				load_idct_block<WIDE, B8>(KI, comp == 0 ? cvY : (comp == 1 ? cvCb : cvCr), L, im.dq[comp], rows, count_classes);
				uint8_t *dst = (comp == 0 ? sY : (comp == 1 ? sCb : sCr)) + (8 * by) * pitch + 8 * bx;
#pragma unroll
				for (int r = 0; r < 8; ++r)
					*reinterpret_cast<uint2 *>(dst + r * pitch) = rows[r];
			}
		}
		__syncthreads();

		/* ---- phase B: chroma rows C = 8m .. 8m+7 -> output rows 2C-1, 2C */
		for (int cc = 0; cc < 8; ++cc) {
			const int C = 8 * m + cc;
			/* row C-1: previous plane row, or the saved row, or (image top) row 0 itself */
			const uint8_t *cbA, *crA, *yA;
			if (cc > 0) {
				cbA = sCb + (cc - 1) * CP;
				crA = sCr + (cc - 1) * CP;
				yA = sY + (2 * cc - 1) * YP;
			} else if (C > 0) {
				cbA = saveCb + sv * CP;
				crA = saveCr + sv * CP;
				yA = saveY + sv * YP;
			} else {
				cbA = sCb;
				crA = sCr;
				yA = sY; /* row -1 is never emitted */
			}
			/* row C, clamped to the last effective chroma row (then equal to row C-1) */
			const uint8_t *cbB = C <= hc - 1 ? sCb + cc * CP : cbA;
			const uint8_t *crB = C <= hc - 1 ? sCr + cc * CP : crA;
			emit_pair(C, cbA, cbB, crA, crB, yA, sY + (2 * cc) * YP);
		}
		/* keep the last rows of this MCU row for the next step (other save buffer: no extra barrier) */
		{
			const int nv = sv ^ 1;
			for (int i = tid; i < YP / 4; i += NT)
				reinterpret_cast<uint32_t *>(saveY + nv * YP)[i] = reinterpret_cast<const uint32_t *>(sY + 15 * YP)[i];
			for (int i = tid; i < CP / 4; i += NT) {
				reinterpret_cast<uint32_t *>(saveCb + nv * CP)[i] = reinterpret_cast<const uint32_t *>(sCb + 7 * CP)[i];
				reinterpret_cast<uint32_t *>(saveCr + nv * CP)[i] = reinterpret_cast<const uint32_t *>(sCr + 7 * CP)[i];
			}
			sv = nv;
		}
	}

	/* ---- epilogue: the band's last row 16*m1-1 pairs chroma row 8*m1-1 (saved) with row 8*m1 */
	{
		const int C = 8 * m1, ra = 2 * C - 1;
		if (ra >= row_lo && ra < row_hi) {
			__syncthreads(); /* planes free, save buffers written */
			const uint8_t *cbA = saveCb + sv * CP, *crA = saveCr + sv * CP;
			const uint8_t *cbB = cbA, *crB = crA;
			if (C <= hc - 1) { /* there is a block row below: its first sample row */
				chroma_halo(m1, 0, sCb, sCr);
				__syncthreads();
				cbB = sCb;
				crB = sCr;
			}
			emit_pair(C, cbA, cbB, crA, crB, saveY + sv * YP, saveY + sv * YP);
		}
	}
}

template <int NOUT, bool WIDE, bool B8 = false>
__global__ __launch_bounds__(MIJ_F420_NT) MIJ_F420_ATTR void k_fused420(const DevImage *__restrict__ imgs, const WorkBand *__restrict__ work, const uint8_t *__restrict__ coef,
																  uint8_t *__restrict__ outbase)
{
	fused_band<NOUT, WIDE, B8, true, MIJ_F420_NT>(imgs, work, coef, outbase);
}

/* The same kernel with eight waves per workgroup, for pictures whose row of MCUs leaves room for only one or two workgroups in a CU's
 * LDS (448 bytes per MCU column: wider than about 2100 pixels): measured per 0.53 Gpix of pictures resident in HBM, 256 -> 512 threads:
 * 2560 x 1440 0.799 -> 0.732 ms, 3840 x 2160 0.970 -> 0.809, 5120 x 2880 0.941 -> 0.739; 1920 x 1080 and narrower lose 1-15 % and stay
 * with four waves (tools/bench_sizes.py, profiles/r02zz_band_threads.txt). */
#ifndef MIJ_F420W_NT
#define MIJ_F420W_NT 512
#endif
template <int NOUT, bool WIDE, bool B8 = false>
__global__ __launch_bounds__(MIJ_F420W_NT) void k_fused420w(const DevImage *__restrict__ imgs, const WorkBand *__restrict__ work, const uint8_t *__restrict__ coef,
																				 uint8_t *__restrict__ outbase)
{
	fused_band<NOUT, WIDE, B8, true, MIJ_F420W_NT>(imgs, work, coef, outbase);
}

/* ... and with sixteen waves where only ONE workgroup fits (wider than about 2900 pixels: 3840 x 2160 0.806 -> 0.722 ms, 5120 x 2880
 * 0.739 -> 0.70 against the eight-wave form; 2560 x 1440, where two fit, loses 13 % with it) */
#define MIJ_F420X_NT 1024
template <int NOUT, bool WIDE, bool B8 = false>
__global__ __launch_bounds__(MIJ_F420X_NT) void k_fused420x(const DevImage *__restrict__ imgs, const WorkBand *__restrict__ work, const uint8_t *__restrict__ coef,
																				 uint8_t *__restrict__ outbase)
{
	fused_band<NOUT, WIDE, B8, true, MIJ_F420X_NT>(imgs, work, coef, outbase);
}

/* ... and with two waves / one wave for narrow pictures, whose row of MCUs does not fill four: per 0.53 Gpix resident, 256 / 128 / 64
 * threads: 64 x 64 0.588 / 0.448 / 0.347 ms, 256 x 256 1.18 / 0.93 / 0.76, 512 x 512 0.80 / 0.68 / 0.75, 800 x 600 0.84 / 0.79 / 0.99,
 * 1024 x 768 0.66 / 0.72 / 0.91 (profiles/r02zz_band_threads.txt): one wave up to 24 MCU columns (384 pixels), two up to 56 (896). */
#define MIJ_F420S_NT 128
#define MIJ_F420T_NT 64
template <int NOUT, bool WIDE, bool B8 = false>
__global__ __launch_bounds__(MIJ_F420S_NT) void k_fused420s(const DevImage *__restrict__ imgs, const WorkBand *__restrict__ work, const uint8_t *__restrict__ coef,
																				 uint8_t *__restrict__ outbase)
{
	fused_band<NOUT, WIDE, B8, true, MIJ_F420S_NT>(imgs, work, coef, outbase);
}
template <int NOUT, bool WIDE, bool B8 = false>
__global__ __launch_bounds__(MIJ_F420T_NT) void k_fused420t(const DevImage *__restrict__ imgs, const WorkBand *__restrict__ work, const uint8_t *__restrict__ coef,
																				 uint8_t *__restrict__ outbase)
{
	fused_band<NOUT, WIDE, B8, true, MIJ_F420T_NT>(imgs, work, coef, outbase);
}

/* h1v2 (4:4:0): see fused_band, H2 = false.  LDS 304 * mcu_x bytes: a 1080p row takes 73 KB, two workgroups per CU, and goes through the
 * eight-wave form like the wide 4:2:0 pictures above (0.753 -> 0.654 ms per 256 images, 0.62 -> 0.71 of the roofline). */
template <int NOUT, bool WIDE, bool B8 = false>
__global__ __launch_bounds__(MIJ_F420_NT) void k_fused440(const DevImage *__restrict__ imgs, const WorkBand *__restrict__ work, const uint8_t *__restrict__ coef,
																			 uint8_t *__restrict__ outbase)
{
	fused_band<NOUT, WIDE, B8, false, MIJ_F420_NT>(imgs, work, coef, outbase);
}
template <int NOUT, bool WIDE, bool B8 = false>
__global__ __launch_bounds__(MIJ_F420W_NT) void k_fused440w(const DevImage *__restrict__ imgs, const WorkBand *__restrict__ work, const uint8_t *__restrict__ coef,
																				 uint8_t *__restrict__ outbase)
{
	fused_band<NOUT, WIDE, B8, false, MIJ_F420W_NT>(imgs, work, coef, outbase);
}

/* Column-segmented forms (fused_band, SEG): pictures whose row of MCUs exceeds a CU's LDS.  Eight waves and segments that let two
 * workgroups share a CU (at most 180 MCU columns of 4:2:0), measured per 0.53 Gpix resident against the two-pass kernels these pictures
 * took before (tools/seg_sweep.sh, profiles/r03_wide_segments.txt): 6000 x 4000 0.847 -> 0.707 ms, 8192 x 5464 0.741 -> 0.643; one
 * workgroup of sixteen waves on a segment that fills the LDS: 0.717 / 0.670; three of four waves on 119 columns: 0.829 / 0.687.  Cutting
 * pictures that DO fit (2560 .. 5120 pixels) into segments as well loses 0-9 % against the w / x forms, so only the others are cut. */
#ifndef MIJ_F420C_NT
#define MIJ_F420C_NT 512
#endif
template <int NOUT, bool WIDE, bool B8 = false>
__global__ __launch_bounds__(MIJ_F420C_NT) void k_fused420c(const DevImage *__restrict__ imgs, const WorkBand *__restrict__ work, const uint8_t *__restrict__ coef,
																				 uint8_t *__restrict__ outbase)
{
	fused_band<NOUT, WIDE, B8, true, MIJ_F420C_NT, true>(imgs, work, coef, outbase);
}
template <int NOUT, bool WIDE, bool B8 = false>
__global__ __launch_bounds__(MIJ_F420C_NT) void k_fused440c(const DevImage *__restrict__ imgs, const WorkBand *__restrict__ work, const uint8_t *__restrict__ coef,
																				 uint8_t *__restrict__ outbase)
{
	fused_band<NOUT, WIDE, B8, false, MIJ_F420C_NT, true>(imgs, work, coef, outbase);
}

/* ------------------------------------------------------------------ fused h2v1 (4:2:2) YCbCr kernel
 *
 * MCU = 16x8 pixels: two luma blocks side by side, one Cb, one Cr block.  Chroma is only stretched
 * horizontally (resample_row_h_2, codec/jpeg.c:1784-1812), so there is no vertical neighbourhood: no halo,
 * no saved rows, any range of MCU rows is an independent band.  Same two phases as k_fused420 on a smaller
 * LDS footprint (Y[8][16*mcu_x], Cb/Cr[8][8*mcu_x] = 256*mcu_x bytes).
 * h_2 in one v_dot4_u32_u8 per sample: (3c[i] + c[i+-1] + 2) >> 2 == (64*(3c[i] + c[i+-1]) + 128) >> 8 on the
 * byte window (c[i0-1], c[i0], c[i0+1], c[i0+2]) of a 4-pixel strip; the first / last columns are the same form
 * with the neighbour clamped, and the reference's last-but-one column (:1805: 3*in[w-2] + in[w-1]) is the
 * odd-pixel weight vector used once more.
 */
__device__ __forceinline__ size_t fused422_lds_bytes(int mcu_x) { return (size_t)mcu_x * 256 + 16; } /* + the dword a last strip reads past the last row */

/* careful per-pixel path for widths that are not a multiple of 4 */
template <int NOUT>
__device__ __forceinline__ void fused422_pixel(const uint8_t *yrow, const uint8_t *cb, const uint8_t *cr, int wc, int x, uint8_t *__restrict__ dst)
{
	const int i = x >> 1;
	int vb, vr;
	if (wc == 1 || x == 0) {
		vb = cb[0];
		vr = cr[0];
	} else if (x == 2 * wc - 1) {
		vb = cb[wc - 1];
		vr = cr[wc - 1];
	} else if (x == 2 * wc - 2) {
		vb = (3 * cb[wc - 2] + cb[wc - 1] + 2) >> 2;
		vr = (3 * cr[wc - 2] + cr[wc - 1] + 2) >> 2;
	} else {
		const int j = (x & 1) ? i + 1 : i - 1;
		vb = (3 * cb[i] + cb[j] + 2) >> 2;
		vr = (3 * cr[i] + cr[j] + 2) >> 2;
	}
	int r, g, b;
	ycbcr_to_rgb(yrow[x], vb, vr, r, g, b);
	store_rgb_px<NOUT>(dst + (size_t)x * NOUT, r, g, b);
}

template <int NOUT, bool WIDE, bool B8, int NT>
__device__ __forceinline__ void fused422_band(const DevImage *__restrict__ imgs, const WorkBand *__restrict__ work, const uint8_t *__restrict__ coef,
																  uint8_t *__restrict__ outbase)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const WorkBand wk = work[blockIdx.x];
	const DevImage &im = imgs[wk.img];
	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	const int mcu_x = im.mcu_x, W = im.width, H = im.height;
	const int YP = 16 * mcu_x, CP = 8 * mcu_x;
	const int wc = (W + 1) >> 1;
	uint8_t *const sY = lds;
	uint8_t *const sCb = sY + 8 * YP;
	uint8_t *const sCr = sCb + 8 * CP;
	const CoefView cvY = coef_view(coef, im.comp[0]), cvCb = coef_view(coef, im.comp[1]), cvCr = coef_view(coef, im.comp[2]);
	uint8_t *const out = outbase + im.out_off;
	const uint32_t opitch = (uint32_t)W * NOUT;
	const int bwY = 2 * mcu_x, bwC = mcu_x;
	const int nYw = (bwY + 63) >> 6, nCw = (bwC + 63) >> 6;
	const int nstrip = (W + 3) >> 2;
	const bool aligned = ((NOUT == 4) || ((W & 3) == 0)) && ((uint64_t)opitch * (uint32_t)H < 0xfffffff0ull);
	const int nfast = aligned ? (W >> 2) : 0;

	IdctK KI;
	KI.init();
	ColorK KC;
	KC.init();
	const uint32_t wE0 = vreg(0x0000c040u); /* pixel x0   : 64*c[i0-1] + 192*c[i0]   */
	const uint32_t wO0 = vreg(0x0040c000u); /* pixel x0+1 : 192*c[i0]  + 64*c[i0+1]  */
	const uint32_t wE1 = vreg(0x00c04000u); /* pixel x0+2 : 64*c[i0]   + 192*c[i0+1] */
	const uint32_t wO1 = vreg(0x40c00000u); /* pixel x0+3 : 192*c[i0+1] + 64*c[i0+2] */

	for (int m = (int)wk.m0; m < (int)wk.m1; ++m) {
		__syncthreads(); /* previous phase B done with the planes */
		/* ---- phase A: IDCT of MCU row m, one block per lane, component uniform per wave (one call site, see fused_band) */
		for (int ww = wave; ww < nYw + 2 * nCw; ww += NT / 64) {
			uint2 rows[8];
			const int comp = ww < nYw ? 0 : ((ww - nYw) < nCw ? 1 : 2);
			const int bx = (ww - (comp == 0 ? 0 : (comp == 1 ? nYw : nYw + nCw))) * 64 + lane;
			const int nb = comp == 0 ? bwY : bwC, pitch = comp == 0 ? YP : CP;
			if (bx < nb) {
				load_idct_block<WIDE, B8>(KI, comp == 0 ? cvY : (comp == 1 ? cvCb : cvCr), (uint32_t)(m * nb + bx), im.dq[comp], rows, im.flags & MIJ_DEV_COUNT_CLASSES);
				uint8_t *dst = (comp == 0 ? sY : (comp == 1 ? sCb : sCr)) + 8 * bx;
#pragma unroll
				for (int r = 0; r < 8; ++r)
					*reinterpret_cast<uint2 *>(dst + r * pitch) = rows[r];
			}
		}
		__syncthreads();

		/* ---- phase B: the 8 pixel rows of the MCU row, 4-pixel strips */
		const int rows_here = min(8, H - 8 * m);
		for (int rr = 0; rr < rows_here; ++rr) {
			const uint8_t *cbR = sCb + rr * CP, *crR = sCr + rr * CP, *yR = sY + rr * YP;
			const uint32_t rowoff = (uint32_t)(8 * m + rr) * opitch;
			for (int s0 = tid; s0 < nfast; s0 += NT) {
				const int i0 = 2 * s0;
				const int d0 = (i0 - 1) >> 2; /* strip 0 reads the dword in front of the row (inside LDS); the edge fix discards it */
				const uint32_t *pb = reinterpret_cast<const uint32_t *>(cbR) + d0, *pr = reinterpret_cast<const uint32_t *>(crR) + d0;
				const uint32_t b0 = pb[0], b1 = pb[1], r0 = pr[0], r1 = pr[1];
				const uint32_t yv = *reinterpret_cast<const uint32_t *>(yR + 4 * s0);
				const uint32_t sh = (uint32_t)(i0 - 1) & 3u;
				uint32_t vb = __builtin_amdgcn_alignbyte(b1, b0, sh), vr = __builtin_amdgcn_alignbyte(r1, r0, sh);
				uint32_t w2 = wE1;
				if (i0 == 0 || i0 + 2 > wc - 1) {
					uint32_t sel = 0x03020100u;
					if (i0 == 0)
						sel = (sel & 0xffffff00u) | 0x01u; /* column -1 -> 0 */
					if (i0 + 2 > wc - 1) {
						sel = (sel & 0x00ffffffu) | 0x02000000u; /* column wc -> wc-1 */
						if (wc > 1)
							w2 = wO0; /* pixel 2*(wc-1): 3*c[wc-2] + c[wc-1], codec/jpeg.c:1805 */
					}
					vb = __builtin_amdgcn_perm(0, vb, sel);
					vr = __builtin_amdgcn_perm(0, vr, sel);
				}
				const uint32_t cb0 = dot4(vb, wE0, KC.w128), cb1 = dot4(vb, wO0, KC.w128), cb2 = dot4(vb, w2, KC.w128), cb3 = dot4(vb, wO1, KC.w128);
				const uint32_t cr0 = dot4(vr, wE0, KC.w128), cr1 = dot4(vr, wO0, KC.w128), cr2 = dot4(vr, w2, KC.w128), cr3 = dot4(vr, wO1, KC.w128);
				const Rgb12 p0 = color_px(KC, __builtin_amdgcn_perm(cr0, yv, KC.p0), __builtin_amdgcn_perm(cb0, yv, KC.p0));
				const Rgb12 p1 = color_px(KC, __builtin_amdgcn_perm(cr1, yv, KC.p1), __builtin_amdgcn_perm(cb1, yv, KC.p1));
				const Rgb12 p2 = color_px(KC, __builtin_amdgcn_perm(cr2, yv, KC.p2), __builtin_amdgcn_perm(cb2, yv, KC.p2));
				const Rgb12 p3 = color_px(KC, __builtin_amdgcn_perm(cr3, yv, KC.p3), __builtin_amdgcn_perm(cb3, yv, KC.p3));
				store_px4<NOUT>(out + (rowoff + (uint32_t)(4 * s0) * NOUT), p0, p1, p2, p3);
			}
			for (int s0 = nfast + tid; s0 < nstrip; s0 += NT) {
				const int x0 = 4 * s0, xe = min(x0 + 4, W);
				for (int x = x0; x < xe; ++x)
					fused422_pixel<NOUT>(yR, cbR, crR, wc, x, out + (size_t)(8 * m + rr) * opitch);
			}
		}
	}
}

/* 256 threads where three or more workgroups of the picture's width fit a CU's LDS (256 bytes per MCU column: up to 3400 pixels), 512
 * where two do (up to 5100), 1024 where one does -- the same ladder as the 4:2:0 band kernel (k_fused420w / x) */
#define MIJ_BAND422(NAME, NT_)                                                                                                      \
	template <int NOUT, bool WIDE, bool B8 = false>                                                                                  \
	__global__ __launch_bounds__(NT_) void NAME(const DevImage *__restrict__ imgs, const WorkBand *__restrict__ work, const uint8_t *__restrict__ coef, \
															  uint8_t *__restrict__ outbase)                                                              \
	{                                                                                                                                \
		fused422_band<NOUT, WIDE, B8, NT_>(imgs, work, coef, outbase);                                                                \
	}
MIJ_BAND422(k_fused422, 256)
MIJ_BAND422(k_fused422w, 512)
MIJ_BAND422(k_fused422x, 1024)
MIJ_BAND422(k_fused422s, 128) /* narrow pictures, as k_fused420s / t */
MIJ_BAND422(k_fused422t, 64)
#undef MIJ_BAND422

/* ------------------------------------------------------------------ fused 1x1 (4:4:4) YCbCr kernel
 *
 * No sub-sampling means no neighbourhood: one lane owns one 8x8 MCU -- three blocks, one per
 * component, same block index L in each plane -- transforms them back to back and colour-converts
 * its 64 pixels entirely in registers.  No LDS, no barriers, no halo.  Reads are the coalesced
 * tile-layout chunks (3 x 128 B per lane); each output row of the MCU is 8 pixels = 24 (RGB) or 32
 * (RGBA) contiguous bytes per lane.  Algorithmic bytes: 384 B read + 64*NOUT written per MCU
 * (9 B/px for RGB: BASELINE config 4's shape).
 */
/* One output row of a lane's 8 x 8 block -- 8 pixels = 8 * NOUT bytes in q[0 .. 2*NOUT-1] -- to the picture.
 * direct: two stores per lane (12 + 12 or 16 + 16 bytes): every 64-byte sector of the row is written in pieces by two instructions, which the
 * counters showed as 1.21x the pixels' bytes in write traffic (round 2, profiles/r02e).
 * via_lds (wave-uniform; the wave's 64 blocks are neighbours in ONE block row and lie wholly inside the picture): the 64 lanes' rows are one
 * contiguous run of 64 * 8 * NOUT bytes; they go through a per-wave LDS row and leave as whole 16-byte chunks in address order, 1 KiB per store
 * instruction, so a sector is written once. */
template <int NOUT>
__device__ __forceinline__ void store_row8(const uint32_t (&q)[2 * NOUT], bool via_lds, uint8_t *lds_row, int lane, uint8_t *__restrict__ dst_lane, uint8_t *__restrict__ dst_wave)
{
	if (via_lds) {
		uint32_t *w = reinterpret_cast<uint32_t *>(lds_row + lane * (8 * NOUT));
		if constexpr (NOUT == 3) {
			*reinterpret_cast<uint2 *>(w) = make_uint2(q[0], q[1]);
			*reinterpret_cast<uint2 *>(w + 2) = make_uint2(q[2], q[3]);
			*reinterpret_cast<uint2 *>(w + 4) = make_uint2(q[4], q[5]);
		} else {
			*reinterpret_cast<uint4 *>(w) = make_uint4(q[0], q[1], q[2], q[3]);
			*reinterpret_cast<uint4 *>(w + 4) = make_uint4(q[4], q[5], q[6], q[7]);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		const uint4 c0 = *reinterpret_cast<const uint4 *>(lds_row + lane * 16);
		__builtin_nontemporal_store((u4v){c0.x, c0.y, c0.z, c0.w}, reinterpret_cast<u4v *>(dst_wave + lane * 16));
		if (NOUT == 4 || lane < 32) {
			const uint4 c1 = *reinterpret_cast<const uint4 *>(lds_row + 1024 + lane * 16);
			__builtin_nontemporal_store((u4v){c1.x, c1.y, c1.z, c1.w}, reinterpret_cast<u4v *>(dst_wave + 1024 + lane * 16));
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier(); /* the row buffer is free again */
	} else if constexpr (NOUT == 3) {
		__builtin_nontemporal_store((u3v){q[0], q[1], q[2]}, reinterpret_cast<u3v *>(dst_lane));
		__builtin_nontemporal_store((u3v){q[3], q[4], q[5]}, reinterpret_cast<u3v *>(dst_lane + 12));
	} else {
		__builtin_nontemporal_store((u4v){q[0], q[1], q[2], q[3]}, reinterpret_cast<u4v *>(dst_lane));
		__builtin_nontemporal_store((u4v){q[4], q[5], q[6], q[7]}, reinterpret_cast<u4v *>(dst_lane + 16));
	}
}

/* four pixels' colour sums -> NOUT packed dwords (see store_px4) */
template <int NOUT>
__device__ __forceinline__ void pack_px4(const Rgb12 &p0, const Rgb12 &p1, const Rgb12 &p2, const Rgb12 &p3, uint32_t *q)
{
	if (NOUT == 4) {
		const int a = 0x7fffffff; /* saturates to 255 */
		q[0] = sat4<12>(p0.r, p0.g, p0.b, a);
		q[1] = sat4<12>(p1.r, p1.g, p1.b, a);
		q[2] = sat4<12>(p2.r, p2.g, p2.b, a);
		q[3] = sat4<12>(p3.r, p3.g, p3.b, a);
	} else {
		q[0] = sat4<12>(p0.r, p0.g, p0.b, p1.r);
		q[1] = sat4<12>(p1.g, p1.b, p2.r, p2.g);
		q[2] = sat4<12>(p2.b, p3.r, p3.g, p3.b);
	}
}

/* what a lane of the 1x1 kernels needs to know about where its block's pixels go */
struct Block1x1 {
	int x0, y0;
	bool whole, via_lds;
	uint8_t *lds_row;
};
template <int NOUT>
__device__ __forceinline__ Block1x1 block_1x1_setup(const DevImage &im, uint32_t L, uint32_t nblk, uint8_t *lds_rows)
{
	Block1x1 g;
	const uint32_t bw = (uint32_t)im.comp[0].bw;
	const uint32_t by = L / bw, bx = L - by * bw;
	g.x0 = (int)bx * 8;
	g.y0 = (int)by * 8;
	g.whole = (g.x0 + 8 <= im.width) && ((NOUT == 4) || ((im.width & 3) == 0));
	/* LDS-transposed stores: every lane of the wave active and whole, all in one block row (lane 0's block row == lane 63's) */
	const int lane = (int)(threadIdx.x & 63u);
	const uint32_t L0 = L - (uint32_t)lane;
	const bool one_row = (L0 / bw) == ((L0 + 63u) / bw) && L0 + 63u < nblk;
	g.via_lds = one_row && __builtin_amdgcn_ballot_w64(g.whole) == ~0ull;
	g.lds_row = lds_rows + (threadIdx.x >> 6) * (64 * 8 * NOUT);
	return g;
}

template <int NOUT, bool WIDE, bool B8 = false>
__global__ __launch_bounds__(256) void k_fused444(const DevImage *__restrict__ imgs, const WorkIdct *__restrict__ work, const uint8_t *__restrict__ coef,
																  uint8_t *__restrict__ outbase)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds_rows[4 * 64 * 8 * NOUT];
	const WorkIdct wk = work[blockIdx.x];
	const DevImage &im = imgs[wk.img];
	const int bw = im.comp[0].bw;
	const uint32_t nblk = (uint32_t)(bw * im.comp[0].bh);
	const uint32_t L = wk.first + threadIdx.x;
	if (L >= nblk)
		return;
	const int W = im.width, H = im.height;
	IdctK KI;
	KI.init();
	uint2 ry[8], rb[8], rr[8];
	/* No sparse-block classes here (round 3, measured on 32 x 4096^2 4:4:4 whose chroma is almost all DC-only / 2x2, tools/ab3.sh): this
	 * kernel sits on the HBM roofline, not on instruction issue -- with the class test between the three transforms it ran 0.67 ms, with all
	 * three blocks' loads issued first 0.74 ms, without any of it 0.61 ms.  What would help here is not reading sparse blocks at all. */
	{
		uint4 c[8];
		load_block_fmt<B8>(coef_view(coef, im.comp[0]), L, im.dq[0], c);
		idct_block<WIDE, B8>(KI, c, im.dq[0], ry);
	}
	{
		uint4 c[8];
		load_block_fmt<B8>(coef_view(coef, im.comp[1]), L, im.dq[1], c);
		idct_block<WIDE, B8>(KI, c, im.dq[1], rb);
	}
	{
		uint4 c[8];
		load_block_fmt<B8>(coef_view(coef, im.comp[2]), L, im.dq[2], c);
		idct_block<WIDE, B8>(KI, c, im.dq[2], rr);
	}
	const Block1x1 g = block_1x1_setup<NOUT>(im, L, nblk, lds_rows);
	const int x0 = g.x0, y0 = g.y0, lane = (int)(threadIdx.x & 63u);
	uint8_t *const out = outbase + im.out_off;
	const size_t opitch = (size_t)W * NOUT;
	ColorK KC;
	KC.init();
	/* (chroma byte j | luma byte j << 16) */
	const uint32_t s0 = vreg(0x0c000c04u), s1 = vreg(0x0c010c05u), s2 = vreg(0x0c020c06u), s3 = vreg(0x0c030c07u);
#pragma unroll
	for (int r = 0; r < 8; ++r) {
		if (y0 + r >= H)
			break;
		uint8_t *dst = out + (size_t)(y0 + r) * opitch + (size_t)x0 * NOUT;
		const uint32_t ylo = ry[r].x, yhi = ry[r].y, blo = rb[r].x, bhi = rb[r].y, rlo = rr[r].x, rhi = rr[r].y;
		const Rgb12 p0 = color_px(KC, __builtin_amdgcn_perm(rlo, ylo, s0), __builtin_amdgcn_perm(blo, ylo, s0));
		const Rgb12 p1 = color_px(KC, __builtin_amdgcn_perm(rlo, ylo, s1), __builtin_amdgcn_perm(blo, ylo, s1));
		const Rgb12 p2 = color_px(KC, __builtin_amdgcn_perm(rlo, ylo, s2), __builtin_amdgcn_perm(blo, ylo, s2));
		const Rgb12 p3 = color_px(KC, __builtin_amdgcn_perm(rlo, ylo, s3), __builtin_amdgcn_perm(blo, ylo, s3));
		const Rgb12 p4 = color_px(KC, __builtin_amdgcn_perm(rhi, yhi, s0), __builtin_amdgcn_perm(bhi, yhi, s0));
		const Rgb12 p5 = color_px(KC, __builtin_amdgcn_perm(rhi, yhi, s1), __builtin_amdgcn_perm(bhi, yhi, s1));
		const Rgb12 p6 = color_px(KC, __builtin_amdgcn_perm(rhi, yhi, s2), __builtin_amdgcn_perm(bhi, yhi, s2));
		const Rgb12 p7 = color_px(KC, __builtin_amdgcn_perm(rhi, yhi, s3), __builtin_amdgcn_perm(bhi, yhi, s3));
		if (g.whole) {
			uint32_t q[2 * NOUT];
			pack_px4<NOUT>(p0, p1, p2, p3, q);
			pack_px4<NOUT>(p4, p5, p6, p7, q + NOUT);
			store_row8<NOUT>(q, g.via_lds, g.lds_row, lane, dst, dst - (size_t)lane * (8 * NOUT));
		} else {
			/* right-edge MCU or a row pitch that is not dword aligned: byte stores of the valid pixels */
			const Rgb12 px[8] = {p0, p1, p2, p3, p4, p5, p6, p7};
#pragma unroll
			for (int j = 0; j < 8; ++j)
				if (x0 + j < W)
					store_rgb_px<NOUT>(dst + j * NOUT, clamp255(opaque(px[j].r >> 12)), clamp255(opaque(px[j].g >> 12)), clamp255(opaque(px[j].b >> 12)));
		}
	}
}

/* ------------------------------------------------------------------ fused kernel for the other 1x1 colour layouts (round 3)
 * RGB-tagged files (three components copied, codec/jpeg.c:2325-2335), Adobe CMYK (:2343-2354) and YCCK (:2355-2366) whose components all
 * have sampling factors 1x1: the same shape as k_fused444 -- a lane owns the 8 x 8 block position, transforms its three or four blocks
 * (with the sparse classes: chroma and K planes of such files are often flat) and converts its 64 pixels in registers -- instead of the
 * two-pass family's round trip of the sample planes through HBM.  The colour mode is wave-uniform (one image per workgroup). */
template <int NOUT, bool WIDE, bool B8 = false>
__global__ __launch_bounds__(256) void k_fused1x1c(const DevImage *__restrict__ imgs, const WorkIdct *__restrict__ work, const uint8_t *__restrict__ coef,
																	uint8_t *__restrict__ outbase)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds_rows[4 * 64 * 8 * NOUT];
	const WorkIdct wk = work[blockIdx.x];
	const DevImage &im = imgs[wk.img];
	const int bw = im.comp[0].bw;
	const uint32_t nblk = (uint32_t)(bw * im.comp[0].bh);
	const uint32_t L = wk.first + threadIdx.x;
	if (L >= nblk)
		return;
	const int W = im.width, H = im.height, color = im.color;
	const bool four = color == MIJ_COLOR_CMYK || color == MIJ_COLOR_YCCK;
	const int count_classes = im.flags & MIJ_DEV_COUNT_CLASSES;
	IdctK KI;
	KI.init();
	uint2 r0[8], r1[8], r2[8], r3[8];
	load_idct_block<WIDE, B8>(KI, coef_view(coef, im.comp[0]), L, im.dq[0], r0, count_classes);
	load_idct_block<WIDE, B8>(KI, coef_view(coef, im.comp[1]), L, im.dq[1], r1, count_classes);
	load_idct_block<WIDE, B8>(KI, coef_view(coef, im.comp[2]), L, im.dq[2], r2, count_classes);
	if (four)
		load_idct_block<WIDE, B8>(KI, coef_view(coef, im.comp[3]), L, im.dq[3], r3, count_classes);
	else {
#pragma unroll
		for (int r = 0; r < 8; ++r)
			r3[r] = make_uint2(0xffffffffu, 0xffffffffu);
	}
	const Block1x1 g = block_1x1_setup<NOUT>(im, L, nblk, lds_rows);
	const int lane = (int)(threadIdx.x & 63u);
	uint8_t *const out = outbase + im.out_off;
	const size_t opitch = (size_t)W * NOUT;
#pragma unroll
	for (int r = 0; r < 8; ++r) {
		if (g.y0 + r >= H)
			break;
		uint8_t *dst = out + (size_t)(g.y0 + r) * opitch + (size_t)g.x0 * NOUT;
		uint8_t px[8][4];
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const int sh = 8 * (j & 3);
			const int a = (int)(((j < 4 ? r0[r].x : r0[r].y) >> sh) & 255u), b = (int)(((j < 4 ? r1[r].x : r1[r].y) >> sh) & 255u);
			const int c = (int)(((j < 4 ? r2[r].x : r2[r].y) >> sh) & 255u), k = (int)(((j < 4 ? r3[r].x : r3[r].y) >> sh) & 255u);
			int R, G, B;
			if (color == MIJ_COLOR_RGB) {
				R = a, G = b, B = c;
			} else if (color == MIJ_COLOR_CMYK) {
				R = blinn8(a, k), G = blinn8(b, k), B = blinn8(c, k);
			} else { /* YCCK */
				ycbcr_to_rgb(a, b, c, R, G, B);
				R = blinn8(255 - R, k), G = blinn8(255 - G, k), B = blinn8(255 - B, k);
			}
			px[j][0] = (uint8_t)R, px[j][1] = (uint8_t)G, px[j][2] = (uint8_t)B, px[j][3] = 255;
		}
		if (g.whole) {
			uint32_t q[2 * NOUT];
			if constexpr (NOUT == 4) {
#pragma unroll
				for (int j = 0; j < 8; ++j)
					q[j] = (uint32_t)px[j][0] | ((uint32_t)px[j][1] << 8) | ((uint32_t)px[j][2] << 16) | 0xff000000u;
			} else {
#pragma unroll
				for (int h = 0; h < 2; ++h) {
					const int j = 4 * h;
					q[3 * h + 0] = (uint32_t)px[j][0] | ((uint32_t)px[j][1] << 8) | ((uint32_t)px[j][2] << 16) | ((uint32_t)px[j + 1][0] << 24);
					q[3 * h + 1] = (uint32_t)px[j + 1][1] | ((uint32_t)px[j + 1][2] << 8) | ((uint32_t)px[j + 2][0] << 16) | ((uint32_t)px[j + 2][1] << 24);
					q[3 * h + 2] = (uint32_t)px[j + 2][2] | ((uint32_t)px[j + 3][0] << 8) | ((uint32_t)px[j + 3][1] << 16) | ((uint32_t)px[j + 3][2] << 24);
				}
			}
			store_row8<NOUT>(q, g.via_lds, g.lds_row, lane, dst, dst - (size_t)lane * (8 * NOUT));
		} else {
#pragma unroll
			for (int j = 0; j < 8; ++j)
				if (g.x0 + j < W)
					store_rgb_px<NOUT>(dst + j * NOUT, px[j][0], px[j][1], px[j][2]);
		}
	}
}

/* ------------------------------------------------------------------ two-pass path, pass 2, specialised per layout
 *
 * k_resample_color above decides everything at run time, per sample (resampler, row scheduler, colour branch, byte
 * loads with a clamp each): about 110 VALU instructions per pixel.  The layouts that actually reach the two-pass path
 * have component 0 (and 3) at full resolution and components 1 and 2 sharing one pair of factors, so pass 2 is
 * compiled once per resampler of load_jpeg_image's choice (codec/jpeg.c:2280-2289):
 *   RS_ROW1  hs 1, vs != 2   resample_row_1, or the generic resampler with hs == 1 (:1765, :1962): the near row
 *   RS_V2    hs 1, vs 2      resample_row_v_2 (:1774-1782): (3 near + far + 2) >> 2, four samples at once in 16-bit lanes
 *   RS_H2    hs 2, vs 1      resample_row_h_2 (:1784-1812): one v_dot4_u32_u8 per sample, as k_fused422
 *   RS_HV2   hs 2, vs 2      resample_row_hv_2 (:1816-1840): one v_dot4_u32_u8 per sample, as k_fused420
 *   RS_GEN2  hs 2, vs > 2    resample_row_generic (:1962-1971): every sample of the near row twice
 *   RS_GEN4  hs 4            the same, four times (4:1:1, 4:1:0)
 * Work item = MIJ_RESAMPLE_ROWS output rows of one image, a thread takes 4-pixel strips: plane bytes come in as dwords,
 * pixels leave as n_out dwords.  YCC = YCbCr -> RGB through color_px / store_px4 (the fused kernels' arithmetic); otherwise
 * the colour branch is store_pixel's (RGB-tagged, CMYK, YCCK).  The host sends an image here only if W % 4 == 0,
 * n_out >= 3 and its factors divide (resample_fast_kind in mij_runtime.hip); everything else stays with k_resample_color.
 */
enum { RS_ROW1 = 0, RS_V2, RS_H2, RS_HV2, RS_GEN2, RS_GEN4, RS_KINDS };

/* bytes (c[i0-1], c[i0], c[i0+1], c[i0+2]) of a plane row, i0 even; bytes outside the row are whatever the neighbouring
 * dword holds (the caller's edge selector replaces them); never reads in front of the row or behind its last dword */
__device__ __forceinline__ uint32_t rs_window(const uint8_t *__restrict__ row, int i0, int last_dword)
{
	/* no branch (the loads of a strip's windows must be in flight together): i0 == 0 takes dword 0 twice and shifts by 3 bytes,
	 * which leaves (c[3], c[0], c[1], c[2]) -- byte 0 is the one the edge selector replaces */
	const uint32_t *p = reinterpret_cast<const uint32_t *>(row);
	const int d0 = (i0 - 1) >> 2;
	const uint32_t lo = p[max(d0, 0)], hi = p[min(d0 + 1, last_dword)];
	return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(i0 - 1) & 3u);
}

template <int KIND, bool YCC, int NOUT>
__global__ __launch_bounds__(256) void k_resample_fast(const DevImage *__restrict__ imgs, const WorkIdct *__restrict__ work, const uint8_t *__restrict__ planes,
																		 uint8_t *__restrict__ outbase)
{
	/* chroma of a strip: four samples packed in one dword (PACKED), or one register per sample with the sample in byte 1 */
	constexpr bool PACKED = KIND != RS_H2 && KIND != RS_HV2;
	/* vs == 2 resamplers work on the row pair (2C-1, 2C), which shares the chroma rows C-1 and C (as k_fused420 does): work item
	 * `first` = chroma rows first/2 and first/2 + 1, i.e. output rows first-1 .. first+2 */
	constexpr bool PAIRS = KIND == RS_V2 || KIND == RS_HV2;
	const WorkIdct wk = work[blockIdx.x];
	const DevImage &im = imgs[wk.img];
	const int W = im.width, H = im.height;
	const int nstrip = W >> 2;
	const int w2Y = im.comp[0].bw * 8, w2C = im.comp[1].bw * 8;
	const int wc = (W + 1) >> 1, lastdw = (w2C >> 2) - 1;
	const int vsC = im.comp[1].vs, yC = im.comp[1].y;
	const bool four = !YCC && (im.color == MIJ_COLOR_CMYK || im.color == MIJ_COLOR_YCCK);
	const uint8_t *const pY = planes + im.comp[0].plane_off, *const pB = planes + im.comp[1].plane_off, *const pR = planes + im.comp[2].plane_off;
	const uint8_t *const pK = planes + im.comp[four ? 3 : 0].plane_off;
	const int w2K = im.comp[four ? 3 : 0].bw * 8;
	uint8_t *const out = outbase + im.out_off;
	ColorK KC;
	KC.init();
	const uint32_t wE0 = vreg(0x0000c040u), wO0 = vreg(0x0040c000u), wE1 = vreg(0x00c04000u), wO1 = vreg(0x40c00000u); /* h_2 weights x64, see k_fused422 */

	/* colour + store of one strip of row r from its up-sampled chroma */
	auto luma = [&](int r, int s) -> uint32_t { return *reinterpret_cast<const uint32_t *>(pY + (size_t)r * w2Y + 4 * s); };
	auto black = [&](int r, int s) -> uint32_t { return four ? *reinterpret_cast<const uint32_t *>(pK + (size_t)r * w2K + 4 * s) : 0u; };
	auto finish = [&](int r, int s, uint32_t yv, uint32_t kv, uint32_t cb4, uint32_t cr4, const uint32_t (&cbv)[4], const uint32_t (&crv)[4]) {
		uint8_t *dst = out + ((size_t)r * W + 4 * s) * NOUT;
		if (YCC) {
			Rgb12 p[4];
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				/* (chroma | luma << 16): byte j of the packed dword, or byte 1 of the sample's own register */
				const uint32_t sel = PACKED ? (0x0c000c04u + (uint32_t)j * 0x00010001u) : (0x0c000c05u + ((uint32_t)j << 16));
				p[j] = color_px(KC, __builtin_amdgcn_perm(PACKED ? cr4 : crv[j], yv, sel), __builtin_amdgcn_perm(PACKED ? cb4 : cbv[j], yv, sel));
			}
			store_px4<NOUT>(dst, p[0], p[1], p[2], p[3]);
		} else {
			uint8_t px[4][4];
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const int smp[4] = {(int)((yv >> (8 * j)) & 255u), (int)(PACKED ? (cb4 >> (8 * j)) & 255u : (cbv[j] >> 8) & 255u),
										  (int)(PACKED ? (cr4 >> (8 * j)) & 255u : (crv[j] >> 8) & 255u), (int)((kv >> (8 * j)) & 255u)};
				store_pixel(px[j], NOUT, im.color, smp);
			}
			uint32_t *q = reinterpret_cast<uint32_t *>(dst);
			if (NOUT == 4) {
#pragma unroll
				for (int j = 0; j < 4; ++j)
					q[j] = px[j][0] | px[j][1] << 8 | px[j][2] << 16 | (uint32_t)px[j][3] << 24;
			} else {
				q[0] = px[0][0] | px[0][1] << 8 | px[0][2] << 16 | (uint32_t)px[1][0] << 24;
				q[1] = px[1][1] | px[1][2] << 8 | px[2][0] << 16 | (uint32_t)px[2][1] << 24;
				q[2] = px[2][2] | px[3][0] << 8 | px[3][1] << 16 | (uint32_t)px[3][2] << 24;
			}
		}
	};
	/* the image's left / right edge of a 2:1 window: column -1 -> 0, column wc -> wc-1 (the reference's end cases are the
	 * general form with the neighbour clamped); *last: the strip holds pixel 2*(wc-1), h_2's odd one out (codec/jpeg.c:1805) */
	auto edge_sel = [&](int i0, bool *last) -> uint32_t {
		uint32_t sel = 0x03020100u;
		if (i0 == 0)
			sel = (sel & 0xffffff00u) | 0x01u;
		*last = i0 + 2 > wc - 1;
		if (*last)
			sel = (sel & 0x00ffffffu) | 0x02000000u;
		return sel;
	};
	const uint32_t none[4] = {0, 0, 0, 0};

	if (PAIRS) {
		for (int C = (int)wk.first >> 1; C < ((int)wk.first >> 1) + 2; ++C) {
			const int ra = 2 * C - 1, rb = 2 * C;
			const bool doA = ra >= 0 && ra < H, doB = rb < H;
			if (!doA && !doB)
				continue;
			/* chroma rows line0 / line1 of the scheduler (select_rows): A is near for row 2C-1, B for row 2C */
			const int rowA = C ? min(C - 1, yC - 1) : 0, rowB = min(C, yC - 1);
			const uint8_t *const bA_ = pB + (size_t)rowA * w2C, *const bB_ = pB + (size_t)rowB * w2C;
			const uint8_t *const rA_ = pR + (size_t)rowA * w2C, *const rB_ = pR + (size_t)rowB * w2C;
			for (int s = threadIdx.x; s < nstrip; s += 256) {
				const uint32_t yA = luma(doA ? ra : rb, s), yB = luma(doB ? rb : ra, s), kA = black(doA ? ra : rb, s), kB = black(doB ? rb : ra, s);
				if (KIND == RS_V2) {
					const uint32_t ba = *reinterpret_cast<const uint32_t *>(bA_ + 4 * s), bb = *reinterpret_cast<const uint32_t *>(bB_ + 4 * s);
					const uint32_t ra4 = *reinterpret_cast<const uint32_t *>(rA_ + 4 * s), rb4 = *reinterpret_cast<const uint32_t *>(rB_ + 4 * s);
					if (doB)
						finish(rb, s, yB, kB, rs_v2(bb, ba), rs_v2(rb4, ra4), none, none);
					if (doA)
						finish(ra, s, yA, kA, rs_v2(ba, bb), rs_v2(ra4, rb4), none, none);
				} else {
					const int i0 = 2 * s;
					uint32_t bA = rs_window(bA_, i0, lastdw), bB = rs_window(bB_, i0, lastdw), rA = rs_window(rA_, i0, lastdw), rB = rs_window(rB_, i0, lastdw);
					if (i0 == 0 || i0 + 2 > wc - 1) {
						bool last;
						const uint32_t sel = edge_sel(i0, &last);
						bA = __builtin_amdgcn_perm(0, bA, sel);
						bB = __builtin_amdgcn_perm(0, bB, sel);
						rA = __builtin_amdgcn_perm(0, rA, sel);
						rB = __builtin_amdgcn_perm(0, rB, sel);
					}
					const uint32_t vb0 = __builtin_amdgcn_perm(bA, bB, KC.v0), vb1 = __builtin_amdgcn_perm(bA, bB, KC.v1), vb2 = __builtin_amdgcn_perm(bA, bB, KC.v2);
					const uint32_t vr0 = __builtin_amdgcn_perm(rA, rB, KC.v0), vr1 = __builtin_amdgcn_perm(rA, rB, KC.v1), vr2 = __builtin_amdgcn_perm(rA, rB, KC.v2);
					if (doB) {
						const uint32_t cbv[4] = {dot4(vb0, KC.wBk1, KC.w128), dot4(vb1, KC.wBk, KC.w128), dot4(vb1, KC.wBk1, KC.w128), dot4(vb2, KC.wBk, KC.w128)};
						const uint32_t crv[4] = {dot4(vr0, KC.wBk1, KC.w128), dot4(vr1, KC.wBk, KC.w128), dot4(vr1, KC.wBk1, KC.w128), dot4(vr2, KC.wBk, KC.w128)};
						finish(rb, s, yB, kB, 0, 0, cbv, crv);
					}
					if (doA) {
						const uint32_t cbv[4] = {dot4(vb0, KC.wAk1, KC.w128), dot4(vb1, KC.wAk, KC.w128), dot4(vb1, KC.wAk1, KC.w128), dot4(vb2, KC.wAk, KC.w128)};
						const uint32_t crv[4] = {dot4(vr0, KC.wAk1, KC.w128), dot4(vr1, KC.wAk, KC.w128), dot4(vr1, KC.wAk1, KC.w128), dot4(vr2, KC.wAk, KC.w128)};
						finish(ra, s, yA, kA, 0, 0, cbv, crv);
					}
				}
			}
		}
		return;
	}

	const int r0 = (int)wk.first, r1 = min(r0 + MIJ_RESAMPLE_ROWS, H);
	for (int r = r0; r < r1; ++r) {
		const RowSel rs = select_rows(r, vsC, yC);
		const uint8_t *const bN = pB + (size_t)rs.near * w2C, *const rN = pR + (size_t)rs.near * w2C;
		for (int s = threadIdx.x; s < nstrip; s += 256) {
			const uint32_t yv = luma(r, s), kv = black(r, s);
			if (KIND == RS_ROW1) {
				finish(r, s, yv, kv, *reinterpret_cast<const uint32_t *>(bN + 4 * s), *reinterpret_cast<const uint32_t *>(rN + 4 * s), none, none);
			} else if (KIND == RS_GEN2) {
				finish(r, s, yv, kv, __builtin_amdgcn_perm(0, *reinterpret_cast<const uint16_t *>(bN + 2 * s), 0x01010000u),
						 __builtin_amdgcn_perm(0, *reinterpret_cast<const uint16_t *>(rN + 2 * s), 0x01010000u), none, none);
			} else if (KIND == RS_GEN4) {
				finish(r, s, yv, kv, (uint32_t)bN[s] * 0x01010101u, (uint32_t)rN[s] * 0x01010101u, none, none);
			} else { /* RS_H2 */
				const int i0 = 2 * s;
				uint32_t vb = rs_window(bN, i0, lastdw), vr = rs_window(rN, i0, lastdw);
				uint32_t w2 = wE1;
				if (i0 == 0 || i0 + 2 > wc - 1) {
					bool last;
					const uint32_t sel = edge_sel(i0, &last);
					if (last)
						w2 = wO0;
					vb = __builtin_amdgcn_perm(0, vb, sel);
					vr = __builtin_amdgcn_perm(0, vr, sel);
				}
				const uint32_t cbv[4] = {dot4(vb, wE0, KC.w128), dot4(vb, wO0, KC.w128), dot4(vb, w2, KC.w128), dot4(vb, wO1, KC.w128)};
				const uint32_t crv[4] = {dot4(vr, wE0, KC.w128), dot4(vr, wO0, KC.w128), dot4(vr, w2, KC.w128), dot4(vr, wO1, KC.w128)};
				finish(r, s, yv, kv, 0, 0, cbv, crv);
			}
		}
	}
}

/* ------------------------------------------------------------------ encoder: colour + subsample + fDCT + quantiser
 *
 * The writer's arithmetic is float and order-sensitive (codec/jpeg_write.c:24-74, :96-118, :283-352):
 * this translation unit is compiled with -ffp-contract=off, every expression keeps the reference's
 * association, and conversions truncate like the C casts do.  IEEE single add/mul are correctly
 * rounded on both sides, so the data units are bit-identical to the host's (tests compare them).
 * One lane = one MCU: its data units are produced one after the other (luma quadrants, then the
 * 2x2-mean chroma units), re-reading the MCU's pixels (L1/L2 hits) rather than holding 768 floats.
 * Output: int16[64] per data unit, zigzag order, MCU after MCU -- what the host Huffman stage reads.
 */
struct EncImage {
	int32_t width, height, comp, subsample;
	int32_t mcu_x, mcu_y, flip, pad;
	uint64_t pix_off, du_off;
	float fy[64], fc[64];
};

typedef float f2 __attribute__((ext_vector_type(2)));

/* stbiw__jpg_DCT (codec/jpeg_write.c:24-74) on T = float, or on T = f2: two independent transforms in
 * the two halves of v_pk_add_f32 / v_pk_mul_f32 operands -- the same IEEE operations in the same order */
template <typename T>
__device__ __forceinline__ void fdct8(T &d0, T &d1, T &d2, T &d3, T &d4, T &d5, T &d6, T &d7)
{
	const T a0 = d0 + d7, a7 = d0 - d7, a1 = d1 + d6, a6 = d1 - d6;
	const T a2 = d2 + d5, a5 = d2 - d5, a3 = d3 + d4, a4 = d3 - d4;
	T b0 = a0 + a3, b3 = a0 - a3, b1 = a1 + a2, b2 = a1 - a2;
	const T o0 = b0 + b1, o4 = b0 - b1;
	const T z1 = (b2 + b3) * 0.707106781f;
	const T o2 = b3 + z1, o6 = b3 - z1;
	b0 = a4 + a5;
	b1 = a5 + a6;
	b2 = a6 + a7;
	const T z5 = (b0 - b2) * 0.382683433f;
	const T z2 = b0 * 0.541196100f + z5;
	const T z4 = b2 * 1.306562965f + z5;
	const T z3 = b1 * 0.707106781f;
	const T z11 = a7 + z3, z13 = a7 - z3;
	d5 = z13 + z2;
	d3 = z13 - z2;
	d1 = z11 + z4;
	d7 = z11 - z4;
	d0 = o0;
	d2 = o2;
	d4 = o4;
	d6 = o6;
}

/* copysign(0.5f, v) = (v & 0x80000000) | 0.5f in one v_bitop3_b32: the compiler's v_bfi_b32 issues at 35 T lane-ops/s, this at 53
 * (profiles/r02s_isa_probe4.txt); truth table for S0 = 0xf0, S1 = 0xcc, S2 = 0xaa: (S0 & S1) | S2 = 0xea */
__device__ __forceinline__ float half_signed(float v)
{
	float d;
	asm("v_bitop3_b32 %0, %1, %2, 0.5 bitop3:0xea" : "=v"(d) : "v"(v), "s"(0x80000000u));
	return d;
}

/* (int)lo | (int)hi << 16 as int16 halves, the C casts' truncation: the second conversion writes its low half straight into the upper
 * half of the first one's result (SDWA destination select), two instructions a pair where shifts and masks made it four or five */
__device__ __forceinline__ uint32_t cvt_pack_i16(float lo, float hi)
{
	uint32_t d;
	asm("v_cvt_i32_f32_e32 %0, %1\n\tv_cvt_i32_f32_sdwa %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "=&v"(d) : "v"(lo), "v"(hi));
	return d;
}

/* Rows, then columns, then quantise into zigzag order and store 128 bytes (codec/jpeg_write.c:96-118).
 * V[k][x] = samples (row 2k, row 2k+1) of column x: the row pass runs on row pairs, a 2x2 re-pairing
 * turns them into column pairs H[y][j] = (col 2j, col 2j+1) of row y for the column pass.
 * "(int)(v < 0 ? v - 0.5f : v + 0.5f)" is spelled v + copysign(0.5, v): identical for every finite v
 * (for v = -0.0 both give 0). */
template <int SWZ = -1>
__device__ __forceinline__ void fdct_quant_store(f2 (&V)[4][8], const float *__restrict__ fdtbl, int16_t *__restrict__ dst, int swz = 0)
{
	constexpr int zz[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
									10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
#pragma unroll
	for (int k = 0; k < 4; ++k)
		fdct8<f2>(V[k][0], V[k][1], V[k][2], V[k][3], V[k][4], V[k][5], V[k][6], V[k][7]);
	f2 H[8][4];
#pragma unroll
	for (int k = 0; k < 4; ++k)
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			H[2 * k][j] = (f2){V[k][2 * j].x, V[k][2 * j + 1].x};
			H[2 * k + 1][j] = (f2){V[k][2 * j].y, V[k][2 * j + 1].y};
		}
#pragma unroll
	for (int j = 0; j < 4; ++j)
		fdct8<f2>(H[0][j], H[1][j], H[2][j], H[3][j], H[4][j], H[5][j], H[6][j], H[7][j]);
	float q[64]; /* rounded by the addition below, still float; zigzag order */
#pragma unroll
	for (int y = 0; y < 8; ++y)
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			f2 v = H[y][j] * (f2){fdtbl[8 * y + 2 * j], fdtbl[8 * y + 2 * j + 1]};
			v = v + (f2){half_signed(v.x), half_signed(v.y)};
			q[zz[8 * y + 2 * j]] = v.x;
			q[zz[8 * y + 2 * j + 1]] = v.y;
		}
	uint32_t *o = reinterpret_cast<uint32_t *>(dst);
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		uint4 w;
		w.x = cvt_pack_i16(q[8 * k + 0], q[8 * k + 1]);
		w.y = cvt_pack_i16(q[8 * k + 2], q[8 * k + 3]);
		w.z = cvt_pack_i16(q[8 * k + 4], q[8 * k + 5]);
		w.w = cvt_pack_i16(q[8 * k + 6], q[8 * k + 7]);
		/* SWZ >= 0: 16-byte chunk k of the unit lands at chunk k ^ swz (LDS staging, see k_encode420) */
		*reinterpret_cast<uint4 *>(o + 4 * (SWZ >= 0 ? (k ^ swz) : k)) = w;
	}
}

/* the same from 64 scalars in row-major order */
__device__ __forceinline__ void fdct_quant_store(float (&d)[64], const float *__restrict__ fdtbl, int16_t *__restrict__ dst)
{
	f2 V[4][8];
#pragma unroll
	for (int k = 0; k < 4; ++k)
#pragma unroll
		for (int x = 0; x < 8; ++x)
			V[k][x] = (f2){d[16 * k + x], d[16 * k + 8 + x]};
	fdct_quant_store(V, fdtbl, dst);
}

/* the three colour transforms, spelled as the reference spells them (codec/jpeg_write.c:298-300) */
template <int C>
__device__ __forceinline__ float enc_component(float r, float g, float b)
{
	if (C == 0)
		return +0.29900f * r + 0.58700f * g + 0.11400f * b - 128;
	if (C == 1)
		return -0.16874f * r - 0.33126f * g + 0.50000f * b;
	return +0.50000f * r - 0.41869f * g - 0.08131f * b;
}

/* keeps the loads of one pixel row from being hoisted above the previous row's arithmetic: without it
 * the fully unrolled fetch of a 16x16 MCU wants ~800 live registers and spills */
__device__ __forceinline__ void enc_row_fence()
{
	asm volatile("" ::: "memory");
	__builtin_amdgcn_sched_barrier(0);
}

/* pixel fetch with the reference's edge replication (codec/jpeg_write.c:289-296); 32-bit offsets:
 * column offsets are computed once per data unit, the row base once per row */
struct EncPix {
	const uint8_t *px;
	int W, H, comp, og, ob, flip;
	__device__ __forceinline__ uint32_t row_base(int row) const
	{
		const int crow = row < H ? row : H - 1;
		return (uint32_t)(flip ? (H - 1 - crow) : crow) * (uint32_t)W * (uint32_t)comp;
	}
	__device__ __forceinline__ uint32_t col_off(int col) const { return (uint32_t)(col < W ? col : W - 1) * (uint32_t)comp; }
	template <int C>
	__device__ __forceinline__ float at(uint32_t p) const
	{
		return enc_component<C>((float)px[p], (float)px[p + og], (float)px[p + ob]);
	}
};

/* One lane = one data unit.  Two kernels so that a wave never mixes the 64-pixel luma units with the
 * 256-pixel (2x2 mean) chroma units: k_encode_y (SUB: 4 units per MCU, else 1), k_encode_c (2 per MCU).
 * work.first counts data units of that kind. */
__device__ __forceinline__ void enc_setup(const EncImage &im, const uint8_t *pix, EncPix &P)
{
	P.px = pix + im.pix_off;
	P.W = im.width;
	P.H = im.height;
	P.comp = im.comp;
	P.og = im.comp > 2 ? 1 : 0;
	P.ob = im.comp > 2 ? 2 : 0;
	P.flip = im.flip;
}

/* byte k of a row of dwords as float (v_cvt_f32_ubyteN) */
template <int K>
__device__ __forceinline__ float enc_byte(const uint32_t *w)
{
	return (float)((w[K >> 2] >> (8 * (K & 3))) & 0xffu);
}

/* Lanes walk the luma data units in raster order of 8x8 blocks (consecutive lanes = horizontally
 * adjacent blocks = contiguous pixel bytes); the unit is stored at its place in MCU order. */
template <int SUB>
__global__ __launch_bounds__(256) void k_encode_y(const EncImage *__restrict__ imgs, const WorkIdct *__restrict__ work, const uint8_t *__restrict__ pix,
																  int16_t *__restrict__ du)
{
	const WorkIdct wk = work[blockIdx.x];
	const EncImage &im = imgs[wk.img];
	const uint32_t per = SUB ? 4u : 1u;
	const uint32_t t = wk.first + threadIdx.x;
	if (t >= (uint32_t)(im.mcu_x * im.mcu_y) * per)
		return;
	const uint32_t bpr = (uint32_t)im.mcu_x * (SUB ? 2u : 1u); /* 8x8 luma blocks per block row */
	const uint32_t by = t / bpr, bx = t - by * bpr;
	const uint32_t m = SUB ? (by >> 1) * (uint32_t)im.mcu_x + (bx >> 1) : t;
	const uint32_t q = SUB ? ((by & 1u) << 1 | (bx & 1u)) : 0u;
	EncPix P;
	enc_setup(im, pix, P);
	const int qy = 8 * (int)by, qx = 8 * (int)bx;
	float d[64];
	if (im.comp == 3 && (im.width & 15) == 0) {
		/* whole rows inside the image, 8 pixels = 24 bytes, 8-byte aligned: three 8-byte loads per row */
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const uint2 *rp = reinterpret_cast<const uint2 *>(P.px + P.row_base(qy + i) + (uint32_t)qx * 3u);
			const uint2 a = rp[0], b = rp[1], c = rp[2];
			const uint32_t w[6] = {a.x, a.y, b.x, b.y, c.x, c.y};
			d[8 * i + 0] = enc_component<0>(enc_byte<0>(w), enc_byte<1>(w), enc_byte<2>(w));
			d[8 * i + 1] = enc_component<0>(enc_byte<3>(w), enc_byte<4>(w), enc_byte<5>(w));
			d[8 * i + 2] = enc_component<0>(enc_byte<6>(w), enc_byte<7>(w), enc_byte<8>(w));
			d[8 * i + 3] = enc_component<0>(enc_byte<9>(w), enc_byte<10>(w), enc_byte<11>(w));
			d[8 * i + 4] = enc_component<0>(enc_byte<12>(w), enc_byte<13>(w), enc_byte<14>(w));
			d[8 * i + 5] = enc_component<0>(enc_byte<15>(w), enc_byte<16>(w), enc_byte<17>(w));
			d[8 * i + 6] = enc_component<0>(enc_byte<18>(w), enc_byte<19>(w), enc_byte<20>(w));
			d[8 * i + 7] = enc_component<0>(enc_byte<21>(w), enc_byte<22>(w), enc_byte<23>(w));
		}
	} else {
		uint32_t co[8];
#pragma unroll
		for (int j = 0; j < 8; ++j)
			co[j] = P.col_off(qx + j);
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const uint32_t rb = P.row_base(qy + i);
#pragma unroll
			for (int j = 0; j < 8; ++j)
				d[8 * i + j] = P.at<0>(rb + co[j]);
		}
	}
	fdct_quant_store(d, im.fy, du + im.du_off / 2 + (size_t)m * (SUB ? 384 : 192) + 64 * q);
}

/* one 2x2-mean chroma sample from two rows of packed RGB dwords; J = sample column 0..7 */
template <int C, int J>
__device__ __forceinline__ float enc_mean4(const uint32_t *r0, const uint32_t *r1)
{
	const float a = enc_component<C>(enc_byte<6 * J + 0>(r0), enc_byte<6 * J + 1>(r0), enc_byte<6 * J + 2>(r0));
	const float b = enc_component<C>(enc_byte<6 * J + 3>(r0), enc_byte<6 * J + 4>(r0), enc_byte<6 * J + 5>(r0));
	const float c = enc_component<C>(enc_byte<6 * J + 0>(r1), enc_byte<6 * J + 1>(r1), enc_byte<6 * J + 2>(r1));
	const float e = enc_component<C>(enc_byte<6 * J + 3>(r1), enc_byte<6 * J + 4>(r1), enc_byte<6 * J + 5>(r1));
	return (a + b + c + e) * 0.25f;
}

template <int C>
__device__ __forceinline__ void enc_mean_row(const uint32_t *r0, const uint32_t *r1, float *d)
{
	d[0] = enc_mean4<C, 0>(r0, r1);
	d[1] = enc_mean4<C, 1>(r0, r1);
	d[2] = enc_mean4<C, 2>(r0, r1);
	d[3] = enc_mean4<C, 3>(r0, r1);
	d[4] = enc_mean4<C, 4>(r0, r1);
	d[5] = enc_mean4<C, 5>(r0, r1);
	d[6] = enc_mean4<C, 6>(r0, r1);
	d[7] = enc_mean4<C, 7>(r0, r1);
}

template <int SUB>
__global__ __launch_bounds__(256) void k_encode_c(const EncImage *__restrict__ imgs, const WorkIdct *__restrict__ work, const uint8_t *__restrict__ pix,
																  int16_t *__restrict__ du)
{
	const WorkIdct wk = work[blockIdx.x];
	const EncImage &im = imgs[wk.img];
	const uint32_t nmcu = (uint32_t)(im.mcu_x * im.mcu_y);
	const uint32_t t = wk.first + threadIdx.x;
	if (t >= nmcu * 2u)
		return;
	/* first all U units in MCU raster order, then all V units: consecutive lanes read adjacent MCUs */
	const uint32_t c = t >= nmcu ? 1u : 0u, m = t - c * nmcu;
	const int my = (int)(m / (uint32_t)im.mcu_x), mx = (int)(m - (uint32_t)my * (uint32_t)im.mcu_x);
	EncPix P;
	enc_setup(im, pix, P);
	float d[64];
	if (SUB) {
		const int y0 = 16 * my, x0 = 16 * mx;
		if (im.comp == 3 && (im.width & 15) == 0) {
			/* 16 pixels = 48 bytes per row, 16-byte aligned: three 16-byte loads per row */
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				const uint4 *p0 = reinterpret_cast<const uint4 *>(P.px + P.row_base(y0 + 2 * i) + (uint32_t)x0 * 3u);
				const uint4 *p1 = reinterpret_cast<const uint4 *>(P.px + P.row_base(y0 + 2 * i + 1) + (uint32_t)x0 * 3u);
				const uint4 a0 = p0[0], a1 = p0[1], a2 = p0[2], b0 = p1[0], b1 = p1[1], b2 = p1[2];
				const uint32_t r0[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
				const uint32_t r1[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
				if (c == 0)
					enc_mean_row<1>(r0, r1, d + 8 * i);
				else
					enc_mean_row<2>(r0, r1, d + 8 * i);
				enc_row_fence();
			}
		} else {
			uint32_t co[16];
#pragma unroll
			for (int j = 0; j < 16; ++j)
				co[j] = P.col_off(x0 + j);
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				const uint32_t r0 = P.row_base(y0 + 2 * i), r1 = P.row_base(y0 + 2 * i + 1);
#pragma unroll
				for (int j = 0; j < 8; ++j) {
					if (c == 0)
						d[8 * i + j] = (P.at<1>(r0 + co[2 * j]) + P.at<1>(r0 + co[2 * j + 1]) + P.at<1>(r1 + co[2 * j]) + P.at<1>(r1 + co[2 * j + 1])) * 0.25f;
					else
						d[8 * i + j] = (P.at<2>(r0 + co[2 * j]) + P.at<2>(r0 + co[2 * j + 1]) + P.at<2>(r1 + co[2 * j]) + P.at<2>(r1 + co[2 * j + 1])) * 0.25f;
				}
				enc_row_fence();
			}
		}
		fdct_quant_store(d, im.fc, du + im.du_off / 2 + (size_t)m * 384 + 256 + 64 * c);
	} else {
		const int y0 = 8 * my, x0 = 8 * mx;
		uint32_t co[8];
#pragma unroll
		for (int j = 0; j < 8; ++j)
			co[j] = P.col_off(x0 + j);
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const uint32_t rb = P.row_base(y0 + i);
#pragma unroll
			for (int j = 0; j < 8; ++j)
				d[8 * i + j] = c == 0 ? P.at<1>(rb + co[j]) : P.at<2>(rb + co[j]);
		}
		fdct_quant_store(d, im.fc, du + im.du_off / 2 + (size_t)m * 192 + 64 + 64 * c);
	}
}

/* ------------------------------------------------------------------ fused 4:2:0 encoder (3-component images; rows staged as whole MCU columns, mij_runtime.hip enc_padded_width)
 *
 * The per-unit kernels above read every pixel twice (luma pass, chroma pass) with lane-strided
 * accesses, convert every byte to float twice, and write each lane's 128-byte unit as eight 16-byte
 * pieces of 64 different cache lines.  Here one workgroup (3 waves) owns a strip of 32 consecutive
 * MCUs (MCU index order, so a strip may run over the end of an MCU row):
 *   load     16 pixel rows x 32 MCUs x 48 B, each thread eight coalesced 16-byte loads -> LDS [16][1536]
 *   convert  waves 0 / 1: lane = (mcu, bx) of luma block row 0 / 1.  Two pixel rows at a time as f2
 *            (top, bottom) pairs: luma straight into the row-pair form the DCT wants; U and V together as
 *            (U, V) pairs with coefficient pairs (a - b*c == a + (-b)*c in IEEE arithmetic, so signed
 *            constants keep the bits), 2x2 means in the reference's order, staged as floats in LDS in
 *            the same row-pair order
 *   DCT      waves 0 / 1 their luma unit; wave 2: lane = (mcu, U|V), unit read back from the float stage
 *   store    units staged in LDS (swizzled 16-byte chunks) over the dead pixel rows, then the strip's 32 x 768
 *            contiguous output bytes written with coalesced 16-byte stores.
 * Pixels are read once and converted once: algorithmic traffic only.
 */
#define MIJ_ENC_STRIP 32
#define MIJ_ENC_PIXROW (MIJ_ENC_STRIP * 48)
/* staged units are unpadded; the 16-byte chunk index is XOR-swizzled with the unit index instead, so that
 * 16-byte accesses of neighbouring lanes fall in different banks: 24 KiB + 16 KiB = 40 KiB, 4 workgroups per CU */
#define MIJ_ENC_DUPITCH 128
#define MIJ_ENC_CPITCH 256
#define MIJ_ENC_LDS_A (16 * MIJ_ENC_PIXROW) /* == MIJ_ENC_STRIP * 6 * MIJ_ENC_DUPITCH */
/* MIJ_ENC_INPLACE (default): the chroma means go over pixel rows their wave has already consumed, so the workgroup needs the 24 KiB
 * of the pixel rows only and five workgroups (15 waves) fit a CU where the separate 16 KiB stage allowed four; 0 = the round-1 layout */
#ifndef MIJ_ENC_INPLACE
#define MIJ_ENC_INPLACE 1
#endif
#if MIJ_ENC_INPLACE
#define MIJ_ENC_LDS MIJ_ENC_LDS_A
#else
#define MIJ_ENC_LDS (MIJ_ENC_LDS_A + MIJ_ENC_STRIP * 2 * MIJ_ENC_CPITCH)
#endif
__device__ __forceinline__ int enc_du_chunk(int u, int k) { return u * MIJ_ENC_DUPITCH + ((k ^ (u & 7)) << 4); }
#if MIJ_ENC_INPLACE
/* Chroma means of MCU j, component c, row pair rp = 2 * wave + kk of the unit, 16-byte chunk q (two columns x two rows): inside the
 * four pixel rows luma wave `wave` read in its step kk (every lane of the wave has read them before any lane writes: one wave, program
 * order), 256 chunks per region, eight chunks per MCU, swizzled with the MCU index so that both the 16 lanes that write one (c, q) and
 * the 16 that read it spread over all eight 16-byte bank groups. */
__device__ __forceinline__ int enc_cf_chunk(int j, int c, int rp, int q)
{
	return (4 * rp) * MIJ_ENC_PIXROW + ((j * 8 + ((c * 4 + q) ^ (j & 7))) << 4);
}
#else
__device__ __forceinline__ int enc_cf_chunk(int j, int c, int rp, int q)
{
	const int u = 2 * j + c, h = 4 * rp + q;
	return MIJ_ENC_LDS_A + u * MIJ_ENC_CPITCH + ((h ^ (u & 15)) << 4);
}
#endif

/* (k.x * a.y, k.y * a.y): the high half of a register pair broadcast by the instruction's operand select.  The compiler
 * only knows the low-half broadcast and copies a.y into a fresh pair first (one v_mov per product, 96 per luma lane). */
__device__ __forceinline__ f2 pk_mul_bhi(f2 k, f2 a)
{
	f2 d;
	asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "s"(k));
	return d;
}

/* byte k of a row of dwords as float, for a (top, bottom) row pair */
template <int K>
__device__ __forceinline__ f2 enc_byte2(const uint32_t *t, const uint32_t *b)
{
	return (f2){enc_byte<K>(t), enc_byte<K>(b)};
}

/* round 1: three waves per SIMD (<= 168 VGPRs; 1.72 ms vs 1.90 ms per 512 1080p images with two).  Round 2: 118 VGPRs once the
 * quantiser table sits behind one pointer and the high-half broadcasts stopped costing copies -> four */
#ifndef MIJ_ENC_WAVES
#define MIJ_ENC_WAVES 4
#endif
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(MIJ_ENC_WAVES, MIJ_ENC_WAVES))) void k_encode420(
	const EncImage *__restrict__ imgs, const WorkIdct *__restrict__ work, const uint8_t *__restrict__ pix, int16_t *__restrict__ du)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint8_t *const spx = lds;                 /* pixel rows, later the staged data units */
	uint8_t *const sdu = lds;
	uint8_t *const scf = lds;                 /* chroma means as floats (enc_cf_chunk) */
	const WorkIdct wk = work[blockIdx.x];
	const EncImage &im = imgs[wk.img];
	const int tid = threadIdx.x;
	const uint32_t nmcu = (uint32_t)(im.mcu_x * im.mcu_y);
	const uint32_t m0 = wk.first;
	const uint32_t cnt = min((uint32_t)MIJ_ENC_STRIP, nmcu - m0);
	EncPix P;
	enc_setup(im, pix, P);

	/* ---- load: thread -> fixed 16-byte column chunk of the strip, rows 2i + (tid / 96) */
	{
		const uint32_t col = (uint32_t)tid % 96u, rsel = (uint32_t)tid / 96u;
		const uint32_t j = col / 3u, part = col - 3u * j;
		const uint32_t m = min(m0 + j, nmcu - 1u); /* past the last MCU: any valid pixels, the units are dropped */
		const uint32_t my = m / (uint32_t)im.mcu_x, mx = m - my * (uint32_t)im.mcu_x;
		const uint32_t cbase = mx * 48u + part * 16u;
		uint4 v[8];
#pragma unroll
		for (int i = 0; i < 8; ++i)
		{
			const u4v t = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(P.px + P.row_base((int)(16u * my + 2u * (uint32_t)i + rsel)) + cbase));
			v[i] = make_uint4(t.x, t.y, t.z, t.w);
		}
#pragma unroll
		for (int i = 0; i < 8; ++i)
			*reinterpret_cast<uint4 *>(spx + (2 * i + (int)rsel) * MIJ_ENC_PIXROW + (int)col * 16) = v[i];
	}
	__syncthreads();

	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	f2 V[4][8];
	if (wave < 2) {
		/* codec/jpeg_write.c:298-300; (U, V) coefficient pairs */
		const f2 KR = {-0.16874f, +0.50000f}, KG = {-0.33126f, -0.41869f}, KB = {+0.50000f, -0.08131f};
		const int j = lane >> 1, bx = lane & 1;
		const uint8_t *src = spx + (8 * wave) * MIJ_ENC_PIXROW + lane * 24;
#pragma unroll
		for (int kk = 0; kk < 2; ++kk) {
			f2 M[2][4]; /* [sample row & 1][sample column] = (U mean, V mean) */
#pragma unroll
			for (int k2 = 0; k2 < 2; ++k2) {
				const int k = 2 * kk + k2;
				const uint2 *tp = reinterpret_cast<const uint2 *>(src + (2 * k) * MIJ_ENC_PIXROW);
				const uint2 *bp = reinterpret_cast<const uint2 *>(src + (2 * k + 1) * MIJ_ENC_PIXROW);
				const uint2 t0 = tp[0], t1 = tp[1], t2 = tp[2], b0 = bp[0], b1 = bp[1], b2 = bp[2];
				const uint32_t t[6] = {t0.x, t0.y, t1.x, t1.y, t2.x, t2.y};
				const uint32_t b[6] = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y};
				f2 Wt[8], Wb[8];
#define MIJ_ENC_PX(XI)                                                                                                      \
	{                                                                                                                       \
		const f2 r = enc_byte2<3 * (XI) + 0>(t, b), g = enc_byte2<3 * (XI) + 1>(t, b), bl = enc_byte2<3 * (XI) + 2>(t, b);      \
		V[k][XI] = 0.29900f * r + 0.58700f * g + 0.11400f * bl - 128.0f;                                                      \
		Wt[XI] = KR * r.x + KG * g.x + KB * bl.x;                                                                             \
		Wb[XI] = pk_mul_bhi(KR, r) + pk_mul_bhi(KG, g) + pk_mul_bhi(KB, bl);                                                  \
	}
				MIJ_ENC_PX(0) MIJ_ENC_PX(1) MIJ_ENC_PX(2) MIJ_ENC_PX(3) MIJ_ENC_PX(4) MIJ_ENC_PX(5) MIJ_ENC_PX(6) MIJ_ENC_PX(7)
#undef MIJ_ENC_PX
#pragma unroll
				for (int jj = 0; jj < 4; ++jj)
					M[k2][jj] = (Wt[2 * jj] + Wt[2 * jj + 1] + Wb[2 * jj] + Wb[2 * jj + 1]) * 0.25f; /* :317-318 */
				enc_row_fence();
			}
			/* sample rows 4*wave + 2*kk (+1) of the chroma unit = its row pair 2*wave + kk, columns 4*bx .. 4*bx+3 */
			const int rp = 2 * wave + kk, q0 = bx * 2; /* 16-byte chunk: 4 floats = 2 columns x (row, row + 1) */
			*reinterpret_cast<float4 *>(scf + enc_cf_chunk(j, 0, rp, q0)) = make_float4(M[0][0].x, M[1][0].x, M[0][1].x, M[1][1].x);
			*reinterpret_cast<float4 *>(scf + enc_cf_chunk(j, 0, rp, q0 + 1)) = make_float4(M[0][2].x, M[1][2].x, M[0][3].x, M[1][3].x);
			*reinterpret_cast<float4 *>(scf + enc_cf_chunk(j, 1, rp, q0)) = make_float4(M[0][0].y, M[1][0].y, M[0][1].y, M[1][1].y);
			*reinterpret_cast<float4 *>(scf + enc_cf_chunk(j, 1, rp, q0 + 1)) = make_float4(M[0][2].y, M[1][2].y, M[0][3].y, M[1][3].y);
		}
	}
	__syncthreads(); /* pixel rows dead, chroma means staged */

	if (wave == 2) {
#pragma unroll
		for (int k = 0; k < 4; ++k)
#pragma unroll
			for (int h = 0; h < 4; ++h) {
				const float4 f = *reinterpret_cast<const float4 *>(scf + enc_cf_chunk(lane >> 1, lane & 1, k, h));
				V[k][2 * h] = (f2){f.x, f.y};
				V[k][2 * h + 1] = (f2){f.z, f.w};
			}
	}
#if MIJ_ENC_INPLACE
	__syncthreads(); /* the means are in registers: the staged units may now go over them */
#endif
	{
		/* one instance of the transform for all three waves, its quantiser table behind ONE wave-uniform pointer: with a call per
		 * branch the compiler merged the tails and carried 64 separate table addresses into them (64 scalar loads, each waited
		 * for where it was used, and their offsets spilled to VGPR lanes) */
		const float *fd = wave == 2 ? im.fc : im.fy;
		const int u = (lane >> 1) * 6 + 2 * wave + (lane & 1); /* wave 2: 4 + (lane & 1) */
		fdct_quant_store<0>(V, fd, reinterpret_cast<int16_t *>(sdu + u * MIJ_ENC_DUPITCH), u & 7);
	}
	__syncthreads();

	/* ---- store: cnt * 768 contiguous bytes */
	{
		uint8_t *out = reinterpret_cast<uint8_t *>(du) + im.du_off + (size_t)m0 * 768u;
		const int nchunk = (int)cnt * 48;
		for (int g = tid; g < nchunk; g += 192)
		{
			const uint4 t = *reinterpret_cast<const uint4 *>(sdu + enc_du_chunk(g >> 3, g & 7));
			__builtin_nontemporal_store((u4v){t.x, t.y, t.z, t.w}, reinterpret_cast<u4v *>(out + (size_t)g * 16u));
		}
	}
}

/* ------------------------------------------------------------------ fused 4:4:4 encoder (3-component images; rows staged as whole MCU columns)
 *
 * The writer takes 4:4:4 for every quality above 90 (codec/jpeg_write.c:221, :283-352: MCU = one 8x8 block per component).
 * The per-unit kernels read the pixels of such a picture three times with per-byte loads; here one workgroup (3 waves)
 * owns a strip of 64 consecutive MCUs (MCU index order, so a strip may run over the end of an MCU row):
 *   load     8 pixel rows x 64 MCUs x 24 B, each thread eight coalesced 8-byte loads -> LDS [8][1536]
 *   convert  wave c = component c (Y, U, V), lane = MCU: the lane's 8 x 24 bytes from LDS, two pixel rows at a time as f2
 *            (top, bottom) pairs straight into the row-pair form the DCT wants (enc_component<c>'s expression on f2)
 *   DCT      every lane its own unit (all three waves carry the same load), quantised with fy (Y) or fc (U, V)
 *   store    units staged in LDS (swizzled 16-byte chunks) over the dead pixel rows, then the strip's 64 x 384 contiguous
 *            output bytes with coalesced 16-byte stores.
 * Each byte is converted three times (once per component) but read from HBM once; algorithmic bytes 3 + 6 per pixel.
 */
#define MIJ_ENC444_STRIP 64
#define MIJ_ENC444_PIXROW (MIJ_ENC444_STRIP * 24)
#define MIJ_ENC444_LDS (MIJ_ENC444_STRIP * 3 * MIJ_ENC_DUPITCH) /* staged units; the pixel rows (8 x 1536 B) lie inside */

template <int C>
__device__ __forceinline__ f2 enc_component2(f2 r, f2 g, f2 b)
{
	if (C == 0)
		return +0.29900f * r + 0.58700f * g + 0.11400f * b - 128.0f;
	if (C == 1)
		return -0.16874f * r - 0.33126f * g + 0.50000f * b;
	return +0.50000f * r - 0.41869f * g - 0.08131f * b;
}

template <int C>
__device__ __forceinline__ void enc444_convert(const uint8_t *src, f2 (&V)[4][8])
{
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const uint2 *tp = reinterpret_cast<const uint2 *>(src + (2 * k) * MIJ_ENC444_PIXROW);
		const uint2 *bp = reinterpret_cast<const uint2 *>(src + (2 * k + 1) * MIJ_ENC444_PIXROW);
		const uint2 t0 = tp[0], t1 = tp[1], t2 = tp[2], b0 = bp[0], b1 = bp[1], b2 = bp[2];
		const uint32_t t[6] = {t0.x, t0.y, t1.x, t1.y, t2.x, t2.y};
		const uint32_t b[6] = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y};
#define MIJ_ENC_PX(XI) V[k][XI] = enc_component2<C>(enc_byte2<3 * (XI) + 0>(t, b), enc_byte2<3 * (XI) + 1>(t, b), enc_byte2<3 * (XI) + 2>(t, b));
		MIJ_ENC_PX(0) MIJ_ENC_PX(1) MIJ_ENC_PX(2) MIJ_ENC_PX(3) MIJ_ENC_PX(4) MIJ_ENC_PX(5) MIJ_ENC_PX(6) MIJ_ENC_PX(7)
#undef MIJ_ENC_PX
		enc_row_fence();
	}
}

__global__ __launch_bounds__(192) void k_encode444(const EncImage *__restrict__ imgs, const WorkIdct *__restrict__ work, const uint8_t *__restrict__ pix,
																	int16_t *__restrict__ du)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	uint8_t *const spx = lds; /* pixel rows, later the staged data units */
	uint8_t *const sdu = lds;
	const WorkIdct wk = work[blockIdx.x];
	const EncImage &im = imgs[wk.img];
	const int tid = threadIdx.x;
	const uint32_t nmcu = (uint32_t)(im.mcu_x * im.mcu_y);
	const uint32_t m0 = wk.first;
	const uint32_t cnt = min((uint32_t)MIJ_ENC444_STRIP, nmcu - m0);
	EncPix P;
	enc_setup(im, pix, P);

	/* ---- load: thread -> fixed 8-byte column chunk of the strip (MCU tid / 3, third tid % 3), rows 0..7 */
	{
		const uint32_t j = (uint32_t)tid / 3u, part = (uint32_t)tid - 3u * j;
		const uint32_t m = min(m0 + j, nmcu - 1u); /* past the last MCU: any valid pixels, the units are dropped */
		const uint32_t my = m / (uint32_t)im.mcu_x, mx = m - my * (uint32_t)im.mcu_x;
		const uint32_t cbase = mx * 24u + part * 8u;
		uint2 v[8];
#pragma unroll
		for (int i = 0; i < 8; ++i)
			v[i] = *reinterpret_cast<const uint2 *>(P.px + P.row_base((int)(8u * my) + i) + cbase);
#pragma unroll
		for (int i = 0; i < 8; ++i)
			*reinterpret_cast<uint2 *>(spx + i * MIJ_ENC444_PIXROW + tid * 8) = v[i];
	}
	__syncthreads();

	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	f2 V[4][8];
	const uint8_t *src = spx + lane * 24;
	if (wave == 0)
		enc444_convert<0>(src, V);
	else if (wave == 1)
		enc444_convert<1>(src, V);
	else
		enc444_convert<2>(src, V);
	__syncthreads(); /* pixel rows dead */

	{
		const int u = lane * 3 + wave;
		fdct_quant_store<0>(V, wave == 0 ? im.fy : im.fc, reinterpret_cast<int16_t *>(sdu + u * MIJ_ENC_DUPITCH), u & 7);
	}
	__syncthreads();

	/* ---- store: cnt * 384 contiguous bytes */
	{
		uint8_t *out = reinterpret_cast<uint8_t *>(du) + im.du_off + (size_t)m0 * 384u;
		const int nchunk = (int)cnt * 24;
		for (int g = tid; g < nchunk; g += 192) {
			const uint4 t = *reinterpret_cast<const uint4 *>(sdu + enc_du_chunk(g >> 3, g & 7));
			__builtin_nontemporal_store((u4v){t.x, t.y, t.z, t.w}, reinterpret_cast<u4v *>(out + (size_t)g * 16u));
		}
	}
}

} /* namespace mij */

#endif
