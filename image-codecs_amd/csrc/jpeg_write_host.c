/*
 * jpeg_write_host.c -- baseline JFIF writer behind stbi_write_jpg / stbi_write_jpg_to_func.
 *
 * Restates the reference encoder (codec/jpeg_write.c:1-388) so that the produced byte stream is
 * identical: same quality mapping and tables (:220-243), same headers (:245-268), the same
 * float colour transform, edge replication and 2x2 chroma mean (:283-325 / :330-352), the same
 * float AAN forward DCT operation order (:24-74, :96-105), the same round-to-nearest quantiser
 * (:107-118) and the same Huffman emission (:120-169, :4-22).  Float results depend on the
 * operation order, so this file must be compiled with -ffp-contract=off and without fast-math.
 *
 * The forward DCT + quantisation of this path (BASELINE config 5) still
 * runs on the host in this round; the GPU kernel for it is the next step behind the same API.
 * The benchmark uses this writer to synthesise its JPEG inputs on the GPU box.
 *
 * The Huffman code tables are derived from the Annex-K BITS/HUFFVAL lists (which the file has
 * to carry anyway for its DHT segment) instead of being spelled out as literal code tables.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <stdint.h>
#include <emmintrin.h>
#include <xmmintrin.h>

#include "image_api.h"
#include "mij_host.h"

static int g_flip_on_write = 0;
void stbi_flip_vertically_on_write(int flag) { g_flip_on_write = flag; }

/* natural index -> zigzag position (codec/jpeg_write.c:1-2) */
static const unsigned char k_zigzag_of[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30,
															 41, 43, 9,  11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
															 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

/* Annex K.3 tables: BITS (index 0 unused) and HUFFVAL */
static const unsigned char k_dc_lum_bits[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const unsigned char k_dc_vals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const unsigned char k_dc_chr_bits[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const unsigned char k_ac_lum_bits[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const unsigned char k_ac_lum_vals[162] = {
	0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
	0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
	0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
	0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
	0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
	0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
	0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
	0xfa};
static const unsigned char k_ac_chr_bits[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const unsigned char k_ac_chr_vals[162] = {
	0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
	0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19,
	0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55,
	0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83,
	0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
	0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
	0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
	0xfa};

/* base quantisation tables, natural order (codec/jpeg_write.c:204-207) */
static const int k_qt_lum[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,  14, 13, 16, 24, 40,  57,
											69, 56, 14, 17, 22,  29,  51,  87,  80, 62, 18, 22, 37,  56,  68,  109, 103, 77, 24, 35, 55,  64,
											81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const int k_qt_chr[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
											99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

typedef struct {
	unsigned short code[256], len[256];
} enc_table;

/* canonical code assignment of Annex C: gives the reference's literal YDC_HT/UVDC_HT/YAC_HT/UVAC_HT */
static void make_enc_table(enc_table *t, const unsigned char *bits, const unsigned char *vals)
{
	int l, i, k = 0;
	unsigned code = 0;
	memset(t, 0, sizeof(*t));
	for (l = 1; l <= 16; ++l) {
		for (i = 0; i < bits[l]; ++i, ++k) {
			t->code[vals[k]] = (unsigned short)code++;
			t->len[vals[k]] = (unsigned short)l;
		}
		code <<= 1;
	}
}

/* Output side.  The reference pushes every byte through the callback one at a time from a 24-bit bit buffer
 * (codec/jpeg_write.c:4-22); the byte stream is a function of the symbols alone, so here the bits gather in a 64-bit
 * accumulator and leave four bytes at a time -- byte stuffing (0xFF -> 0xFF 0x00) only when one of the four is 0xFF --
 * into a buffer handed to the callback in chunks of up to 4 KiB.  1080p: 7 ms -> 2.6 ms per picture. */
typedef struct {
	stbi_write_func *func;
	void *context;
	unsigned char buf[4096 + 16];
	int used;
	uint64_t acc; /* the low `nacc` bits are pending output, oldest bit highest */
	int nacc;
} jw_sink;

static void sink_flush(jw_sink *s)
{
	if (s->used) {
		s->func(s->context, s->buf, s->used);
		s->used = 0;
	}
}
static inline void sink_byte(jw_sink *s, unsigned char c)
{
	if (s->used >= 4096)
		sink_flush(s);
	s->buf[s->used++] = c;
}
static void sink_bytes(jw_sink *s, const void *p, int n)
{
	const unsigned char *b = (const unsigned char *)p;
	int i;
	for (i = 0; i < n; ++i)
		sink_byte(s, b[i]);
}

/* codec/jpeg_write.c:4-22: len <= 27 bits (a code of at most 16 plus at most 11 magnitude bits) */
static inline void put_bits(jw_sink *s, unsigned code, int len)
{
	s->acc = (s->acc << len) | (uint64_t)code;
	s->nacc += len;
	if (s->nacc >= 32) {
		const uint32_t w = (uint32_t)(s->acc >> (s->nacc - 32));
		s->nacc -= 32;
		if (s->used > 4096 - 8)
			sink_flush(s);
		/* a byte of w is 0xFF <=> the same byte of ~w is zero (the classic has-zero-byte test) */
		if ((((~w) - 0x01010101u) & w & 0x80808080u) == 0) {
			unsigned char *o = s->buf + s->used;
			o[0] = (unsigned char)(w >> 24);
			o[1] = (unsigned char)(w >> 16);
			o[2] = (unsigned char)(w >> 8);
			o[3] = (unsigned char)w;
			s->used += 4;
		} else {
			int k;
			for (k = 24; k >= 0; k -= 8) {
				const unsigned char c = (unsigned char)(w >> k);
				s->buf[s->used++] = c;
				if (c == 255)
					s->buf[s->used++] = 0;
			}
		}
	}
}

/* the whole bytes still pending (the reference emits a byte as soon as it has eight bits; what is left below a byte after the
 * final fill bits is dropped there too, codec/jpeg_write.c:358-360) */
static void put_bits_finish(jw_sink *s)
{
	while (s->nacc >= 8) {
		const unsigned char c = (unsigned char)(s->acc >> (s->nacc - 8));
		s->nacc -= 8;
		sink_byte(s, c);
		if (c == 255)
			sink_byte(s, 0);
	}
}

/* magnitude category and the bits that follow it (codec/jpeg_write.c:76-86); val != 0 */
static inline void magnitude_bits(int val, unsigned *bits, int *nbits)
{
	const unsigned a = (unsigned)(val < 0 ? -val : val);
	const int v = val < 0 ? val - 1 : val, n = 32 - __builtin_clz(a);
	*nbits = n;
	*bits = (unsigned)v & ((1u << n) - 1u);
}

/* one 8-point AAN pass over p[0], p[s], ... p[7s]; operation order of codec/jpeg_write.c:24-74 */
static inline void fdct8(float *p, int s)
{
	float d0 = p[0], d1 = p[s], d2 = p[2 * s], d3 = p[3 * s], d4 = p[4 * s], d5 = p[5 * s], d6 = p[6 * s], d7 = p[7 * s];
	float a0 = d0 + d7, a7 = d0 - d7;
	float a1 = d1 + d6, a6 = d1 - d6;
	float a2 = d2 + d5, a5 = d2 - d5;
	float a3 = d3 + d4, a4 = d3 - d4;
	/* even part */
	float b0 = a0 + a3, b3 = a0 - a3;
	float b1 = a1 + a2, b2 = a1 - a2;
	float z1, z2, z3, z4, z5, z11, z13;
	float o0 = b0 + b1, o4 = b0 - b1;
	float o2, o6;
	z1 = (b2 + b3) * 0.707106781f;
	o2 = b3 + z1;
	o6 = b3 - z1;
	/* odd part */
	b0 = a4 + a5;
	b1 = a5 + a6;
	b2 = a6 + a7;
	z5 = (b0 - b2) * 0.382683433f;
	z2 = b0 * 0.541196100f + z5;
	z4 = b2 * 1.306562965f + z5;
	z3 = b1 * 0.707106781f;
	z11 = a7 + z3;
	z13 = a7 - z3;
	p[5 * s] = z13 + z2;
	p[3 * s] = z13 - z2;
	p[s] = z11 + z4;
	p[7 * s] = z11 - z4;
	p[0] = o0;
	p[2 * s] = o2;
	p[4 * s] = o4;
	p[6 * s] = o6;
}

/* The same pass on four independent 1-D transforms at once (one per SSE lane): every lane performs exactly the operations of
 * fdct8 in exactly its order -- mulps / addps / subps are IEEE single operations, nothing is contracted (-ffp-contract=off) --
 * so the results are the scalar code's bit for bit. */
static inline void fdct8_ps(__m128 *d)
{
	const __m128 c707 = _mm_set1_ps(0.707106781f), c382 = _mm_set1_ps(0.382683433f), c541 = _mm_set1_ps(0.541196100f), c1306 = _mm_set1_ps(1.306562965f);
	const __m128 a0 = _mm_add_ps(d[0], d[7]), a7 = _mm_sub_ps(d[0], d[7]), a1 = _mm_add_ps(d[1], d[6]), a6 = _mm_sub_ps(d[1], d[6]);
	const __m128 a2 = _mm_add_ps(d[2], d[5]), a5 = _mm_sub_ps(d[2], d[5]), a3 = _mm_add_ps(d[3], d[4]), a4 = _mm_sub_ps(d[3], d[4]);
	__m128 b0 = _mm_add_ps(a0, a3), b3 = _mm_sub_ps(a0, a3), b1 = _mm_add_ps(a1, a2), b2 = _mm_sub_ps(a1, a2);
	const __m128 o0 = _mm_add_ps(b0, b1), o4 = _mm_sub_ps(b0, b1);
	const __m128 z1 = _mm_mul_ps(_mm_add_ps(b2, b3), c707);
	const __m128 o2 = _mm_add_ps(b3, z1), o6 = _mm_sub_ps(b3, z1);
	__m128 z2, z3, z4, z5, z11, z13;
	b0 = _mm_add_ps(a4, a5);
	b1 = _mm_add_ps(a5, a6);
	b2 = _mm_add_ps(a6, a7);
	z5 = _mm_mul_ps(_mm_sub_ps(b0, b2), c382);
	z2 = _mm_add_ps(_mm_mul_ps(b0, c541), z5);
	z4 = _mm_add_ps(_mm_mul_ps(b2, c1306), z5);
	z3 = _mm_mul_ps(b1, c707);
	z11 = _mm_add_ps(a7, z3);
	z13 = _mm_sub_ps(a7, z3);
	d[5] = _mm_add_ps(z13, z2);
	d[3] = _mm_sub_ps(z13, z2);
	d[1] = _mm_add_ps(z11, z4);
	d[7] = _mm_sub_ps(z11, z4);
	d[0] = o0;
	d[2] = o2;
	d[4] = o4;
	d[6] = o6;
}

/* forward DCT + quantise one data unit into zigzag order (codec/jpeg_write.c:96-118): rows, then columns, then
 * "(int)(v < 0 ? v - 0.5f : v + 0.5f)" with v = coefficient * fdtbl.  Row pass: the block is transposed so that the eight row
 * transforms sit in SSE lanes, transformed, transposed back; the column pass has its transforms in lanes as the block lies. */
static void transform_du(float *cdu, int stride, const float *fdtbl, int16_t *du)
{
	__m128 lo[8], hi[8], t[8];
	int y, j;
	for (y = 0; y < 8; ++y) {
		lo[y] = _mm_loadu_ps(cdu + y * stride);
		hi[y] = _mm_loadu_ps(cdu + y * stride + 4);
	}
	/* rows 0..3 and 4..7 as lanes: t[k] = (column k of rows 0..3), u[k] = (column k of rows 4..7) */
	{
		__m128 u[8];
		__m128 r0 = lo[0], r1 = lo[1], r2 = lo[2], r3 = lo[3];
		_MM_TRANSPOSE4_PS(r0, r1, r2, r3);
		t[0] = r0, t[1] = r1, t[2] = r2, t[3] = r3;
		r0 = hi[0], r1 = hi[1], r2 = hi[2], r3 = hi[3];
		_MM_TRANSPOSE4_PS(r0, r1, r2, r3);
		t[4] = r0, t[5] = r1, t[6] = r2, t[7] = r3;
		r0 = lo[4], r1 = lo[5], r2 = lo[6], r3 = lo[7];
		_MM_TRANSPOSE4_PS(r0, r1, r2, r3);
		u[0] = r0, u[1] = r1, u[2] = r2, u[3] = r3;
		r0 = hi[4], r1 = hi[5], r2 = hi[6], r3 = hi[7];
		_MM_TRANSPOSE4_PS(r0, r1, r2, r3);
		u[4] = r0, u[5] = r1, u[6] = r2, u[7] = r3;
		fdct8_ps(t);
		fdct8_ps(u);
		/* back: row y = (t[0..7] lane y) for y < 4, (u[0..7] lane y - 4) otherwise */
		r0 = t[0], r1 = t[1], r2 = t[2], r3 = t[3];
		_MM_TRANSPOSE4_PS(r0, r1, r2, r3);
		lo[0] = r0, lo[1] = r1, lo[2] = r2, lo[3] = r3;
		r0 = t[4], r1 = t[5], r2 = t[6], r3 = t[7];
		_MM_TRANSPOSE4_PS(r0, r1, r2, r3);
		hi[0] = r0, hi[1] = r1, hi[2] = r2, hi[3] = r3;
		r0 = u[0], r1 = u[1], r2 = u[2], r3 = u[3];
		_MM_TRANSPOSE4_PS(r0, r1, r2, r3);
		lo[4] = r0, lo[5] = r1, lo[6] = r2, lo[7] = r3;
		r0 = u[4], r1 = u[5], r2 = u[6], r3 = u[7];
		_MM_TRANSPOSE4_PS(r0, r1, r2, r3);
		hi[4] = r0, hi[5] = r1, hi[6] = r2, hi[7] = r3;
	}
	fdct8_ps(lo); /* columns 0..3 */
	fdct8_ps(hi); /* columns 4..7 */
	for (y = 0, j = 0; y < 8; ++y, j += 8) {
		const __m128 half = _mm_set1_ps(0.5f), zero = _mm_setzero_ps();
		const __m128 va = _mm_mul_ps(lo[y], _mm_loadu_ps(fdtbl + j)), vb = _mm_mul_ps(hi[y], _mm_loadu_ps(fdtbl + j + 4));
		/* v < 0 ? v - 0.5f : v + 0.5f, then the C cast's truncation */
		const __m128 ma = _mm_cmplt_ps(va, zero), mb = _mm_cmplt_ps(vb, zero);
		const __m128 ra = _mm_or_ps(_mm_and_ps(ma, _mm_sub_ps(va, half)), _mm_andnot_ps(ma, _mm_add_ps(va, half)));
		const __m128 rb = _mm_or_ps(_mm_and_ps(mb, _mm_sub_ps(vb, half)), _mm_andnot_ps(mb, _mm_add_ps(vb, half)));
		int32_t q[8];
		int x;
		_mm_storeu_si128((__m128i *)q, _mm_cvttps_epi32(ra));
		_mm_storeu_si128((__m128i *)(q + 4), _mm_cvttps_epi32(rb));
		for (x = 0; x < 8; ++x)
			du[k_zigzag_of[j + x]] = (int16_t)q[x];
	}
}

/* Huffman-code one quantised data unit (codec/jpeg_write.c:120-169); returns its DC.  The reference scans the unit for its last
 * non-zero coefficient and then for every run of zeros; the same symbols come out of walking the set bits of the unit's
 * non-zero mask (sixteen int16 per SSE2 compare), a code and its magnitude bits leaving in one put_bits. */
static int emit_du(jw_sink *s, const int16_t *du, int dc_pred, const enc_table *hdc, const enc_table *hac)
{
	const int diff = du[0] - dc_pred;
	unsigned bits;
	int nbits, prev = 0, k;
	uint64_t nz = 0;
	const __m128i zero = _mm_setzero_si128();
	if (diff == 0) {
		put_bits(s, hdc->code[0], hdc->len[0]);
	} else {
		magnitude_bits(diff, &bits, &nbits);
		put_bits(s, ((unsigned)hdc->code[nbits] << nbits) | bits, hdc->len[nbits] + nbits);
	}
	for (k = 0; k < 4; ++k) {
		const __m128i a = _mm_loadu_si128((const __m128i *)(du + 16 * k)), b = _mm_loadu_si128((const __m128i *)(du + 16 * k + 8));
		const unsigned m = (unsigned)_mm_movemask_epi8(_mm_packs_epi16(_mm_cmpeq_epi16(a, zero), _mm_cmpeq_epi16(b, zero))); /* bit i: coefficient is zero */
		nz |= (uint64_t)(~m & 0xffffu) << (16 * k);
	}
	nz &= ~(uint64_t)1;
	if (!nz) {
		put_bits(s, hac->code[0x00], hac->len[0x00]);
		return du[0];
	}
	while (nz) {
		const int i = __builtin_ctzll(nz);
		int run = i - prev - 1;
		prev = i;
		nz &= nz - 1;
		for (; run >= 16; run -= 16)
			put_bits(s, hac->code[0xF0], hac->len[0xF0]);
		magnitude_bits(du[i], &bits, &nbits);
		put_bits(s, ((unsigned)hac->code[(run << 4) + nbits] << nbits) | bits, hac->len[(run << 4) + nbits] + nbits);
	}
	if (prev != 63)
		put_bits(s, hac->code[0x00], hac->len[0x00]);
	return du[0];
}

/*
 * The writer in three separable steps (so that step 2 can run on the GPU, mij_enc_* in mij.h):
 *   1. mjw_plan_init      quality mapping, quantisation + scaled-reciprocal tables (codec/jpeg_write.c:220-243)
 *   2. mjw_transform_host colour transform, edge replication, 2x2 chroma mean, fDCT, quantiser for every
 *                         data unit (:283-352 minus the Huffman calls): DU blocks, MCU order, zigzag order
 *   3. mjw_emit           headers (:245-268), Huffman emission (:120-169), padding and EOI (:358-363)
 */
int mjw_plan_init(mjw_plan *p, int width, int height, int comp, int quality)
{
	static const float aasf[8] = {1.0f * 2.828427125f,         1.387039845f * 2.828427125f, 1.306562965f * 2.828427125f, 1.175875602f * 2.828427125f,
											1.0f * 2.828427125f,         0.785694958f * 2.828427125f, 0.541196100f * 2.828427125f, 0.275899379f * 2.828427125f};
	int i, row, col, k, mcu;
	if (!p || !width || !height || comp > 4 || comp < 1 || width < 0 || height < 0)
		return 0;
	quality = quality ? quality : 90;
	p->subsample = quality <= 90 ? 1 : 0;
	quality = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
	quality = quality < 50 ? 5000 / quality : 200 - quality * 2;
	for (i = 0; i < 64; ++i) {
		int yq = (k_qt_lum[i] * quality + 50) / 100;
		int cq = (k_qt_chr[i] * quality + 50) / 100;
		p->ytab[k_zigzag_of[i]] = (unsigned char)(yq < 1 ? 1 : (yq > 255 ? 255 : yq));
		p->ctab[k_zigzag_of[i]] = (unsigned char)(cq < 1 ? 1 : (cq > 255 ? 255 : cq));
	}
	for (row = 0, k = 0; row < 8; ++row)
		for (col = 0; col < 8; ++col, ++k) {
			p->fdtbl_y[k] = 1 / (p->ytab[k_zigzag_of[k]] * aasf[row] * aasf[col]);
			p->fdtbl_c[k] = 1 / (p->ctab[k_zigzag_of[k]] * aasf[row] * aasf[col]);
		}
	p->width = width;
	p->height = height;
	p->comp = comp;
	mcu = p->subsample ? 16 : 8;
	p->mcu_x = (width + mcu - 1) / mcu;
	p->mcu_y = (height + mcu - 1) / mcu;
	p->du_per_mcu = p->subsample ? 6 : 3;
	return 1;
}

size_t mjw_plan_du_count(const mjw_plan *p) { return (size_t)p->mcu_x * (size_t)p->mcu_y * (size_t)p->du_per_mcu; }

void mjw_transform_host(const mjw_plan *p, const void *data, int flip, int16_t *du)
{
	const int width = p->width, height = p->height, comp = p->comp;
	const int og = comp > 2 ? 1 : 0, ob = comp > 2 ? 2 : 0; /* comp 1/2: grey replicated */
	const unsigned char *px = (const unsigned char *)data;
	const int mcu = p->subsample ? 16 : 8;
	int x, y, row, col, pos;
	float Y[256], U[256], V[256], R[256], G[256], B[256];
	for (y = 0; y < height; y += mcu)
		for (x = 0; x < width; x += mcu) {
			const int npx = mcu * mcu;
			for (row = y, pos = 0; row < y + mcu; ++row) {
				int crow = row < height ? row : height - 1; /* replicate the last row / column */
				int base = (flip ? (height - 1 - crow) : crow) * width * comp;
				for (col = x; col < x + mcu; ++col, ++pos) {
					int q = base + (col < width ? col : width - 1) * comp;
					R[pos] = px[q], G[pos] = px[q + og], B[pos] = px[q + ob];
				}
			}
			/* codec/jpeg_write.c:298-300, four pixels per step; the association of the scalar expressions:
			 *   Y = ((0.299 r + 0.587 g) + 0.114 b) - 128    U = ((-0.16874 r) - 0.33126 g) + 0.5 b    V = ((0.5 r) - 0.41869 g) - 0.08131 b */
			for (pos = 0; pos < npx; pos += 4) {
				const __m128 r = _mm_loadu_ps(R + pos), g = _mm_loadu_ps(G + pos), b = _mm_loadu_ps(B + pos);
				_mm_storeu_ps(Y + pos, _mm_sub_ps(_mm_add_ps(_mm_add_ps(_mm_mul_ps(_mm_set1_ps(+0.29900f), r), _mm_mul_ps(_mm_set1_ps(0.58700f), g)),
																			  _mm_mul_ps(_mm_set1_ps(0.11400f), b)), _mm_set1_ps(128.0f)));
				_mm_storeu_ps(U + pos, _mm_add_ps(_mm_sub_ps(_mm_mul_ps(_mm_set1_ps(-0.16874f), r), _mm_mul_ps(_mm_set1_ps(0.33126f), g)),
															 _mm_mul_ps(_mm_set1_ps(0.50000f), b)));
				_mm_storeu_ps(V + pos, _mm_sub_ps(_mm_sub_ps(_mm_mul_ps(_mm_set1_ps(+0.50000f), r), _mm_mul_ps(_mm_set1_ps(0.41869f), g)),
															 _mm_mul_ps(_mm_set1_ps(0.08131f), b)));
			}
			if (p->subsample) {
				float su[64], sv[64];
				int yy, xx;
				transform_du(Y + 0, 16, p->fdtbl_y, du);
				transform_du(Y + 8, 16, p->fdtbl_y, du + 64);
				transform_du(Y + 128, 16, p->fdtbl_y, du + 128);
				transform_du(Y + 136, 16, p->fdtbl_y, du + 192);
				for (yy = 0, pos = 0; yy < 8; ++yy)
					for (xx = 0; xx < 8; ++xx, ++pos) {
						int j = yy * 32 + xx * 2;
						su[pos] = (U[j + 0] + U[j + 1] + U[j + 16] + U[j + 17]) * 0.25f;
						sv[pos] = (V[j + 0] + V[j + 1] + V[j + 16] + V[j + 17]) * 0.25f;
					}
				transform_du(su, 8, p->fdtbl_c, du + 256);
				transform_du(sv, 8, p->fdtbl_c, du + 320);
				du += 384;
			} else {
				transform_du(Y, 8, p->fdtbl_y, du);
				transform_du(U, 8, p->fdtbl_c, du + 64);
				transform_du(V, 8, p->fdtbl_c, du + 128);
				du += 192;
			}
		}
}

int mjw_emit(const mjw_plan *p, const int16_t *du, mjw_write_func *func, void *context)
{
	enc_table ydc, yac, cdc, cac;
	jw_sink *s;
	const int width = p->width, height = p->height;
	if (!func || !du)
		return 0;
	s = (jw_sink *)calloc(1, sizeof(*s));
	if (!s)
		return 0;
	s->func = func;
	s->context = context;
	make_enc_table(&ydc, k_dc_lum_bits, k_dc_vals);
	make_enc_table(&cdc, k_dc_chr_bits, k_dc_vals);
	make_enc_table(&yac, k_ac_lum_bits, k_ac_lum_vals);
	make_enc_table(&cac, k_ac_chr_bits, k_ac_chr_vals);
	/* headers (codec/jpeg_write.c:245-268) */
	{
		static const unsigned char soi_app0_dqt[] = {0xFF, 0xD8, 0xFF, 0xE0, 0, 0x10, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0, 0xFF, 0xDB, 0, 0x84, 0};
		static const unsigned char sos[] = {0xFF, 0xDA, 0, 0xC, 3, 1, 0, 2, 0x11, 3, 0x11, 0, 0x3F, 0};
		unsigned char sof_dht[24] = {0xFF, 0xC0, 0, 0x11, 8, 0, 0, 0, 0, 3, 1, 0, 0, 2, 0x11, 1, 3, 0x11, 1, 0xFF, 0xC4, 0x01, 0xA2, 0};
		sof_dht[5] = (unsigned char)(height >> 8);
		sof_dht[6] = (unsigned char)(height & 0xff);
		sof_dht[7] = (unsigned char)(width >> 8);
		sof_dht[8] = (unsigned char)(width & 0xff);
		sof_dht[11] = (unsigned char)(p->subsample ? 0x22 : 0x11);
		sink_bytes(s, soi_app0_dqt, (int)sizeof(soi_app0_dqt));
		sink_bytes(s, p->ytab, 64);
		sink_byte(s, 1);
		sink_bytes(s, p->ctab, 64);
		sink_bytes(s, sof_dht, (int)sizeof(sof_dht));
		sink_bytes(s, k_dc_lum_bits + 1, 16);
		sink_bytes(s, k_dc_vals, 12);
		sink_byte(s, 0x10);
		sink_bytes(s, k_ac_lum_bits + 1, 16);
		sink_bytes(s, k_ac_lum_vals, 162);
		sink_byte(s, 1);
		sink_bytes(s, k_dc_chr_bits + 1, 16);
		sink_bytes(s, k_dc_vals, 12);
		sink_byte(s, 0x11);
		sink_bytes(s, k_ac_chr_bits + 1, 16);
		sink_bytes(s, k_ac_chr_vals, 162);
		sink_bytes(s, sos, (int)sizeof(sos));
	}
	{
		int dcy = 0, dcu = 0, dcv = 0;
		size_t m, nm = (size_t)p->mcu_x * (size_t)p->mcu_y;
		s->acc = 0;
		s->nacc = 0;
		for (m = 0; m < nm; ++m) {
			if (p->subsample) {
				dcy = emit_du(s, du, dcy, &ydc, &yac);
				dcy = emit_du(s, du + 64, dcy, &ydc, &yac);
				dcy = emit_du(s, du + 128, dcy, &ydc, &yac);
				dcy = emit_du(s, du + 192, dcy, &ydc, &yac);
				dcu = emit_du(s, du + 256, dcu, &cdc, &cac);
				dcv = emit_du(s, du + 320, dcv, &cdc, &cac);
				du += 384;
			} else {
				dcy = emit_du(s, du, dcy, &ydc, &yac);
				dcu = emit_du(s, du + 64, dcu, &cdc, &cac);
				dcv = emit_du(s, du + 128, dcv, &cdc, &cac);
				du += 192;
			}
		}
		put_bits(s, 0x7F, 7); /* pad to a byte boundary with ones */
		put_bits_finish(s);
	}
	sink_byte(s, 0xFF);
	sink_byte(s, 0xD9);
	sink_flush(s);
	free(s);
	return 1;
}

typedef struct {
	unsigned char *out;
	size_t cap, len;
	int overflow;
} mem_sink;
static void mem_sink_write(void *context, void *data, int size)
{
	mem_sink *m = (mem_sink *)context;
	if (m->len + (size_t)size > m->cap) {
		m->overflow = 1;
		return;
	}
	memcpy(m->out + m->len, data, (size_t)size);
	m->len += (size_t)size;
}
size_t mjw_emit_to_memory(const mjw_plan *p, const int16_t *du, unsigned char *out, size_t cap)
{
	mem_sink m;
	if (!p || !du || !out)
		return 0;
	m.out = out;
	m.cap = cap;
	m.len = 0;
	m.overflow = 0;
	if (!mjw_emit(p, du, mem_sink_write, &m) || m.overflow)
		return 0;
	return m.len;
}

int stbi_write_jpg_to_func(stbi_write_func *func, void *context, int x, int y, int comp, const void *data, int quality)
{
	mjw_plan plan;
	int16_t *du;
	int ok;
	if (!func || !data)
		return 0;
	if (!mjw_plan_init(&plan, x, y, comp, quality))
		return 0;
	du = (int16_t *)malloc(mjw_plan_du_count(&plan) * 64 * sizeof(int16_t));
	if (!du)
		return 0;
	mjw_transform_host(&plan, data, g_flip_on_write, du);
	ok = mjw_emit(&plan, du, func, context);
	free(du);
	return ok;
}

int mjw_flip_on_write(void) { return g_flip_on_write; }

static void file_sink(void *context, void *data, int size) { fwrite(data, 1, (size_t)size, (FILE *)context); }

int stbi_write_jpg(char const *filename, int x, int y, int comp, const void *data, int quality)
{
	FILE *f = fopen(filename, "wb");
	int ok;
	if (!f)
		return 0;
	ok = stbi_write_jpg_to_func(file_sink, f, x, y, comp, data, quality);
	fclose(f);
	return ok;
}
