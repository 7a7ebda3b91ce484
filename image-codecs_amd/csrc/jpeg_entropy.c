/*
 * jpeg_entropy.c -- host half of the JPEG decode path: byte source, marker parsing, Huffman
 * tables and the sequential entropy-coded-segment walk (baseline + progressive).
 *
 * Behaviour follows /root/reference/codec/jpeg.c (line numbers cited per function); the code
 * is written for this project: decoded coefficients are NOT de-quantised or inverse
 * transformed here, they are stored as int16 in the GPU staging planes of mij.h (tile layout).
 * The bit reader keeps the reference's exact state machine (32-bit buffer, refill to >24 bits,
 * zero feed after a marker) because the observable behaviour on truncated, padded and
 * restart-marker streams depends on it.
 */
#include "jpeg_entropy.h"

#include <emmintrin.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ byte source */

void mjh_reader_mem(mjh_reader *r, const uint8_t *data, int len)
{
	memset(&r->io, 0, sizeof(r->io));
	r->user = NULL;
	r->from_callbacks = 0;
	r->already_read = 0;
	r->buflen = 0;
	r->p = r->orig = data;
	r->end = r->orig_end = data + len;
}

/* common.c:10-28 */
static void reader_refill(mjh_reader *r)
{
	int n = r->io.read(r->user, (char *)r->buf, r->buflen);
	r->already_read += (int)(r->p - r->orig);
	if (n == 0) {
		/* end of file: behave like a memory source that holds one 0 byte */
		r->from_callbacks = 0;
		r->buf[0] = 0;
		r->p = r->buf;
		r->end = r->buf + 1;
	} else {
		r->p = r->buf;
		r->end = r->buf + n;
	}
}

void mjh_reader_callbacks(mjh_reader *r, const stbi_io_callbacks *io, void *user)
{
	r->io = *io;
	r->user = user;
	r->buflen = (int)sizeof(r->buf);
	r->from_callbacks = 1;
	r->already_read = 0;
	r->p = r->orig = r->buf;
	reader_refill(r);
	r->orig_end = r->end;
}

static int file_read(void *user, char *data, int size) { return (int)fread(data, 1, (size_t)size, (FILE *)user); }
static void file_skip(void *user, int n)
{
	int ch;
	fseek((FILE *)user, n, SEEK_CUR);
	ch = fgetc((FILE *)user); /* make feof() meaningful right after the seek */
	if (ch != EOF)
		ungetc(ch, (FILE *)user);
}
static int file_eof(void *user) { return feof((FILE *)user) || ferror((FILE *)user); }

void mjh_reader_file(mjh_reader *r, FILE *f)
{
	stbi_io_callbacks io;
	io.read = file_read;
	io.skip = file_skip;
	io.eof = file_eof;
	mjh_reader_callbacks(r, &io, (void *)f);
}

void mjh_reader_rewind(mjh_reader *r)
{
	r->p = r->orig;
	r->end = r->orig_end;
}

long mjh_reader_unread(const mjh_reader *r) { return (long)(r->end - r->p); }

/* common.c:30-40 */
static inline unsigned rd8(mjh_reader *r)
{
	if (r->p < r->end)
		return *r->p++;
	if (r->from_callbacks) {
		reader_refill(r);
		return *r->p++;
	}
	return 0;
}

/* common.c:45-58 */
static int rd_eof(mjh_reader *r)
{
	if (r->io.read) {
		if (!r->io.eof(r->user))
			return 0;
		if (r->from_callbacks == 0)
			return 1;
	}
	return r->p >= r->end;
}

/* common.c:64-84 */
static void rd_skip(mjh_reader *r, int n)
{
	if (n == 0)
		return;
	if (n < 0) {
		r->p = r->end;
		return;
	}
	if (r->io.read) {
		int have = (int)(r->end - r->p);
		if (have < n) {
			r->p = r->end;
			r->io.skip(r->user, n - have);
			return;
		}
	}
	/* a memory source may step past its end: every later read then yields 0 */
	if ((long)(r->end - r->p) < (long)n)
		r->p = r->end;
	else
		r->p += n;
}

static int rd16be(mjh_reader *r)
{
	int hi = (int)rd8(r);
	return (hi << 8) + (int)rd8(r);
}

/* ------------------------------------------------------------------ tables */

static int fail(mjh_decoder *d, const char *why)
{
	d->reason = why;
	return 0;
}

/* zigzag index -> natural index, with the reference's 15 spill entries (codec/jpeg.c:293-305) */
static const uint8_t k_dezigzag[64 + 15] = {
	0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5,
	12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
	35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
	58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
	63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

/* zigzag index -> int16 offset inside the block's tile slot: (P>>3)*512 + (P&7) */
static uint16_t k_tile_off[64 + 15];
static uint8_t k_pos_of_zig[64 + 15]; /* zigzag index -> in-block position P = 8*column + slot (the escape bytes' order) */
static uint8_t k_zig_of_pos[64]; /* in-tile position P = 8*chunk + slot -> zigzag index */
static int k_tile_off_ready;
static int k_have_pdep; /* the CPU has BMI2 pdep and popcnt at full speed: the refinement scans take their correction bits a run at a time */
static void init_zig_masks(void);

static void init_tile_off(void)
{
	int k;
	if (k_tile_off_ready)
		return;
	for (k = 0; k < 64 + 15; ++k) {
		int nat = k_dezigzag[k];
		int P = 8 * (nat & 7) + mij_rowslot[nat >> 3];
		k_tile_off[k] = (uint16_t)(((P >> 3) << 9) + (P & 7));
		k_pos_of_zig[k] = (uint8_t)P;
		if (k < 64)
			k_zig_of_pos[P] = (uint8_t)k;
	}
	init_zig_masks();
	__builtin_cpu_init();
	/* pdep is microcoded (hundreds of cycles) before Zen 3; MIJ_NO_PDEP=1 forces the bit-at-a-time loop (tests compare the two) */
	k_have_pdep = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("popcnt") && !__builtin_cpu_is("znver1") && !__builtin_cpu_is("znver2") &&
					  !(getenv("MIJ_NO_PDEP") && getenv("MIJ_NO_PDEP")[0] == '1');
	k_tile_off_ready = 1;
}

static inline size_t block_base(int L) { return ((size_t)(L >> 6) << 12) + ((size_t)(L & 63) << 3); }

/* A progressive scan walks a plane of tens of megabytes block after block; a block is eight 16-byte rows 1 KiB apart, four neighbouring blocks
 * share their cache lines.  Every fourth block asks for the lines the same blocks of the tile MJH_PREFETCH_TILES further on will need.
 * (Touching memory behind the plane's end is harmless: prefetches never fault.) */
#ifndef MJH_PREFETCH_TILES
#define MJH_PREFETCH_TILES 1
#endif
static inline void prefetch_tile_ahead(const int16_t *blk, int L)
{
#if MJH_PREFETCH_TILES
	if (!(L & 3)) {
		const char *p = (const char *)(blk + 4096 * MJH_PREFETCH_TILES);
		int c;
		for (c = 0; c < 8; ++c)
			_mm_prefetch(p + 1024 * c, _MM_HINT_T0);
	}
#else
	(void)blk;
	(void)L;
#endif
}

/* codec/jpeg.c:88-134 */
static int build_huffman(mjh_decoder *d, mjh_huff *h, const int *count)
{
	int i, j, k = 0;
	unsigned code = 0;
	for (i = 0; i < 16; ++i)
		for (j = 0; j < count[i]; ++j) {
			if (k >= 256)
				return fail(d, "bad code lengths"); /* the reference would overrun its arrays here */
			h->size[k++] = (uint8_t)(i + 1);
		}
	h->size[k] = 0;

	k = 0;
	for (j = 1; j <= 16; ++j) {
		h->delta[j] = k - (int)code;
		if (h->size[k] == j) {
			while (h->size[k] == j)
				h->code[k++] = (uint16_t)(code++);
			if (code - 1 >= (1u << j))
				return fail(d, "bad code lengths");
		}
		h->maxcode[j] = code << (16 - j);
		code <<= 1;
	}
	h->maxcode[17] = 0xffffffffu;

	memset(h->fast, 255, sizeof(h->fast));
	for (i = 0; i < k; ++i) {
		int s = h->size[i];
		if (s <= MJH_FAST_BITS) {
			int first = h->code[i] << (MJH_FAST_BITS - s);
			int span = 1 << (MJH_FAST_BITS - s);
			memset(h->fast + first, i, (size_t)span); /* symbol index 255 keeps meaning "slow path" */
		}
	}
	return 1;
}

/* codec/jpeg.c:138-165 */
static void build_fast_ac(int16_t *fac, const mjh_huff *h)
{
	int i;
	for (i = 0; i < (1 << MJH_FAST_BITS); ++i) {
		int sym = h->fast[i];
		fac[i] = 0;
		if (sym < 255) {
			int rs = h->values[sym];
			int run = (rs >> 4) & 15, magbits = rs & 15, len = h->size[sym];
			if (magbits && len + magbits <= MJH_FAST_BITS) {
				int v = ((i << len) & ((1 << MJH_FAST_BITS) - 1)) >> (MJH_FAST_BITS - magbits);
				if (v < (1 << (magbits - 1)))
					v += (int)((~0u << magbits) + 1);
				if (v >= -128 && v <= 127)
					fac[i] = (int16_t)(v * 256 + run * 16 + len + magbits);
			}
		}
	}
}

/* The symbols of an AC refinement scan are (run, size 1) followed by one sign bit, ZRL and EOB runs (codec/jpeg.c:507-545).  For the 9-bit
 * window i: when the code in front is one of "(r, 1) + its sign bit inside the window", "ZRL" or "EOB0" the entry is
 *     length taken (code, plus the sign bit) | r << 4 | kind << 8 | sign << 10     kind 1 = coefficient, 2 = ZRL, 3 = EOB0; 0 = not here
 * so that the refinement loop needs one lookup where huff_decode_r + get_bit_r take four dependent ones.  Anything else (long codes, EOB
 * runs with extra bits, sizes other than 1 -- which the reference rejects) leaves the entry 0 and goes the ordinary way. */
static void build_fast_refine(uint16_t *fr, const mjh_huff *h)
{
	int i;
	for (i = 0; i < (1 << MJH_FAST_BITS); ++i) {
		const int sym = h->fast[i];
		fr[i] = 0;
		if (sym < 255) {
			const int rs = h->values[sym], r = (rs >> 4) & 15, s = rs & 15, len = h->size[sym];
			if (s == 1 && len + 1 <= MJH_FAST_BITS)
				fr[i] = (uint16_t)((len + 1) | (r << 4) | (1 << 8) | (((i >> (MJH_FAST_BITS - 1 - len)) & 1) << 10));
			else if (s == 0 && r == 15)
				fr[i] = (uint16_t)(len | (15 << 4) | (2 << 8));
			else if (s == 0 && r == 0)
				fr[i] = (uint16_t)(len | (3 << 8));
		}
	}
}

/* ------------------------------------------------------------------ bit reader */

static const uint32_t k_bmask[17] = {0, 1, 3, 7, 15, 31, 63, 127, 255, 511, 1023, 2047, 4095, 8191, 16383, 32767, 65535};
/* (-1 << n) + 1; the reference's table stops at n = 15 (codec/jpeg.c:246) */
static const int32_t k_bias[17] = {0, -1, -3, -7, -15, -31, -63, -127, -255, -511, -1023, -2047, -4095, -8191, -16383, -32767, -65535};

#define MARKER_NONE 0xff

/* codec/jpeg.c:167-187 */
static void bits_grow(mjh_decoder *d)
{
	/* Several bytes at once when none of them is 0xff and no marker has been seen: the same bytes land
	 * at the same positions as in the byte loop below (bits under code_bits are zero by construction). */
	if (!d->nomore && d->code_bits >= 0 && d->code_bits <= 24 && d->r->end - d->r->p >= 4) {
		const uint8_t *p = d->r->p;
		uint32_t w = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
		int n = (32 - d->code_bits) >> 3; /* bytes the loop below would take: 1..4 */
		uint32_t keep = w & (0xffffffffu << (32 - 8 * n));
		uint32_t y = ~keep; /* a zero byte in y <=> one of the n stream bytes is 0xff */
		if (!((y - 0x01010101u) & ~y & 0x80808080u)) {
			d->code_buffer |= keep >> d->code_bits;
			d->code_bits += 8 * n;
			d->r->p = p + n;
			return;
		}
	}
	do {
		unsigned b = d->nomore ? 0 : rd8(d->r);
		if (b == 0xff) {
			unsigned c = rd8(d->r);
			while (c == 0xff)
				c = rd8(d->r);
			if (c != 0) {
				d->marker = (unsigned char)c;
				d->nomore = 1;
				return;
			}
		}
		/* code_bits can be negative on streams that ran dry at a marker; the byte is then 0.  It can also exceed 24
		 * here: a DC "category" above 16 from a damaged table makes the reference refill with a full buffer
		 * (codec/jpeg.c:254), its shift count goes negative and x86 takes it modulo 32 -- spelled out */
		if (b)
			d->code_buffer |= b << ((24 - d->code_bits) & 31);
		d->code_bits += 8;
	} while (d->code_bits <= 24);
}

static inline uint32_t rotl32(uint32_t x, int n) { return (x << (n & 31)) | (x >> ((32 - n) & 31)); }

/* The bit register of stbi__jpeg (code_buffer / code_bits, codec/jpeg.c:69-70) held in locals across a
 * block: the decoder struct is only touched when more bytes are needed. */
typedef struct {
	uint32_t buf;
	int bits;
} bitreg;

static inline bitreg reg_load(const mjh_decoder *d)
{
	bitreg b;
	b.buf = d->code_buffer;
	b.bits = d->code_bits;
	return b;
}

static inline void reg_store(mjh_decoder *d, const bitreg *b)
{
	d->code_buffer = b->buf;
	d->code_bits = b->bits;
}

static inline void reg_grow(mjh_decoder *d, bitreg *b)
{
	/* bits_grow's several-bytes-at-once case on the local register */
	mjh_reader *r = d->r;
	if (!d->nomore && (unsigned)b->bits <= 24 && r->end - r->p >= 4) {
		uint32_t w, keep, y;
		int n = (32 - b->bits) >> 3;
		memcpy(&w, r->p, 4);
		w = __builtin_bswap32(w);
		keep = w & (0xffffffffu << (32 - 8 * n));
		y = ~keep;
		if (!((y - 0x01010101u) & ~y & 0x80808080u)) {
			b->buf |= keep >> b->bits;
			b->bits += 8 * n;
			r->p += n;
			return;
		}
	}
	reg_store(d, b);
	bits_grow(d);
	*b = reg_load(d);
}

/* codec/jpeg.c:193-243 */
static inline int huff_decode_r(mjh_decoder *d, const mjh_huff *h, bitreg *b)
{
	unsigned top, temp;
	int k, c;
	if (b->bits < 16)
		reg_grow(d, b);
	top = b->buf >> (32 - MJH_FAST_BITS);
	k = h->fast[top];
	if (k < 255) {
		int s = h->size[k];
		if (s > b->bits)
			return -1;
		b->buf <<= s;
		b->bits -= s;
		return h->values[k];
	}
	temp = b->buf >> 16;
	for (k = MJH_FAST_BITS + 1;; ++k)
		if (temp < h->maxcode[k])
			break;
	if (k == 17) {
		b->bits -= 16;
		return -1;
	}
	if (k > b->bits)
		return -1;
	c = (int)((b->buf >> (32 - k)) & k_bmask[k]) + h->delta[k];
	b->bits -= k;
	b->buf <<= k;
	return h->values[c & 255];
}

/* codec/jpeg.c:250-265 */
static inline int extend_receive_r(mjh_decoder *d, int n, bitreg *b)
{
	uint32_t k;
	int32_t sgn;
	if (b->bits < n)
		reg_grow(d, b);
	if (n < 0 || n > 16)
		return 0;
	sgn = (int32_t)b->buf >> 31;
	k = rotl32(b->buf, n);
	b->buf = k & ~k_bmask[n];
	k &= k_bmask[n];
	b->bits -= n;
	return (int)k + (k_bias[n] & ~sgn);
}

/* reg_grow on a bit register held in two plain locals (the AC refinement loop: through the bitreg struct the compiler keeps the register
 * in memory there.  The baseline block loops were tried this way too and came out 8 % SLOWER than with the struct and huff_decode_r out
 * of line -- measured, 1080p q=90, so they stay as they are). */
static inline __attribute__((always_inline)) void grow_s(mjh_decoder *d, uint32_t *buf, int *bits)
{
	mjh_reader *r = d->r;
	if (!d->nomore && (unsigned)*bits <= 24 && r->end - r->p >= 4) {
		uint32_t w, keep, y;
		const int n = (32 - *bits) >> 3;
		memcpy(&w, r->p, 4);
		w = __builtin_bswap32(w);
		keep = w & (0xffffffffu << (32 - 8 * n));
		y = ~keep;
		if (!((y - 0x01010101u) & ~y & 0x80808080u)) {
			*buf |= keep >> *bits;
			*bits += 8 * n;
			r->p += n;
			return;
		}
	}
	d->code_buffer = *buf;
	d->code_bits = *bits;
	bits_grow(d);
	*buf = d->code_buffer;
	*bits = d->code_bits;
}

/* ------------------------------------------------------------------ block decoders */

static void zero_block(int16_t *blk)
{
	int c;
	for (c = 0; c < 8; ++c)
		memset(blk + (c << 9), 0, 16);
}

static inline int iabs16(int v)
{
	int s = (int16_t)v;
	return s < 0 ? -s : s;
}

/*
 * Baseline block (codec/jpeg.c:308-370).  blk points at the block's tile slot; qz is the
 * component's quantisation table in zigzag order (only used for the WIDE_IDCT bound: the
 * multiply itself happens on the GPU).
 */
static int decode_block(mjh_decoder *d, int16_t *blk, const mjh_huff *hdc, const mjh_huff *hac, const int16_t *fac, mjh_comp *cp,
								const uint16_t *qz)
{
	int diff, dc, k, t;
	int32_t l1;
	bitreg b = reg_load(d);

	/* the reference refills here and again on entry to its Huffman routine (:313, :197): two calls
	 * differ from one when the first stopped at a marker, so both stay */
	if (b.bits < 16)
		reg_grow(d, &b);
	t = huff_decode_r(d, hdc, &b);
	if (t < 0) {
		reg_store(d, &b);
		return fail(d, "bad huffman code");
	}
	if (cp->touched)
		zero_block(blk);

	diff = t ? extend_receive_r(d, t, &b) : 0;
	dc = (int)((unsigned)cp->dc_pred + (unsigned)diff);
	cp->dc_pred = dc;
	blk[0] = (int16_t)dc;
	l1 = iabs16((int)((unsigned)dc * qz[0]));

	k = 1;
	do {
		int c, r, s;
		if (b.bits < 16)
			reg_grow(d, &b);
		c = (int)(b.buf >> (32 - MJH_FAST_BITS));
		r = fac[c];
		if (r) {
			k += (r >> 4) & 15;
			s = r & 15;
			b.buf <<= s;
			b.bits -= s;
			blk[k_tile_off[k]] = (int16_t)(r >> 8);
			l1 += iabs16((r >> 8) * qz[k]);
			++k;
		} else {
			int rs = huff_decode_r(d, hac, &b);
			if (rs < 0) {
				reg_store(d, &b);
				return fail(d, "bad huffman code");
			}
			s = rs & 15;
			r = rs >> 4;
			if (s == 0) {
				if (rs != 0xf0)
					break;
				k += 16;
			} else {
				int v;
				k += r;
				v = extend_receive_r(d, s, &b);
				blk[k_tile_off[k]] = (int16_t)v;
				l1 += iabs16((int)((unsigned)v * qz[k]));
				++k;
			}
		}
	} while (k < 64);
	reg_store(d, &b);
	if (l1 > d->max_block_l1)
		d->max_block_l1 = l1;
	return 1;
}

/*
 * The same block into COMPACT planes (mij.h, "compact coefficient planes"): the low byte of every coefficient at the byte offset
 * the int16 layout has as element offset, the DC term in the component's int16 array (the byte in its place holds the block's flags),
 * and -- only for a block that holds a coefficient outside -128..127 -- 64 escape bytes h with coefficient == sext8(low) + 256 * h.
 * The fast-AC table only holds values inside a byte (build_fast_ac), so only the slow path can escape.  Same symbols, same order,
 * same failure points as decode_block: the planes k_pack_c8 would make from decode_block's output, byte for byte (tests).
 */
static int decode_block_c8(mjh_decoder *d, mjh_comp *cp, int L, const mjh_huff *hdc, const mjh_huff *hac, const int16_t *fac, const uint16_t *qz)
{
	uint8_t *blk = cp->lo8 + block_base(L), *esc = NULL;
	int diff, dc, k, t;
	int32_t l1;
	bitreg b = reg_load(d);

	if (b.bits < 16)
		reg_grow(d, &b);
	t = huff_decode_r(d, hdc, &b);
	if (t < 0) {
		reg_store(d, &b);
		return fail(d, "bad huffman code");
	}
	if (cp->touched) {
		int c;
		for (c = 0; c < 8; ++c)
			memset(blk + (c << 9), 0, 8);
	}

	diff = t ? extend_receive_r(d, t, &b) : 0;
	dc = (int)((unsigned)cp->dc_pred + (unsigned)diff);
	cp->dc_pred = dc;
	cp->dc16[L] = (int16_t)dc;
	l1 = iabs16((int)((unsigned)dc * qz[0]));

	k = 1;
	do {
		int c, r, s;
		if (b.bits < 16)
			reg_grow(d, &b);
		c = (int)(b.buf >> (32 - MJH_FAST_BITS));
		r = fac[c];
		if (r) {
			k += (r >> 4) & 15;
			s = r & 15;
			b.buf <<= s;
			b.bits -= s;
			blk[k_tile_off[k]] = (uint8_t)(r >> 8);
			if (esc) /* a spill position written twice: what was there may have had a high byte */
				esc[k_pos_of_zig[k]] = 0;
			l1 += iabs16((r >> 8) * qz[k]);
			++k;
		} else {
			int rs = huff_decode_r(d, hac, &b);
			if (rs < 0) {
				reg_store(d, &b);
				return fail(d, "bad huffman code");
			}
			s = rs & 15;
			r = rs >> 4;
			if (s == 0) {
				if (rs != 0xf0)
					break;
				k += 16;
			} else {
				int v;
				k += r;
				v = extend_receive_r(d, s, &b);
				blk[k_tile_off[k]] = (uint8_t)v;
				if (esc || v != (int8_t)v) {
					if (!esc) {
						esc = cp->hi8 + ((size_t)L << 6);
						memset(esc, 0, 64);
						blk[0] = 1; /* flags: escaped */
						d->any_escape = 1;
					}
					esc[k_pos_of_zig[k]] = (uint8_t)(((int16_t)v - (int8_t)v) >> 8);
				}
				l1 += iabs16((int)((unsigned)v * qz[k]));
				++k;
			}
		}
	} while (k < 64);
	reg_store(d, &b);
	if (l1 > d->max_block_l1)
		d->max_block_l1 = l1;
	return 1;
}

/* codec/jpeg.c:268-278 on the local register */
static inline int get_bits_r(mjh_decoder *d, int n, bitreg *b)
{
	uint32_t k;
	if (b->bits < n)
		reg_grow(d, b);
	k = rotl32(b->buf, n);
	b->buf = k & ~k_bmask[n];
	k &= k_bmask[n];
	b->bits -= n;
	return (int)k;
}

/* codec/jpeg.c:280-289 on the local register */
static inline int get_bit_r(mjh_decoder *d, bitreg *b)
{
	uint32_t k;
	if (b->bits < 1)
		reg_grow(d, b);
	k = b->buf;
	b->buf <<= 1;
	--b->bits;
	return (int)(k & 0x80000000u);
}

/* codec/jpeg.c:372-402 */
static int decode_block_prog_dc(mjh_decoder *d, int16_t *blk, const mjh_huff *hdc, mjh_comp *cp)
{
	bitreg b;
	if (d->spec_end != 0)
		return fail(d, "can't merge dc and ac");
	b = reg_load(d);
	if (b.bits < 16)
		reg_grow(d, &b);
	if (d->succ_high == 0) {
		int t, diff, dc;
		zero_block(blk);
		t = huff_decode_r(d, hdc, &b);
		if (t < 0) {
			reg_store(d, &b);
			return fail(d, "can't merge dc and ac");
		}
		diff = t ? extend_receive_r(d, t, &b) : 0;
		dc = (int)((unsigned)cp->dc_pred + (unsigned)diff);
		cp->dc_pred = dc;
		blk[0] = (int16_t)((unsigned)dc << d->succ_low);
	} else {
		if (get_bit_r(d, &b))
			blk[0] = (int16_t)(blk[0] + (int16_t)(1 << d->succ_low));
	}
	reg_store(d, &b);
	return 1;
}

/* codec/jpeg.c:497-505: "if (stbi__jpeg_get_bit(j)) if ((*p & bit) == 0) { if (*p > 0) *p += bit; else *p -= bit; }" -- without
 * data-dependent branches (the correction bits of a photograph are coin flips: as branches they mispredict every other time) */
static inline void refine_nonzero(mjh_decoder *d, int16_t *p, int bit, bitreg *b)
{
	const int take = get_bit_r(d, b) != 0;
	const int v = *p;
	const int need = take & ((v & bit) == 0);
	const int delta = v > 0 ? bit : -bit;
	*p = (int16_t)(v + (need ? delta : 0));
}

/* non-zero map of a block in zigzag order: eight 16-byte chunks (one per column) compared against zero give the map in position
 * order P; the fixed bit permutation P -> zigzag index goes through eight 256-entry tables (one per byte of the map: 16 KiB, built
 * once) instead of a loop over the set bits (a luma block of a photograph has twenty of them) */
static uint64_t k_zig_mask_of_byte[8][256];
static void init_zig_masks(void)
{
	int c, v, bit;
	for (c = 0; c < 8; ++c)
		for (v = 0; v < 256; ++v) {
			uint64_t m = 0;
			for (bit = 0; bit < 8; ++bit)
				if (v & (1 << bit))
					m |= 1ull << k_zig_of_pos[8 * c + bit];
			k_zig_mask_of_byte[c][v] = m;
		}
}
static inline uint64_t block_nonzero_mask(const int16_t *blk)
{
	const __m128i zero = _mm_setzero_si128();
	uint64_t zm = 0;
	int c;
	for (c = 0; c < 8; ++c) {
		const __m128i v = _mm_loadu_si128((const __m128i *)(blk + (c << 9)));
		const __m128i eq = _mm_cmpeq_epi16(v, zero);
		const unsigned m8 = (unsigned)_mm_movemask_epi8(_mm_packs_epi16(eq, zero)) & 0xffu; /* 1 = zero */
		zm |= k_zig_mask_of_byte[c][m8 ^ 0xffu];
	}
	return zm;
}

static inline uint64_t band_mask(int lo, int hi) { return (hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1)) & ~((1ull << lo) - 1); }

/* pdep / popcnt through inline assembly: the library is built for plain x86-64 and executes these only where init_tile_off found them */
static inline uint64_t pdep64(uint64_t src, uint64_t mask)
{
	uint64_t r;
	__asm__("pdep %2, %1, %0" : "=r"(r) : "r"(src), "r"(mask));
	return r;
}
static inline int popcnt64(uint64_t x)
{
	uint64_t r;
	__asm__("popcnt %1, %0" : "=r"(r) : "r"(x) : "cc");
	return (int)r;
}
static inline uint64_t bitrev64(uint64_t x)
{
	x = __builtin_bswap64(x);
	x = ((x & 0x0f0f0f0f0f0f0f0full) << 4) | ((x >> 4) & 0x0f0f0f0f0f0f0f0full);
	x = ((x & 0x3333333333333333ull) << 2) | ((x >> 2) & 0x3333333333333333ull);
	x = ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);
	return x;
}

/* The correction bits of the already non-zero coefficients `where` (a zigzag-order mask, not empty), which the reference reads one
 * get_bit per coefficient in ascending order (codec/jpeg.c:497-505, :536-553): all of them in one or a few reads, the first bit read
 * deposited at the lowest position.  Returns the mask of the coefficients whose bit was set.  The bit register is a first-in-first-out
 * of the stream's bits followed by zeros once the data has run out, so reading n bits at once returns what n single reads return. */
static inline uint64_t read_corrections(mjh_decoder *d, uint64_t where, bitreg *b)
{
	const int n = popcnt64(where);
	uint64_t bits = 0;
	int left = n;
	if (n == 1)
		return get_bit_r(d, b) ? where : 0;
	while (left > 16) {
		bits = (bits << 16) | (uint64_t)get_bits_r(d, 16, b);
		left -= 16;
	}
	bits = (bits << left) | (uint64_t)get_bits_r(d, left, b);
	return pdep64(bitrev64(bits) >> (64 - n), where);
}

/* codec/jpeg.c:499-504 for every coefficient of `corr` */
static inline void apply_corrections(int16_t *blk, uint64_t corr, int bit)
{
	while (corr) {
		int16_t *p = &blk[k_tile_off[__builtin_ctzll(corr)]];
		const int v = *p; /* not zero: the mask was made from the block */
		const int sgn = v >> 31;
		const int delta = (bit ^ sgn) - sgn; /* v > 0: +bit, v < 0: -bit */
		*p = (int16_t)(v + ((v & bit) == 0 ? delta : 0));
		corr &= corr - 1;
	}
}

/* A symbol of an AC refinement scan that is not in the one-lookup table (a long code, an EOB run with extra bits, a size the reference
 * rejects): codec/jpeg.c:507-533 on the decoder's own bit register.  Returns the run (64: to the end of the band) | what the symbol
 * puts there (1: +bit, 2: -bit, 0: nothing) << 8, or -1 for a bad code. */
static __attribute__((noinline)) int refine_symbol_slow(mjh_decoder *d, const mjh_huff *hac)
{
	bitreg b = reg_load(d);
	int rs = huff_decode_r(d, hac, &b), r, s;
	if (rs < 0) {
		reg_store(d, &b);
		(void)fail(d, "bad huffman code");
		return -1;
	}
	s = rs & 15;
	r = rs >> 4;
	if (s == 0) {
		if (r < 15) {
			d->eob_run = (1 << r) - 1;
			if (r)
				d->eob_run += get_bits_r(d, r, &b);
			r = 64; /* run to the end of the band */
		}
		/* r == 15: sixteen zeros, a run with s = 0 */
	} else {
		if (s != 1) {
			reg_store(d, &b);
			(void)fail(d, "bad huffman code");
			return -1;
		}
		s = get_bit_r(d, &b) ? 1 : 2;
	}
	reg_store(d, &b);
	return r | (s << 8);
}

/* The symbols of one block of an AC refinement scan (codec/jpeg.c:507-553) where the CPU has pdep: the target of a run is a select on the
 * zero map, the correction bits of the coefficients passed on the way come in one read.  The bit register is two plain locals here
 * (through the bitreg struct the compiler kept it on the stack, a store-to-load round trip on the chain from one symbol to the next).
 * Returns 0 on a bad code (d->reason set); *corr_out = the coefficients whose correction bit was set.  (Tried and dropped: topping the bit
 * register up at every symbol instead of when it runs low, and reading the correction bits without a branch -- both slower, the loop is
 * bound by its instruction count, not by mispredicted branches.) */
static __attribute__((noinline)) int refine_symbols_wide(mjh_decoder *d, int16_t *blk, const mjh_huff *hac, const uint16_t *fref, uint64_t nz, uint64_t zero, int bit, uint64_t *corr_out)
{
	const int spec_end = d->spec_end;
	uint32_t buf = d->code_buffer;
	int bits = d->code_bits;
	uint64_t corr = 0;
	int k = d->spec_start;
	do {
		int r, s;
		unsigned e;
		if (bits < 16) /* where huff_decode_r refills */
			grow_s(d, &buf, &bits);
		e = fref[buf >> (32 - MJH_FAST_BITS)];
		if (e && (int)(e & 15u) <= bits) { /* the code and, for a coefficient, its sign bit in one step (build_fast_refine) */
			const int len = (int)(e & 15u), kind = (int)(e >> 8) & 3;
			const int sv = (e & 0x400u) ? bit : -bit;
			buf <<= len;
			bits -= len;
			r = kind == 3 ? 64 : (int)(e >> 4) & 15; /* EOB0 (the run count stays 0): to the end of the band */
			s = kind == 1 ? sv : 0;
		} else {
			int rs;
			d->code_buffer = buf;
			d->code_bits = bits;
			rs = refine_symbol_slow(d, hac);
			if (rs < 0)
				return 0;
			r = rs & 255;
			s = (rs >> 8) == 1 ? bit : ((rs >> 8) == 2 ? -bit : 0);
			buf = d->code_buffer;
			bits = d->code_bits;
		}
		{
			/* skip r zero-history coefficients, refining the non-zero ones passed on the way, then put s into the next zero one */
			const uint64_t from = ~0ull << k; /* k <= spec_end <= 63 */
			const uint64_t hit = pdep64(r < 64 ? 1ull << r : 0, zero & from); /* the (r+1)-th zero-history position from k on, if the band has one */
			uint64_t passed = nz & from;
			if (hit) {
				const int t = __builtin_ctzll(hit);
				passed &= hit - 1;
				blk[k_tile_off[t]] = (int16_t)s;
				k = t + 1;
			} else /* the run leaves the band: everything up to its end is refined, s is dropped (:536-553 ends the same way) */
				k = spec_end + 1;
			if (passed) {
				/* read_corrections on the locals: the first bit read belongs to the lowest coefficient */
				int n = popcnt64(passed);
				uint64_t v = 0;
				while (n > 16) {
					if (bits < 16)
						grow_s(d, &buf, &bits);
					v = (v << 16) | (buf >> 16);
					buf <<= 16;
					bits -= 16;
					n -= 16;
				}
				if (bits < n)
					grow_s(d, &buf, &bits);
				v = (v << n) | (uint64_t)(((uint64_t)buf << n) >> 32);
				buf <<= n;
				bits -= n;
				n = popcnt64(passed);
				corr |= pdep64(bitrev64(v) >> (64 - n), passed);
			}
		}
	} while (k <= spec_end);
	d->code_buffer = buf;
	d->code_bits = bits;
	*corr_out = corr;
	return 1;
}

/* codec/jpeg.c:406-558 */
static int decode_block_prog_ac(mjh_decoder *d, int16_t *blk, const mjh_huff *hac, const int16_t *fac, const uint16_t *fref)
{
	const int spec_end = d->spec_end;
	int k;
	bitreg b;
	if (d->spec_start == 0)
		return fail(d, "can't merge dc and ac");

	if (d->succ_high == 0) {
		int shift = d->succ_low;
		if (d->eob_run) {
			--d->eob_run;
			return 1;
		}
		b = reg_load(d);
		k = d->spec_start;
		do {
			int c, r, s;
			if (b.bits < 16)
				reg_grow(d, &b);
			c = (int)(b.buf >> (32 - MJH_FAST_BITS));
			r = fac[c];
			if (r) {
				k += (r >> 4) & 15;
				s = r & 15;
				b.buf <<= s;
				b.bits -= s;
				blk[k_tile_off[k]] = (int16_t)((unsigned)(r >> 8) << shift);
				++k;
			} else {
				int rs = huff_decode_r(d, hac, &b);
				if (rs < 0) {
					reg_store(d, &b);
					return fail(d, "bad huffman code");
				}
				s = rs & 15;
				r = rs >> 4;
				if (s == 0) {
					if (r < 15) {
						d->eob_run = (1 << r);
						if (r)
							d->eob_run += get_bits_r(d, r, &b);
						--d->eob_run;
						break;
					}
					k += 16;
				} else {
					k += r;
					blk[k_tile_off[k]] = (int16_t)((unsigned)extend_receive_r(d, s, &b) << shift);
					++k;
				}
			}
		} while (k <= spec_end);
		reg_store(d, &b);
	} else {
		int bit = (int16_t)(1 << d->succ_low);
		/* bit k set <=> coefficient k (zigzag order) of this block is already non-zero.  The reference
		 * walks all positions of the band and tests each (codec/jpeg.c:497-505, :536-553); visiting only
		 * the set bits, in the same ascending order, reads the same correction bits for the same
		 * coefficients.  Coefficients placed by this call lie behind the walk and are never revisited.
		 * Where the CPU has pdep (k_have_pdep) the walk is not a walk at all: the target of a run of r zero-history coefficients is
		 * the (r+1)-th set bit of the zero map, the non-zero ones passed on the way are a mask, their correction bits are read together
		 * (read_corrections) and applied once the block's symbols are through -- corrections touch history coefficients, placements
		 * zero-history ones, so the order between them does not matter. */
		const uint64_t band = band_mask(d->spec_start, spec_end);
		uint64_t nz = block_nonzero_mask(blk) & band;
		uint64_t corr = 0;
		if (k_have_pdep) {
			if (d->eob_run) {
				--d->eob_run;
				if (nz) {
					b = reg_load(d);
					corr = read_corrections(d, nz, &b);
					reg_store(d, &b);
				}
			} else if (!refine_symbols_wide(d, blk, hac, fref, nz, ~nz & band, bit, &corr))
				return 0;
			if (corr)
				apply_corrections(blk, corr, bit);
			return 1;
		}
		b = reg_load(d);
		if (d->eob_run) {
			--d->eob_run;
			while (nz) {
				refine_nonzero(d, &blk[k_tile_off[__builtin_ctzll(nz)]], bit, &b);
				nz &= nz - 1;
			}
		} else {
			k = d->spec_start;
			do {
				int r, s;
				unsigned e;
				if (b.bits < 16) /* where huff_decode_r refills */
					reg_grow(d, &b);
				e = fref[b.buf >> (32 - MJH_FAST_BITS)];
				if (e && (int)(e & 15u) <= b.bits) { /* the code and, for a coefficient, its sign bit in one step (build_fast_refine) */
					const int len = (int)(e & 15u), kind = (int)(e >> 8) & 3;
					b.buf <<= len;
					b.bits -= len;
					r = (int)(e >> 4) & 15;
					s = 0;
					if (kind == 1)
						s = (e & 0x400u) ? bit : -bit;
					else if (kind == 3) {
						d->eob_run = 0;
						r = 64; /* run to the end of the band */
					}
				} else {
					int rs = huff_decode_r(d, hac, &b);
					if (rs < 0) {
						reg_store(d, &b);
						return fail(d, "bad huffman code");
					}
					s = rs & 15;
					r = rs >> 4;
					if (s == 0) {
						if (r < 15) {
							d->eob_run = (1 << r) - 1;
							if (r)
								d->eob_run += get_bits_r(d, r, &b);
							r = 64; /* run to the end of the band */
						}
						/* r == 15: sixteen zeros, handled by the run below with s = 0 */
					} else {
						if (s != 1) {
							reg_store(d, &b);
							return fail(d, "bad huffman code");
						}
						s = get_bit_r(d, &b) ? bit : -bit;
					}
				}
				/* skip r zero-history coefficients, refining the non-zero ones passed on the way, then put
				 * s into the next zero one */
				while (k <= spec_end) {
					const uint64_t ahead = nz >> k;
					const int gap = ahead ? __builtin_ctzll(ahead) : 64;
					const int avail = gap < spec_end + 1 - k ? gap : spec_end + 1 - k;
					if (r < avail) {
						k += r;
						blk[k_tile_off[k]] = (int16_t)s;
						++k;
						break;
					}
					r -= avail;
					k += avail;
					if (k > spec_end)
						break;
					refine_nonzero(d, &blk[k_tile_off[k]], bit, &b);
					++k;
				}
			} while (k <= spec_end);
		}
		reg_store(d, &b);
	}
	return 1;
}

/* ------------------------------------------------------------------ scans */

#define IS_RESTART(x) ((x) >= 0xd0 && (x) <= 0xd7)

/* codec/jpeg.c:1142-1153 */
static void entropy_reset(mjh_decoder *d)
{
	d->code_bits = 0;
	d->code_buffer = 0;
	d->nomore = 0;
	d->comp[0].dc_pred = d->comp[1].dc_pred = d->comp[2].dc_pred = d->comp[3].dc_pred = 0;
	d->marker = MARKER_NONE;
	d->todo = d->restart_interval ? d->restart_interval : 0x7fffffff;
	d->eob_run = 0;
}

/* after each MCU: count the restart interval down (codec/jpeg.c:1180-1189 and twins).
 * Returns 0 to keep going, 1 when the scan must stop here (missing RST: keep what we have). */
static inline int restart_check(mjh_decoder *d)
{
	if (--d->todo <= 0) {
		if (d->code_bits < 24)
			bits_grow(d);
		if (!IS_RESTART(d->marker))
			return 1;
		entropy_reset(d);
	}
	return 0;
}

/* codec/jpeg.c:1155-1317 */
static int parse_entropy_coded_data(mjh_decoder *d)
{
	uint16_t qz[4][64 + 15];
	int ci, k;
	entropy_reset(d);
	for (ci = 0; ci < d->scan_n; ++ci) {
		const uint16_t *q = d->dequant[d->comp[d->order[ci]].tq];
		for (k = 0; k < 64 + 15; ++k)
			qz[ci][k] = q[k_dezigzag[k]];
	}

	if (!d->progressive) {
		if (d->scan_n == 1) {
			mjh_comp *cp = &d->comp[d->order[0]];
			int w = (cp->x + 7) >> 3, h = (cp->y + 7) >> 3, i, j;
			const mjh_huff *hdc = &d->huff_dc[cp->hd], *hac = &d->huff_ac[cp->ha];
			const int16_t *fac = d->fast_ac[cp->ha];
			for (j = 0; j < h; ++j)
				for (i = 0; i < w; ++i) {
					if (!(d->compact ? decode_block_c8(d, cp, i + j * cp->bw, hdc, hac, fac, qz[0]) : decode_block(d, cp->plane + block_base(i + j * cp->bw), hdc, hac, fac, cp, qz[0])))
						return 0;
					if (restart_check(d)) {
						cp->touched = 1;
						return 1;
					}
				}
			cp->touched = 1;
			return 1;
		} else {
			int i, j, x, y;
			for (j = 0; j < d->mcu_y; ++j)
				for (i = 0; i < d->mcu_x; ++i) {
					for (ci = 0; ci < d->scan_n; ++ci) {
						mjh_comp *cp = &d->comp[d->order[ci]];
						const mjh_huff *hdc = &d->huff_dc[cp->hd], *hac = &d->huff_ac[cp->ha];
						const int16_t *fac = d->fast_ac[cp->ha];
						for (y = 0; y < cp->v; ++y)
							for (x = 0; x < cp->h; ++x) {
								int L = (i * cp->h + x) + (j * cp->v + y) * cp->bw;
								if (!(d->compact ? decode_block_c8(d, cp, L, hdc, hac, fac, qz[ci]) : decode_block(d, cp->plane + block_base(L), hdc, hac, fac, cp, qz[ci])))
									return 0;
							}
					}
					if (restart_check(d))
						goto interleaved_done;
				}
		interleaved_done:
			for (ci = 0; ci < d->scan_n; ++ci)
				d->comp[d->order[ci]].touched = 1;
			return 1;
		}
	} else {
		if (d->scan_n == 1) {
			mjh_comp *cp = &d->comp[d->order[0]];
			int w = (cp->x + 7) >> 3, h = (cp->y + 7) >> 3, i, j;
			for (j = 0; j < h; ++j)
				for (i = 0; i < w; ++i) {
					int16_t *blk = cp->plane + block_base(i + j * cp->bw);
					prefetch_tile_ahead(blk, i + j * cp->bw);
					if (d->spec_start == 0) {
						if (!decode_block_prog_dc(d, blk, &d->huff_dc[cp->hd], cp))
							return 0;
					} else {
						if (!decode_block_prog_ac(d, blk, &d->huff_ac[cp->ha], d->fast_ac[cp->ha], d->fast_refine[cp->ha]))
							return 0;
					}
					if (restart_check(d))
						return 1;
				}
			return 1;
		} else {
			int i, j, x, y;
			for (j = 0; j < d->mcu_y; ++j)
				for (i = 0; i < d->mcu_x; ++i) {
					for (ci = 0; ci < d->scan_n; ++ci) {
						mjh_comp *cp = &d->comp[d->order[ci]];
						for (y = 0; y < cp->v; ++y)
							for (x = 0; x < cp->h; ++x) {
								int L = (i * cp->h + x) + (j * cp->v + y) * cp->bw;
								if (!decode_block_prog_dc(d, cp->plane + block_base(L), &d->huff_dc[cp->hd], cp))
									return 0;
							}
					}
					if (restart_check(d))
						return 1;
				}
			return 1;
		}
	}
}

/* ------------------------------------------------------------------ markers */

/* codec/jpeg.c:1119-1134 */
static unsigned get_marker(mjh_decoder *d)
{
	unsigned x;
	if (d->marker != MARKER_NONE) {
		x = d->marker;
		d->marker = MARKER_NONE;
		return x;
	}
	x = rd8(d->r);
	if (x != 0xff)
		return MARKER_NONE;
	while (x == 0xff)
		x = rd8(d->r);
	return x;
}

/* codec/jpeg.c:1349-1468 */
static int process_marker(mjh_decoder *d, unsigned m)
{
	mjh_reader *r = d->r;
	int L;
	switch (m) {
	case MARKER_NONE:
		return fail(d, "expected marker");

	case 0xDD: /* DRI */
		if (rd16be(r) != 4)
			return fail(d, "bad DRI len");
		d->restart_interval = rd16be(r);
		return 1;

	case 0xDB: /* DQT */
		L = rd16be(r) - 2;
		while (L > 0) {
			int q = (int)rd8(r);
			int p = q >> 4, t = q & 15, i;
			if (p != 0 && p != 1)
				return fail(d, "bad DQT type");
			if (t > 3)
				return fail(d, "bad DQT table");
			for (i = 0; i < 64; ++i)
				d->dequant[t][k_dezigzag[i]] = (uint16_t)(p ? rd16be(r) : (int)rd8(r));
			L -= p ? 129 : 65;
		}
		return L == 0;

	case 0xC4: /* DHT */
		L = rd16be(r) - 2;
		while (L > 0) {
			int sizes[16], i, n = 0;
			int q = (int)rd8(r);
			int tc = q >> 4, th = q & 15;
			mjh_huff *h;
			if (tc > 1 || th > 3)
				return fail(d, "bad DHT header");
			for (i = 0; i < 16; ++i) {
				sizes[i] = (int)rd8(r);
				n += sizes[i];
			}
			L -= 17;
			h = tc == 0 ? &d->huff_dc[th] : &d->huff_ac[th];
			if (!build_huffman(d, h, sizes))
				return 0;
			for (i = 0; i < n; ++i)
				h->values[i] = (uint8_t)rd8(r);
			if (tc != 0) {
				build_fast_ac(d->fast_ac[th], h);
				build_fast_refine(d->fast_refine[th], h);
			}
			L -= n;
		}
		return L == 0;
	}

	if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
		L = rd16be(r);
		if (L < 2)
			return fail(d, m == 0xFE ? "bad COM len" : "bad APP len");
		L -= 2;
		if (m == 0xE0 && L >= 5) {
			static const unsigned char tag[5] = {'J', 'F', 'I', 'F', 0};
			int ok = 1, i;
			for (i = 0; i < 5; ++i)
				if (rd8(r) != tag[i])
					ok = 0;
			L -= 5;
			if (ok)
				d->jfif = 1;
		} else if (m == 0xEE && L >= 12) {
			static const unsigned char tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
			int ok = 1, i;
			for (i = 0; i < 6; ++i)
				if (rd8(r) != tag[i])
					ok = 0;
			L -= 6;
			if (ok) {
				rd8(r);    /* version */
				rd16be(r); /* flags0 */
				rd16be(r); /* flags1 */
				d->app14 = (int)rd8(r);
				L -= 6;
			}
		}
		rd_skip(r, L);
		return 1;
	}
	return fail(d, "unknown marker");
}

/* codec/jpeg.c:1471-1521 */
static int process_scan_header(mjh_decoder *d)
{
	mjh_reader *r = d->r;
	int i, aa;
	int Ls = rd16be(r);
	d->scan_n = (int)rd8(r);
	if (d->scan_n < 1 || d->scan_n > 4 || d->scan_n > d->img_n)
		return fail(d, "bad SOS component count");
	if (Ls != 6 + 2 * d->scan_n)
		return fail(d, "bad SOS len");
	for (i = 0; i < d->scan_n; ++i) {
		int id = (int)rd8(r), which;
		int q = (int)rd8(r);
		for (which = 0; which < d->img_n; ++which)
			if (d->comp[which].id == id)
				break;
		if (which == d->img_n)
			return 0; /* no reason set by the reference either */
		d->comp[which].hd = q >> 4;
		if (d->comp[which].hd > 3)
			return fail(d, "bad DC huff");
		d->comp[which].ha = q & 15;
		if (d->comp[which].ha > 3)
			return fail(d, "bad AC huff");
		d->order[i] = which;
	}
	d->spec_start = (int)rd8(r);
	d->spec_end = (int)rd8(r);
	aa = (int)rd8(r);
	d->succ_high = aa >> 4;
	d->succ_low = aa & 15;
	if (d->progressive) {
		if (d->spec_start > 63 || d->spec_end > 63 || d->spec_start > d->spec_end || d->succ_high > 13 || d->succ_low > 13)
			return fail(d, "bad SOS");
	} else {
		if (d->spec_start != 0)
			return fail(d, "bad SOS");
		if (d->succ_high != 0 || d->succ_low != 0)
			return fail(d, "bad SOS");
		d->spec_end = 63;
	}
	return 1;
}

static int mul_fits_int(int a, int b)
{
	if (a < 0 || b < 0)
		return 0;
	if (b == 0)
		return 1;
	return a <= INT_MAX / b;
}

/* codec/jpeg.c:1549-1659 */
static int process_frame_header(mjh_decoder *d, int mode)
{
	mjh_reader *r = d->r;
	int Lf, p, i, q, h_max = 1, v_max = 1, c;
	Lf = rd16be(r);
	if (Lf < 11)
		return fail(d, "bad SOF len");
	p = (int)rd8(r);
	if (p != 8)
		return fail(d, "only 8-bit");
	d->img_y = rd16be(r);
	if (d->img_y == 0)
		return fail(d, "no header height");
	d->img_x = rd16be(r);
	if (d->img_x == 0)
		return fail(d, "0 width");
	/* 16-bit fields cannot exceed STBI_MAX_DIMENSIONS (1<<24): the "too large" tests of :1565-1568 never fire */
	c = (int)rd8(r);
	if (c != 3 && c != 1 && c != 4)
		return fail(d, "bad component count");
	d->img_n = c;
	if (Lf != 8 + 3 * d->img_n)
		return fail(d, "bad SOF len");

	d->rgb = 0;
	for (i = 0; i < d->img_n; ++i) {
		static const unsigned char rgb[3] = {'R', 'G', 'B'};
		d->comp[i].id = (int)rd8(r);
		if (d->img_n == 3 && d->comp[i].id == rgb[i])
			++d->rgb;
		q = (int)rd8(r);
		d->comp[i].h = q >> 4;
		if (!d->comp[i].h || d->comp[i].h > 4)
			return fail(d, "bad H");
		d->comp[i].v = q & 15;
		if (!d->comp[i].v || d->comp[i].v > 4)
			return fail(d, "bad V");
		d->comp[i].tq = (int)rd8(r);
		if (d->comp[i].tq > 3)
			return fail(d, "bad TQ");
	}
	if (mode != MJH_SCAN_LOAD)
		return 1;

	if (!(mul_fits_int(d->img_x, d->img_y) && mul_fits_int(d->img_x * d->img_y, d->img_n)))
		return fail(d, "too large");

	for (i = 0; i < d->img_n; ++i) {
		if (d->comp[i].h > h_max)
			h_max = d->comp[i].h;
		if (d->comp[i].v > v_max)
			v_max = d->comp[i].v;
	}
	d->h_max = h_max;
	d->v_max = v_max;
	d->mcu_w = h_max * 8;
	d->mcu_h = v_max * 8;
	d->mcu_x = (d->img_x + d->mcu_w - 1) / d->mcu_w;
	d->mcu_y = (d->img_y + d->mcu_h - 1) / d->mcu_h;

	for (i = 0; i < d->img_n; ++i) {
		mjh_comp *cp = &d->comp[i];
		cp->x = (d->img_x * cp->h + h_max - 1) / h_max;
		cp->y = (d->img_y * cp->v + v_max - 1) / v_max;
		cp->w2 = d->mcu_x * cp->h * 8;
		cp->h2 = d->mcu_y * cp->v * 8;
		cp->bw = cp->w2 >> 3;
		cp->bh = cp->h2 >> 3;
		cp->plane = NULL;
		cp->touched = 0;
		/* the reference mallocs w2*h2+15 bytes here (and 2x that for progressive) */
		if (!mul_fits_int(cp->w2, cp->h2) || cp->w2 * cp->h2 > INT_MAX - 15)
			return fail(d, "outofmem");
		if (d->progressive && (!mul_fits_int(cp->w2 * cp->h2, 2) || cp->w2 * cp->h2 * 2 > INT_MAX - 15))
			return fail(d, "outofmem");
	}
	return 1;
}

/* codec/jpeg.c:1670-1699 */
int mjh_decode_header(mjh_decoder *d, mjh_reader *r, int mode)
{
	unsigned m;
	init_tile_off();
	d->r = r;
	d->reason = NULL;
	d->jfif = 0;
	d->app14 = -1;
	d->marker = MARKER_NONE;
	d->restart_interval = 0;
	d->max_block_l1 = 0;
	m = get_marker(d);
	if (m != 0xd8)
		return fail(d, "no SOI");
	if (mode == MJH_SCAN_TYPE)
		return 1;
	m = get_marker(d);
	while (!(m == 0xc0 || m == 0xc1 || m == 0xc2)) {
		if (!process_marker(d, m))
			return 0;
		m = get_marker(d);
		while (m == MARKER_NONE) {
			if (rd_eof(r))
				return fail(d, "no SOF");
			m = get_marker(d);
		}
	}
	d->progressive = (m == 0xc2);
	return process_frame_header(d, mode);
}

/* codec/jpeg.c:2241-2249 and the colour branches of :2320-2431 */
/* The colour branch load_jpeg_image takes (codec/jpeg.c:2244, :2320-2431) from what the markers seen SO FAR say.
 * The reference decides after stbi__decode_jpeg_image has processed every marker up to EOI, and APP0 / APP14 may
 * follow SOF or sit between scans (:1432-1461 run from any stbi__process_marker call): callers ask again once the
 * scans are done (or, for the GPU walk, once SOS is reached and EOI is known to follow the data). */
int mjh_color_mode(const mjh_decoder *d, int n_out)
{
	const int is_rgb = d->img_n == 3 && (d->rgb == 3 || (d->app14 == 0 && !d->jfif));
	if (d->img_n == 1)
		return MIJ_COLOR_GREY;
	if (d->img_n == 3)
		return is_rgb ? MIJ_COLOR_RGB : (n_out >= 3 ? MIJ_COLOR_YCBCR : MIJ_COLOR_GREY);
	return d->app14 == 0 ? MIJ_COLOR_CMYK : (d->app14 == 2 ? MIJ_COLOR_YCCK : MIJ_COLOR_YCBCRA);
}

int mjh_describe(const mjh_decoder *d, int req_comp, mij_image_desc *out)
{
	int n, i;
	if (req_comp < 0 || req_comp > 4)
		return 0;
	memset(out, 0, sizeof(*out));
	n = req_comp ? req_comp : (d->img_n >= 3 ? 3 : 1);
	out->width = d->img_x;
	out->height = d->img_y;
	out->ncomp = d->img_n;
	out->n_out = n;
	out->color = mjh_color_mode(d, n);
	out->flags = 0;
	out->h_max = d->h_max;
	out->v_max = d->v_max;
	out->mcu_x = d->mcu_x;
	out->mcu_y = d->mcu_y;
	for (i = 0; i < d->img_n; ++i) {
		out->comp[i].h = d->comp[i].h;
		out->comp[i].v = d->comp[i].v;
		out->comp[i].tq = d->comp[i].tq;
		out->comp[i].x = d->comp[i].x;
		out->comp[i].y = d->comp[i].y;
		out->comp[i].bw = d->comp[i].bw;
		out->comp[i].bh = d->comp[i].bh;
	}
	memcpy(out->dequant, d->dequant, sizeof(out->dequant));
	return 1;
}

/* L1 of every de-quantised block of a finished progressive image (the reference multiplies at
 * codec/jpeg.c:1319-1324; here only the bound for MIJ_FLAG_WIDE_IDCT is computed). */
static void progressive_l1(mjh_decoder *d)
{
	int ci;
	for (ci = 0; ci < d->img_n; ++ci) {
		const mjh_comp *cp = &d->comp[ci];
		const uint16_t *q = d->dequant[cp->tq];
		uint16_t qp[64];
		int nblk = cp->bw * cp->bh, ntile = (nblk + 63) >> 6, t, P, l;
		for (P = 0; P < 64; ++P) {
			int col = P >> 3, slot = P & 7, row = 0, rr;
			for (rr = 0; rr < 8; ++rr)
				if (mij_rowslot[rr] == slot)
					row = rr;
			qp[P] = q[8 * row + col];
		}
		for (t = 0; t < ntile; ++t) {
			const int16_t *tile = cp->plane + ((size_t)t << 12);
			/* a block's column chunk is 8 contiguous int16: (short)(coef * q) wraps like pmullw, and
			 * |-32768| = 32768 survives as an unsigned 16-bit lane */
			for (l = 0; l < 64; ++l) {
				__m128i sum = _mm_setzero_si128(), zero = _mm_setzero_si128();
				int c;
				uint32_t lanes[4];
				for (c = 0; c < 8; ++c) {
					__m128i v = _mm_loadu_si128((const __m128i *)(tile + (c << 9) + (l << 3)));
					__m128i q8 = _mm_loadu_si128((const __m128i *)(qp + 8 * c));
					__m128i m = _mm_mullo_epi16(v, q8);
					__m128i a = _mm_max_epi16(m, _mm_sub_epi16(zero, m));
					sum = _mm_add_epi32(sum, _mm_add_epi32(_mm_unpacklo_epi16(a, zero), _mm_unpackhi_epi16(a, zero)));
				}
				_mm_storeu_si128((__m128i *)lanes, sum);
				{
					int32_t l1 = (int32_t)(lanes[0] + lanes[1] + lanes[2] + lanes[3]);
					if (l1 > d->max_block_l1)
						d->max_block_l1 = l1;
				}
			}
		}
	}
}

/* codec/jpeg.c:1713-1755 */
int mjh_decode_scans(mjh_decoder *d)
{
	unsigned m = get_marker(d);
	while (m != 0xd9) {
		if (m == 0xda) {
			if (!process_scan_header(d))
				return 0;
			if (!parse_entropy_coded_data(d))
				return 0;
			if (d->marker == MARKER_NONE) {
				/* tolerate zero padding after the entropy data: look for the next 0xff */
				while (!rd_eof(d->r)) {
					unsigned x = rd8(d->r);
					if (x == 255) {
						d->marker = (unsigned char)rd8(d->r);
						break;
					}
				}
			}
		} else if (m == 0xdc) {
			int Ld = rd16be(d->r);
			int NL = rd16be(d->r);
			if (Ld != 4)
				return fail(d, "bad DNL len");
			if (NL != d->img_y)
				return fail(d, "bad DNL height");
		} else {
			if (!process_marker(d, m))
				return 0;
		}
		m = get_marker(d);
	}
	if (d->progressive && !d->defer_l1)
		progressive_l1(d);
	return 1;
}

int mjh_needs_wide_idct(const mjh_decoder *d) { return d->max_block_l1 > MIJ_BLOCK_L1_LIMIT; }

/* ------------------------------------------------------------------ one-call memory forms */

static int probe_common(mjh_decoder *d, mjh_reader *r, const uint8_t *buf, int len, int req_comp, mij_image_desc *desc, const char **reason)
{
	mjh_reader_mem(r, buf, len);
	if (!mjh_decode_header(d, r, MJH_SCAN_TYPE)) {
		*reason = "unknown image type";
		return 0;
	}
	mjh_reader_rewind(r);
	if (req_comp < 0 || req_comp > 4) {
		*reason = "bad req_comp";
		return 0;
	}
	if (!mjh_decode_header(d, r, MJH_SCAN_LOAD)) {
		*reason = d->reason;
		return 0;
	}
	if (!mjh_describe(d, req_comp, desc)) {
		*reason = "bad req_comp";
		return 0;
	}
	return 1;
}

int mjh_probe_memory(const uint8_t *buf, int len, int req_comp, mij_image_desc *desc, const char **reason)
{
	const char *why = NULL;
	mjh_reader r;
	mjh_decoder *d = (mjh_decoder *)calloc(1, sizeof(*d));
	int ok;
	if (!d) {
		if (reason)
			*reason = "outofmem";
		return 0;
	}
	ok = probe_common(d, &r, buf, len, req_comp, desc, &why);
	free(d);
	if (reason)
		*reason = why;
	return ok;
}

int mjh_attach_staging(mjh_decoder *d, const mij_image_desc *desc, uint8_t *region, int want_compact)
{
	int i;
	init_tile_off();
	d->any_escape = 0;
	d->defer_l1 = (want_compact && d->progressive) ? 1 : 0; /* the planes will be packed on the device: k_pack_c8 takes the L1 bound there */
	if (want_compact && !d->progressive) {
		for (i = 0; i < desc->ncomp; ++i) {
			size_t lo, dc, hi;
			mij_compact_offsets(desc, i, &lo, &dc, &hi);
			d->comp[i].plane = NULL;
			d->comp[i].lo8 = region + lo;
			d->comp[i].dc16 = (int16_t *)(region + dc);
			d->comp[i].hi8 = region + hi;
		}
		memset(region, 0, mij_compact_main_bytes(desc));
		d->compact = 1;
		return 1;
	}
	{
		size_t off = 0;
		for (i = 0; i < desc->ncomp; ++i) {
			d->comp[i].plane = (int16_t *)region + off;
			d->comp[i].lo8 = d->comp[i].hi8 = NULL;
			d->comp[i].dc16 = NULL;
			off += mij_plane_elems((uint32_t)(desc->comp[i].bw * desc->comp[i].bh));
		}
		memset(region, 0, off * sizeof(int16_t));
	}
	d->compact = 0;
	return 0;
}

uint32_t mjh_stage_flags(const mjh_decoder *d)
{
	return (mjh_needs_wide_idct(d) ? MIJ_FLAG_WIDE_IDCT : 0u) | (d->compact ? MIJ_FLAG_STAGED_COMPACT : 0u) | ((d->compact && d->any_escape) ? MIJ_FLAG_HAS_ESCAPES : 0u) |
			 ((d->progressive && d->defer_l1) ? MIJ_FLAG_L1_ON_DEVICE : 0u);
}

int mjh_decode_memory_fmt(const uint8_t *buf, int len, int req_comp, mij_image_desc *desc, uint8_t *region, size_t region_bytes, int want_compact, const char **reason)
{
	const char *why = NULL;
	mjh_reader r;
	mjh_decoder *d = (mjh_decoder *)calloc(1, sizeof(*d));
	int ok = 0;
	if (!d) {
		if (reason)
			*reason = "outofmem";
		return 0;
	}
	if (probe_common(d, &r, buf, len, req_comp, desc, &why)) {
		if (mij_image_region_bytes(desc) > region_bytes) {
			why = "outofmem";
			goto done;
		}
		mjh_attach_staging(d, desc, region, want_compact);
		if (!mjh_decode_scans(d)) {
			why = d->reason;
			goto done;
		}
		desc->flags |= mjh_stage_flags(d);
		desc->color = mjh_color_mode(d, desc->n_out); /* JFIF / Adobe markers behind SOF count too (codec/jpeg.c:2244) */
		ok = 1;
	}
done:
	free(d);
	if (reason)
		*reason = why;
	return ok;
}

int mjh_decode_memory(const uint8_t *buf, int len, int req_comp, mij_image_desc *desc, int16_t *arena, size_t arena_elems, const char **reason)
{
	const char *why = NULL;
	mjh_reader r;
	mjh_decoder *d = (mjh_decoder *)calloc(1, sizeof(*d));
	int ok = 0, i;
	size_t off = 0;
	if (!d) {
		if (reason)
			*reason = "outofmem";
		return 0;
	}
	if (probe_common(d, &r, buf, len, req_comp, desc, &why)) {
		for (i = 0; i < desc->ncomp; ++i) {
			size_t n = mij_plane_elems((uint32_t)(desc->comp[i].bw * desc->comp[i].bh));
			if (off + n > arena_elems) {
				why = "outofmem";
				goto done;
			}
			d->comp[i].plane = arena + off;
			off += n;
		}
		memset(arena, 0, off * sizeof(int16_t));
		if (!mjh_decode_scans(d)) {
			why = d->reason;
			goto done;
		}
		if (mjh_needs_wide_idct(d))
			desc->flags |= MIJ_FLAG_WIDE_IDCT;
		desc->color = mjh_color_mode(d, desc->n_out); /* JFIF / Adobe markers behind SOF count too (codec/jpeg.c:2244) */
		ok = 1;
	}
done:
	free(d);
	if (reason)
		*reason = why;
	return ok;
}

/* ------------------------------------------------------------------ scan extraction for the GPU entropy stage
 *
 * Header parsing as above; then, if the file is what the GPU walk takes -- one baseline scan carrying all
 * components interleaved in frame order, the entropy data (cut at RSTn markers if a restart interval is set)
 * followed by EOI -- the
 * segment is copied out with its 0xFF00 stuffing removed (codec/jpeg.c:171-184) together with the tables.
 * Everything else returns 2 ("use the host walk"): that path then reproduces the reference's behaviour,
 * including its failure reasons, so nothing about odd files is decided here.
 */
int mjh_extract_scan(const uint8_t *buf, int len, int req_comp, mjg_scan *scan, uint8_t *stream, size_t stream_cap, size_t *stream_len, const char **reason)
{
	const char *why = NULL;
	mjh_reader r;
	mjh_decoder *d = (mjh_decoder *)calloc(1, sizeof(*d));
	int rc = 2, ci, k, c;
	unsigned m;
	if (reason)
		*reason = NULL;
	if (!d) {
		if (reason)
			*reason = "outofmem";
		return 0;
	}
	if (!probe_common(d, &r, buf, len, req_comp, &scan->desc, &why)) {
		if (reason)
			*reason = why;
		rc = 0;
		goto done;
	}
	m = get_marker(d);
	while (m != 0xda) {
		if (m == 0xd9 || m == 0xdc || m == MARKER_NONE)
			goto done;
		if (!process_marker(d, m))
			goto done;
		m = get_marker(d);
	}
	if (!process_scan_header(d))
		goto done;
	scan->desc.color = mjh_color_mode(d, scan->desc.n_out); /* markers between SOF and SOS; EOI must follow the data (below) */
	if (d->progressive || d->scan_n != d->img_n || (d->img_n != 1 && d->img_n != 3))
		goto done;
	for (ci = 0; ci < d->scan_n; ++ci)
		if (d->order[ci] != ci)
			goto done;
	if (d->img_n == 1 && (d->comp[0].h != 1 || d->comp[0].v != 1))
		goto done; /* a lone component is walked over its own block grid (codec/jpeg.c:1160-1190), not over MCUs */
	{
		/* The entropy data: runs of bytes up to a 0xff that is not followed by 0x00.  Without a restart interval
		 * that marker must be EOI; with one, the data is cut at every RSTn (the reference accepts any RST number,
		 * :1183) into exactly ceil(MCUs / interval) pieces, the last one ending at EOI. */
		const uint8_t *q = r.p, *end = r.end;
		const uint32_t nmcu = (uint32_t)d->mcu_x * (uint32_t)d->mcu_y;
		const uint32_t dri = (uint32_t)d->restart_interval;
		const uint32_t want_seg = dri ? (nmcu + dri - 1) / dri : 1;
		uint32_t nseg = 0; /* the table itself is appended behind the data */
		size_t n = 0, seg_start = 0;
		uint32_t *table = NULL;
		int bpm = 0, at_eoi = 0;
		if (want_seg > 65536u)
			goto done;
		table = (uint32_t *)malloc(sizeof(uint32_t) * 2 * want_seg);
		if (!table)
			goto done;
		while (!at_eoi) {
			const uint8_t *f = (const uint8_t *)memchr(q, 0xff, (size_t)(end - q));
			size_t run;
			if (!f || f + 1 >= end) {
				free(table);
				goto done; /* no marker behind the data: truncated file, the host walk knows what the reference does */
			}
			run = (size_t)(f - q) + (f[1] == 0x00 ? 1u : 0u); /* a stuffed 0xff is data */
			if (n + run + 64 > stream_cap) {
				free(table);
				goto done;
			}
			memcpy(stream + n, q, run);
			n += run;
			if (f[1] == 0x00) {
				q = f + 2;
				continue;
			}
			if (f[1] == 0xd9)
				at_eoi = 1;
			else if (!(dri && f[1] >= 0xd0 && f[1] <= 0xd7)) {
				free(table);
				goto done; /* fill bytes, a stray restart marker, another scan, ...: host walk */
			}
			if (nseg >= want_seg) {
				free(table);
				goto done; /* more restart markers than intervals */
			}
			table[2 * nseg] = (uint32_t)seg_start;
			table[2 * nseg + 1] = (uint32_t)(n - seg_start);
			++nseg;
			memset(stream + n, 0, 32);
			n = (n + 32 + 3) & ~(size_t)3;
			seg_start = n;
			q = f + 2;
		}
		if (nseg != want_seg) {
			free(table);
			goto done; /* EOI before the last interval */
		}
		scan->n_seg = dri ? nseg : 0;
		scan->restart_mcus = dri;
		scan->reserved = 0;
		n = (n + 7) & ~(size_t)7;
		if (n + sizeof(uint32_t) * 2 * nseg + 32 > stream_cap) {
			free(table);
			goto done;
		}
		scan->seg_table_off = (uint32_t)n;
		memcpy(stream + n, table, sizeof(uint32_t) * 2 * nseg);
		n += sizeof(uint32_t) * 2 * nseg;
		free(table);
		*stream_len = n;
		for (ci = 0; ci < d->img_n; ++ci) {
			int x, y;
			for (y = 0; y < d->comp[ci].v; ++y)
				for (x = 0; x < d->comp[ci].h; ++x) {
					if (bpm >= 10)
						goto done;
					scan->blk_comp[bpm] = (uint8_t)ci;
					scan->blk_dx[bpm] = (uint8_t)x;
					scan->blk_dy[bpm] = (uint8_t)y;
					++bpm;
				}
			scan->dc_tab[ci] = (uint8_t)d->comp[ci].hd;
			scan->ac_tab[ci] = (uint8_t)(4 + d->comp[ci].ha);
			for (k = 0; k < 64; ++k)
				scan->qz[ci][k] = d->dequant[d->comp[ci].tq][k_dezigzag[k]];
		}
		scan->blocks_per_mcu = (uint32_t)bpm;
		scan->nblocks = (uint32_t)bpm * (uint32_t)d->mcu_x * (uint32_t)d->mcu_y;
		for (c = 0; c < 8; ++c) {
			const mjh_huff *h = c < 4 ? &d->huff_dc[c] : &d->huff_ac[c - 4];
			mjg_huff *g = &scan->huff[c];
			memcpy(g->fast, h->fast, sizeof(g->fast));
			memcpy(g->size, h->size, sizeof(g->size));
			memcpy(g->values, h->values, sizeof(g->values));
			memcpy(g->maxcode, h->maxcode, sizeof(g->maxcode));
			memset(g->delta, 0, sizeof(g->delta));
			memcpy(g->delta, h->delta, sizeof(h->delta));
		}
		rc = 1;
	}
done:
	free(d);
	return rc;
}
