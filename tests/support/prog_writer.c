/*
 * Test-side progressive JPEG writer.
 *
 * The reference can only *read* progressive files (codec/jpeg.c:372-558, :1326-1347); it has no writer
 * for them, and the GPU box has no libjpeg.  This tool re-emits already-quantised coefficients as an
 * SOF2 stream with spectral-selection and successive-approximation scans (ITU-T T.81 Annex G), with
 * per-scan optimal Huffman tables (Annex K.2) so that EOB runs appear in the stream.  A progressive
 * file made from the same coefficients as a baseline file must decode to the same pixels -- that is
 * the self-check tests/ apply before using its output at full size.
 *
 * Not part of the product library: built by __graft_entry__.build() into tests/support/.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_CORR_BITS 1000

typedef struct {
	uint8_t *out;
	long cap, len;
	uint32_t acc;
	int nacc;
	int gather;
	long freq[257];
	uint16_t code[256];
	uint8_t size[256];
	unsigned eobrun;
	uint8_t corr[MAX_CORR_BITS + 64];
	int ncorr;
} pw_state;

typedef struct {
	int ncomp_in_scan, comp[4];
	int ss, se, ah, al;
} pw_scan;

static void put_byte(pw_state *s, unsigned b)
{
	if (s->len < s->cap)
		s->out[s->len] = (uint8_t)b;
	++s->len;
}

static void put_u16(pw_state *s, unsigned v)
{
	put_byte(s, v >> 8);
	put_byte(s, v & 255);
}

static void put_bits(pw_state *s, unsigned value, int n)
{
	if (n == 0)
		return;
	s->acc = (s->acc << n) | (value & ((1u << n) - 1));
	s->nacc += n;
	while (s->nacc >= 8) {
		unsigned b = (s->acc >> (s->nacc - 8)) & 255;
		put_byte(s, b);
		if (b == 0xff)
			put_byte(s, 0);
		s->nacc -= 8;
	}
}

static void flush_bits(pw_state *s)
{
	if (s->nacc > 0)
		put_bits(s, 0x7f, 8 - s->nacc);
	s->acc = 0;
	s->nacc = 0;
}

static void emit_symbol(pw_state *s, int sym)
{
	if (s->gather)
		++s->freq[sym];
	else
		put_bits(s, s->code[sym], s->size[sym]);
}

static void emit_bits(pw_state *s, unsigned value, int n)
{
	if (!s->gather)
		put_bits(s, value, n);
}

static void emit_buffered(pw_state *s, const uint8_t *bits, int n)
{
	int i;
	if (s->gather)
		return;
	for (i = 0; i < n; ++i)
		put_bits(s, bits[i], 1);
}

static int bitlen(unsigned v)
{
	int n = 0;
	while (v) {
		++n;
		v >>= 1;
	}
	return n;
}

static void emit_eobrun(pw_state *s)
{
	if (s->eobrun > 0) {
		int nbits = bitlen(s->eobrun) - 1;
		emit_symbol(s, nbits << 4);
		if (nbits)
			emit_bits(s, s->eobrun, nbits);
		s->eobrun = 0;
		emit_buffered(s, s->corr, s->ncorr);
		s->ncorr = 0;
	}
}

/* T.81 Annex K.2: code lengths from frequencies, limited to 16 bits, symbol 256 reserved so that no
 * real symbol gets the all-ones code. */
static void gen_table(long *freq, uint8_t bits[17], uint8_t vals[256], int *nvals)
{
	int codesize[257], others[257], cnt[33];
	int i, j, c1, c2, p;
	memset(codesize, 0, sizeof codesize);
	memset(cnt, 0, sizeof cnt);
	for (i = 0; i < 257; ++i)
		others[i] = -1;
	freq[256] = 1;
	for (;;) {
		long v = -1;
		c1 = -1;
		for (i = 0; i <= 256; ++i)
			if (freq[i] && (c1 < 0 || freq[i] <= v)) {
				v = freq[i];
				c1 = i;
			}
		c2 = -1;
		v = -1;
		for (i = 0; i <= 256; ++i)
			if (freq[i] && i != c1 && (c2 < 0 || freq[i] <= v)) {
				v = freq[i];
				c2 = i;
			}
		if (c2 < 0)
			break;
		freq[c1] += freq[c2];
		freq[c2] = 0;
		for (++codesize[c1]; others[c1] >= 0;) {
			c1 = others[c1];
			++codesize[c1];
		}
		others[c1] = c2;
		for (++codesize[c2]; others[c2] >= 0;) {
			c2 = others[c2];
			++codesize[c2];
		}
	}
	for (i = 0; i <= 256; ++i)
		if (codesize[i])
			++cnt[codesize[i] > 32 ? 32 : codesize[i]];
	for (i = 32; i > 16; --i)
		while (cnt[i] > 0) {
			j = i - 2;
			while (cnt[j] == 0)
				--j;
			cnt[i] -= 2;
			++cnt[i - 1];
			cnt[j + 1] += 2;
			--cnt[j];
		}
	for (i = 16; cnt[i] == 0; --i)
		;
	--cnt[i];
	bits[0] = 0;
	for (i = 1; i <= 16; ++i)
		bits[i] = (uint8_t)cnt[i];
	p = 0;
	for (i = 1; i <= 32; ++i)
		for (j = 0; j < 256; ++j)
			if (codesize[j] == i)
				vals[p++] = (uint8_t)j;
	*nvals = p;
}

static void derive_codes(pw_state *s, const uint8_t bits[17], const uint8_t *vals)
{
	unsigned code = 0;
	int l, i, k = 0;
	memset(s->size, 0, sizeof s->size);
	for (l = 1; l <= 16; ++l) {
		for (i = 0; i < bits[l]; ++i, ++k) {
			s->code[vals[k]] = (uint16_t)code++;
			s->size[vals[k]] = (uint8_t)l;
		}
		code <<= 1;
	}
}

/* ------------------------------------------------------------------ frame description */

typedef struct {
	const int16_t *plane[4]; /* [bh][bw][64], zigzag order */
	int ncomp, width, height, hs[4], vs[4], hmax, vmax, mcu_x, mcu_y;
	int bw[4], bh[4], cw[4], ch[4]; /* padded grid / blocks that non-interleaved scans visit */
	int dc_pred[4];
} pw_frame;

static const int16_t *block_at(const pw_frame *f, int c, int bx, int by) { return f->plane[c] + ((size_t)by * f->bw[c] + bx) * 64; }

static void dc_first(pw_state *s, pw_frame *f, int c, const int16_t *blk, int al)
{
	int v = blk[0] >> al; /* arithmetic: T.81 G.1.2.1 point transform for DC */
	int diff = v - f->dc_pred[c], a = diff < 0 ? -diff : diff, nbits = bitlen((unsigned)a);
	f->dc_pred[c] = v;
	emit_symbol(s, nbits);
	if (nbits)
		emit_bits(s, (unsigned)(diff < 0 ? diff - 1 : diff), nbits);
}

static void ac_first(pw_state *s, const int16_t *blk, int ss, int se, int al)
{
	int k, r = 0;
	for (k = ss; k <= se; ++k) {
		int v = blk[k], a, nbits;
		unsigned payload;
		if (v < 0) {
			a = (-v) >> al;
			payload = (unsigned)~a;
		} else {
			a = v >> al;
			payload = (unsigned)a;
		}
		if (a == 0) {
			++r;
			continue;
		}
		emit_eobrun(s);
		while (r > 15) {
			emit_symbol(s, 0xf0);
			r -= 16;
		}
		nbits = bitlen((unsigned)a);
		emit_symbol(s, (r << 4) | nbits);
		emit_bits(s, payload, nbits);
		r = 0;
	}
	if (r > 0) {
		++s->eobrun;
		if (s->eobrun == 0x7fff)
			emit_eobrun(s);
	}
}

static void ac_refine(pw_state *s, const int16_t *blk, int ss, int se, int al)
{
	int absv[64], k, r = 0, br = 0, eob = 0;
	uint8_t *brbuf = s->corr + s->ncorr;
	for (k = ss; k <= se; ++k) {
		int v = blk[k];
		absv[k] = (v < 0 ? -v : v) >> al;
		if (absv[k] == 1)
			eob = k;
	}
	for (k = ss; k <= se; ++k) {
		int a = absv[k];
		if (a == 0) {
			++r;
			continue;
		}
		while (r > 15 && k <= eob) {
			emit_eobrun(s);
			emit_symbol(s, 0xf0);
			r -= 16;
			emit_buffered(s, brbuf, br);
			brbuf = s->corr;
			br = 0;
		}
		if (a > 1) {
			brbuf[br++] = (uint8_t)(a & 1);
			continue;
		}
		emit_eobrun(s);
		emit_symbol(s, (r << 4) | 1);
		emit_bits(s, blk[k] < 0 ? 0u : 1u, 1);
		emit_buffered(s, brbuf, br);
		brbuf = s->corr;
		br = 0;
		r = 0;
	}
	if (r > 0 || br > 0) {
		/* emit_eobrun() above may have reset the buffer start: keep pending bits contiguous */
		if (brbuf != s->corr + s->ncorr)
			memmove(s->corr + s->ncorr, brbuf, (size_t)br);
		++s->eobrun;
		s->ncorr += br;
		if (s->eobrun == 0x7fff || s->ncorr > MAX_CORR_BITS - 64 + 1)
			emit_eobrun(s);
	}
}

static void encode_scan_pass(pw_state *s, pw_frame *f, const pw_scan *sc)
{
	int i, j, ci, x, y;
	s->eobrun = 0;
	s->ncorr = 0;
	memset(f->dc_pred, 0, sizeof f->dc_pred);
	if (sc->ss == 0) {
		/* DC scans: interleaved over the MCU grid when they carry several components */
		if (sc->ncomp_in_scan == 1) {
			int c = sc->comp[0];
			for (j = 0; j < f->ch[c]; ++j)
				for (i = 0; i < f->cw[c]; ++i) {
					const int16_t *blk = block_at(f, c, i, j);
					if (sc->ah == 0)
						dc_first(s, f, c, blk, sc->al);
					else
						emit_bits(s, (unsigned)(blk[0] >> sc->al) & 1u, 1);
				}
		} else {
			for (j = 0; j < f->mcu_y; ++j)
				for (i = 0; i < f->mcu_x; ++i)
					for (ci = 0; ci < sc->ncomp_in_scan; ++ci) {
						int c = sc->comp[ci];
						for (y = 0; y < f->vs[c]; ++y)
							for (x = 0; x < f->hs[c]; ++x) {
								const int16_t *blk = block_at(f, c, i * f->hs[c] + x, j * f->vs[c] + y);
								if (sc->ah == 0)
									dc_first(s, f, c, blk, sc->al);
								else
									emit_bits(s, (unsigned)(blk[0] >> sc->al) & 1u, 1);
							}
					}
		}
	} else {
		int c = sc->comp[0];
		for (j = 0; j < f->ch[c]; ++j)
			for (i = 0; i < f->cw[c]; ++i) {
				const int16_t *blk = block_at(f, c, i, j);
				if (sc->ah == 0)
					ac_first(s, blk, sc->ss, sc->se, sc->al);
				else
					ac_refine(s, blk, sc->ss, sc->se, sc->al);
			}
		emit_eobrun(s);
	}
}

static void write_scan(pw_state *s, pw_frame *f, const pw_scan *sc)
{
	int i, needs_table = !(sc->ss == 0 && sc->ah != 0);
	if (needs_table) {
		uint8_t bits[17], vals[256];
		int nvals;
		memset(s->freq, 0, sizeof s->freq);
		s->gather = 1;
		encode_scan_pass(s, f, sc);
		s->gather = 0;
		gen_table(s->freq, bits, vals, &nvals);
		derive_codes(s, bits, vals);
		put_u16(s, 0xffc4);
		put_u16(s, (unsigned)(2 + 1 + 16 + nvals));
		put_byte(s, sc->ss == 0 ? 0x00 : 0x10); /* DC table 0 / AC table 0, redefined per scan */
		for (i = 1; i <= 16; ++i)
			put_byte(s, bits[i]);
		for (i = 0; i < nvals; ++i)
			put_byte(s, vals[i]);
	}
	put_u16(s, 0xffda);
	put_u16(s, (unsigned)(6 + 2 * sc->ncomp_in_scan));
	put_byte(s, (unsigned)sc->ncomp_in_scan);
	for (i = 0; i < sc->ncomp_in_scan; ++i) {
		put_byte(s, (unsigned)sc->comp[i] + 1);
		put_byte(s, 0x00);
	}
	put_byte(s, (unsigned)sc->ss);
	put_byte(s, (unsigned)sc->se);
	put_byte(s, (unsigned)((sc->ah << 4) | sc->al));
	encode_scan_pass(s, f, sc);
	flush_bits(s);
}

static int add_scan(pw_scan *list, int n, int ncomp, int c, int ss, int se, int ah, int al)
{
	int i;
	list[n].ncomp_in_scan = c < 0 ? ncomp : 1;
	for (i = 0; i < 4; ++i)
		list[n].comp[i] = c < 0 ? i : c;
	list[n].ss = ss;
	list[n].se = se;
	list[n].ah = ah;
	list[n].al = al;
	return n + 1;
}

/*
 * planes[c]: quantised coefficients of component c, [bh][bw][64] in zigzag order on the padded MCU grid
 * (bw = mcu_x * hs[c]).  qtab: two tables in zigzag order (component 0 uses table 0, the others table 1).
 * script 0: DC then full AC band per component, no successive approximation;
 * script 1: the ten-scan spectral + successive-approximation progression (Al up to 2, refinement scans).
 * Returns the stream length (which may exceed cap: nothing past cap is written), or -1 on bad arguments.
 */
long pw_write_progressive(const int16_t *const *planes, int ncomp, int width, int height, const int *hs, const int *vs,
								  const uint8_t *qtab /* [2][64] */, int script, uint8_t *out, long cap)
{
	pw_state *s;
	pw_frame f;
	pw_scan scans[16];
	int n = 0, c, i;
	long len;
	if ((ncomp != 1 && ncomp != 3) || width < 1 || height < 1 || width > 65535 || height > 65535)
		return -1;
	memset(&f, 0, sizeof f);
	f.ncomp = ncomp;
	f.width = width;
	f.height = height;
	for (c = 0; c < ncomp; ++c) {
		if (hs[c] < 1 || hs[c] > 4 || vs[c] < 1 || vs[c] > 4)
			return -1;
		f.hs[c] = hs[c];
		f.vs[c] = vs[c];
		if (hs[c] > f.hmax)
			f.hmax = hs[c];
		if (vs[c] > f.vmax)
			f.vmax = vs[c];
		f.plane[c] = planes[c];
	}
	f.mcu_x = (width + 8 * f.hmax - 1) / (8 * f.hmax);
	f.mcu_y = (height + 8 * f.vmax - 1) / (8 * f.vmax);
	for (c = 0; c < ncomp; ++c) {
		int x = (width * f.hs[c] + f.hmax - 1) / f.hmax, y = (height * f.vs[c] + f.vmax - 1) / f.vmax;
		f.bw[c] = f.mcu_x * f.hs[c];
		f.bh[c] = f.mcu_y * f.vs[c];
		f.cw[c] = (x + 7) >> 3;
		f.ch[c] = (y + 7) >> 3;
	}
	if (script == 0) {
		n = add_scan(scans, n, ncomp, ncomp == 1 ? 0 : -1, 0, 0, 0, 0);
		for (c = 0; c < ncomp; ++c)
			n = add_scan(scans, n, ncomp, c, 1, 63, 0, 0);
	} else if (ncomp == 1) {
		n = add_scan(scans, n, ncomp, 0, 0, 0, 0, 1);
		n = add_scan(scans, n, ncomp, 0, 1, 5, 0, 2);
		n = add_scan(scans, n, ncomp, 0, 6, 63, 0, 2);
		n = add_scan(scans, n, ncomp, 0, 1, 63, 2, 1);
		n = add_scan(scans, n, ncomp, 0, 0, 0, 1, 0);
		n = add_scan(scans, n, ncomp, 0, 1, 63, 1, 0);
	} else {
		n = add_scan(scans, n, ncomp, -1, 0, 0, 0, 1);
		n = add_scan(scans, n, ncomp, 0, 1, 5, 0, 2);
		n = add_scan(scans, n, ncomp, 2, 1, 63, 0, 1);
		n = add_scan(scans, n, ncomp, 1, 1, 63, 0, 1);
		n = add_scan(scans, n, ncomp, 0, 6, 63, 0, 2);
		n = add_scan(scans, n, ncomp, 0, 1, 63, 2, 1);
		n = add_scan(scans, n, ncomp, -1, 0, 0, 1, 0);
		n = add_scan(scans, n, ncomp, 2, 1, 63, 1, 0);
		n = add_scan(scans, n, ncomp, 1, 1, 63, 1, 0);
		n = add_scan(scans, n, ncomp, 0, 1, 63, 1, 0);
	}

	s = (pw_state *)calloc(1, sizeof *s);
	if (!s)
		return -1;
	s->out = out;
	s->cap = cap;
	put_u16(s, 0xffd8);
	for (i = 0; i < (ncomp == 1 ? 1 : 2); ++i) {
		int k;
		put_u16(s, 0xffdb);
		put_u16(s, 67);
		put_byte(s, (unsigned)i);
		for (k = 0; k < 64; ++k)
			put_byte(s, qtab[i * 64 + k]);
	}
	put_u16(s, 0xffc2);
	put_u16(s, (unsigned)(8 + 3 * ncomp));
	put_byte(s, 8);
	put_u16(s, (unsigned)height);
	put_u16(s, (unsigned)width);
	put_byte(s, (unsigned)ncomp);
	for (c = 0; c < ncomp; ++c) {
		put_byte(s, (unsigned)c + 1);
		put_byte(s, (unsigned)((f.hs[c] << 4) | f.vs[c]));
		put_byte(s, c == 0 ? 0 : 1);
	}
	for (i = 0; i < n; ++i)
		write_scan(s, &f, &scans[i]);
	put_u16(s, 0xffd9);
	len = s->len;
	free(s);
	return len;
}

/* ------------------------------------------------------------------ baseline (SOF0) twin with restart intervals
 *
 * One interleaved scan, optimal tables per class (table 0: component 0, table 1: the others), and, when
 * restart_mcus > 0, a DRI segment plus an RSTn marker after every restart_mcus MCUs (T.81 E.1.4, F.1.1.5.3):
 * the layout cameras write and the reference's own writer never does.
 */
typedef struct {
	long freq[4][257]; /* DC0, DC1, AC0, AC1 */
	uint16_t code[4][256];
	uint8_t size[4][256];
} bl_tabs;

static void bl_symbol(pw_state *s, bl_tabs *t, int which, int sym)
{
	if (s->gather)
		++t->freq[which][sym];
	else
		put_bits(s, t->code[which][sym], t->size[which][sym]);
}

static void bl_block(pw_state *s, bl_tabs *t, int cls, const int16_t *blk, int *pred)
{
	int diff = blk[0] - *pred, a = diff < 0 ? -diff : diff, nbits = bitlen((unsigned)a), k, r = 0;
	*pred = blk[0];
	bl_symbol(s, t, cls, nbits);
	if (nbits)
		emit_bits(s, (unsigned)(diff < 0 ? diff - 1 : diff), nbits);
	for (k = 1; k < 64; ++k) {
		int v = blk[k], m;
		if (v == 0) {
			++r;
			continue;
		}
		while (r > 15) {
			bl_symbol(s, t, 2 + cls, 0xf0);
			r -= 16;
		}
		m = v < 0 ? -v : v;
		nbits = bitlen((unsigned)m);
		bl_symbol(s, t, 2 + cls, (r << 4) | nbits);
		emit_bits(s, (unsigned)(v < 0 ? v - 1 : v), nbits);
		r = 0;
	}
	if (r > 0)
		bl_symbol(s, t, 2 + cls, 0x00);
}

static void bl_scan(pw_state *s, bl_tabs *t, pw_frame *f, int restart_mcus)
{
	int i, j, c, x, y, count = 0, rst = 0;
	memset(f->dc_pred, 0, sizeof f->dc_pred);
	for (j = 0; j < f->mcu_y; ++j)
		for (i = 0; i < f->mcu_x; ++i) {
			if (restart_mcus && count == restart_mcus) {
				if (!s->gather) {
					flush_bits(s);
					put_u16(s, 0xffd0u + (unsigned)(rst & 7));
				}
				++rst;
				count = 0;
				memset(f->dc_pred, 0, sizeof f->dc_pred);
			}
			for (c = 0; c < f->ncomp; ++c)
				for (y = 0; y < f->vs[c]; ++y)
					for (x = 0; x < f->hs[c]; ++x)
						bl_block(s, t, c == 0 ? 0 : 1, block_at(f, c, i * f->hs[c] + x, j * f->vs[c] + y), &f->dc_pred[c]);
			++count;
		}
}

/* app14_transform >= 0: an Adobe APP14 segment with that colour-transform byte in front of the tables (3 components,
 * transform 0: RGB-tagged; 4 components: 0 CMYK, 2 YCCK -- codec/jpeg.c:1528-1549, :2234-2244); -1: none */
long pw_write_baseline_ex(const int16_t *const *planes, int ncomp, int width, int height, const int *hs, const int *vs, const uint8_t *qtab /* [2][64] */,
								  int restart_mcus, int app14_transform, uint8_t *out, long cap);

long pw_write_baseline(const int16_t *const *planes, int ncomp, int width, int height, const int *hs, const int *vs, const uint8_t *qtab /* [2][64] */,
							  int restart_mcus, uint8_t *out, long cap)
{
	if (ncomp != 1 && ncomp != 3)
		return -1;
	return pw_write_baseline_ex(planes, ncomp, width, height, hs, vs, qtab, restart_mcus, -1, out, cap);
}

long pw_write_baseline_ex(const int16_t *const *planes, int ncomp, int width, int height, const int *hs, const int *vs, const uint8_t *qtab /* [2][64] */,
								  int restart_mcus, int app14_transform, uint8_t *out, long cap)
{
	pw_state *s;
	bl_tabs *t;
	pw_frame f;
	int c, i, k;
	long len;
	if ((ncomp != 1 && ncomp != 3 && ncomp != 4) || width < 1 || height < 1 || width > 65535 || height > 65535 || restart_mcus < 0 || restart_mcus > 65535)
		return -1;
	memset(&f, 0, sizeof f);
	f.ncomp = ncomp;
	f.width = width;
	f.height = height;
	for (c = 0; c < ncomp; ++c) {
		if (hs[c] < 1 || hs[c] > 4 || vs[c] < 1 || vs[c] > 4)
			return -1;
		f.hs[c] = hs[c];
		f.vs[c] = vs[c];
		if (hs[c] > f.hmax)
			f.hmax = hs[c];
		if (vs[c] > f.vmax)
			f.vmax = vs[c];
		f.plane[c] = planes[c];
	}
	f.mcu_x = (width + 8 * f.hmax - 1) / (8 * f.hmax);
	f.mcu_y = (height + 8 * f.vmax - 1) / (8 * f.vmax);
	for (c = 0; c < ncomp; ++c) {
		f.bw[c] = f.mcu_x * f.hs[c];
		f.bh[c] = f.mcu_y * f.vs[c];
	}
	s = (pw_state *)calloc(1, sizeof *s);
	t = (bl_tabs *)calloc(1, sizeof *t);
	if (!s || !t) {
		free(s);
		free(t);
		return -1;
	}
	s->out = out;
	s->cap = cap;
	s->gather = 1;
	bl_scan(s, t, &f, restart_mcus);
	s->gather = 0;
	put_u16(s, 0xffd8);
	if (app14_transform >= 0) {
		static const char tag[5] = {'A', 'd', 'o', 'b', 'e'};
		put_u16(s, 0xffee);
		put_u16(s, 14);
		for (k = 0; k < 5; ++k)
			put_byte(s, (unsigned char)tag[k]);
		put_u16(s, 100); /* version */
		put_u16(s, 0);   /* flags0 */
		put_u16(s, 0);   /* flags1 */
		put_byte(s, (unsigned)app14_transform);
	}
	for (i = 0; i < (ncomp == 1 ? 1 : 2); ++i) {
		put_u16(s, 0xffdb);
		put_u16(s, 67);
		put_byte(s, (unsigned)i);
		for (k = 0; k < 64; ++k)
			put_byte(s, qtab[i * 64 + k]);
	}
	put_u16(s, 0xffc0);
	put_u16(s, (unsigned)(8 + 3 * ncomp));
	put_byte(s, 8);
	put_u16(s, (unsigned)height);
	put_u16(s, (unsigned)width);
	put_byte(s, (unsigned)ncomp);
	for (c = 0; c < ncomp; ++c) {
		put_byte(s, (unsigned)c + 1);
		put_byte(s, (unsigned)((f.hs[c] << 4) | f.vs[c]));
		put_byte(s, c == 0 ? 0 : 1);
	}
	for (i = 0; i < 4; ++i) { /* DC0 DC1 AC0 AC1 */
		uint8_t bits[17], vals[256];
		unsigned code = 0;
		int nvals, l, n, q = 0;
		if (ncomp == 1 && (i & 1))
			continue;
		gen_table(t->freq[i], bits, vals, &nvals);
		for (l = 1; l <= 16; ++l) {
			for (n = 0; n < bits[l]; ++n, ++q) {
				t->code[i][vals[q]] = (uint16_t)code++;
				t->size[i][vals[q]] = (uint8_t)l;
			}
			code <<= 1;
		}
		put_u16(s, 0xffc4);
		put_u16(s, (unsigned)(2 + 1 + 16 + nvals));
		put_byte(s, (unsigned)(((i >> 1) << 4) | (i & 1)));
		for (l = 1; l <= 16; ++l)
			put_byte(s, bits[l]);
		for (n = 0; n < nvals; ++n)
			put_byte(s, vals[n]);
	}
	if (restart_mcus) {
		put_u16(s, 0xffdd);
		put_u16(s, 4);
		put_u16(s, (unsigned)restart_mcus);
	}
	put_u16(s, 0xffda);
	put_u16(s, (unsigned)(6 + 2 * ncomp));
	put_byte(s, (unsigned)ncomp);
	for (c = 0; c < ncomp; ++c) {
		put_byte(s, (unsigned)c + 1);
		put_byte(s, c == 0 ? 0x00 : 0x11);
	}
	put_byte(s, 0);
	put_byte(s, 63);
	put_byte(s, 0);
	bl_scan(s, t, &f, restart_mcus);
	flush_bits(s);
	put_u16(s, 0xffd9);
	len = s->len;
	free(s);
	free(t);
	return len;
}
