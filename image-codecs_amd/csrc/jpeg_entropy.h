/*
 * jpeg_entropy.h -- host half of the JPEG decode path (product code, plain C).
 *
 * Marker parsing and the sequential Huffman bitstream walk stay on the CPU, as in the
 * reference (codec/jpeg.c:88-558 tables/bit reader/block decoders, :1119-1756 markers, scans,
 * frames).  What changes is where a decoded block goes: instead of being de-quantised and
 * handed to idct_block_kernel (codec/jpeg.c:1178,:1217,:1342) its *quantised* coefficients are
 * written into the tile-layout staging planes of mij.h, from where the GPU takes over.
 *
 * Byte-source semantics (EOF reads as 0, callback buffering, rewind) follow the reference's
 * stbi__context (common.c:10-126) so that truncated and padded files behave identically.
 */
#ifndef MIJ_JPEG_ENTROPY_H
#define MIJ_JPEG_ENTROPY_H

#include <stddef.h>
#include <stdint.h>
#include "image_api.h"
#include "mij.h"
#include "mij_host.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- byte source: restates stbi__context (common.c:10-126, Appendix A of SURVEY.md) ---- */
typedef struct {
	const uint8_t *p, *end;
	const uint8_t *orig, *orig_end;
	stbi_io_callbacks io; /* io.read == NULL: memory source */
	void *user;
	int from_callbacks; /* still refilling through io.read */
	int buflen;
	int already_read;
	uint8_t buf[128];
} mjh_reader;

void mjh_reader_mem(mjh_reader *r, const uint8_t *data, int len);
void mjh_reader_callbacks(mjh_reader *r, const stbi_io_callbacks *io, void *user);
void mjh_reader_file(mjh_reader *r, FILE *f);
void mjh_reader_rewind(mjh_reader *r);
/* bytes buffered but not consumed: what stbi_load_from_file seeks back (convert.c:208) */
long mjh_reader_unread(const mjh_reader *r);

/* ---- decoder state ---- */
#define MJH_FAST_BITS 9

typedef struct {
	uint8_t fast[1 << MJH_FAST_BITS]; /* symbol index for codes <= 9 bits, 255 = take the slow path */
	uint16_t code[256];
	uint8_t values[256];
	uint8_t size[257];
	uint32_t maxcode[18];
	int32_t delta[17];
} mjh_huff;

typedef struct {
	int id, h, v, tq, hd, ha;
	int dc_pred;
	int x, y, w2, h2;
	int bw, bh;       /* blocks per row / rows of the padded grid */
	int16_t *plane;   /* tile-layout staging plane (mij.h), owned by the caller */
	int touched;      /* a previous scan already wrote blocks of this component */
	/* compact staging (mjh_decoder.compact; baseline files only): low-byte tiles, int16 DC array, escape bytes (mij_compact_offsets) */
	uint8_t *lo8;
	int16_t *dc16;
	uint8_t *hi8;
} mjh_comp;

enum { MJH_SCAN_LOAD = 0, MJH_SCAN_TYPE = 1, MJH_SCAN_HEADER = 2 };

typedef struct {
	mjh_reader *r;
	mjh_huff huff_dc[4], huff_ac[4];
	uint16_t dequant[4][64]; /* natural order (codec/jpeg.c:1376) */
	int16_t fast_ac[4][1 << MJH_FAST_BITS];
	/* AC refinement scans (codec/jpeg.c:478-545): code + the sign bit that follows it in one lookup (build_fast_refine); progressive files only */
	uint16_t fast_refine[4][1 << MJH_FAST_BITS];

	int img_x, img_y, img_n;
	int h_max, v_max, mcu_x, mcu_y, mcu_w, mcu_h;
	mjh_comp comp[4];

	/* bit reader (codec/jpeg.c:64-67) */
	uint32_t code_buffer;
	int code_bits;
	unsigned char marker;
	int nomore;

	int progressive, spec_start, spec_end, succ_high, succ_low, eob_run;
	int jfif, app14, rgb;
	int scan_n, order[4];
	int restart_interval, todo;

	/* 1: baseline blocks go straight into COMPACT planes (comp[].lo8 / dc16 / hi8) instead of int16 tiles -- what the decode kernels
	 * read, so no pack pass and half the bytes over PCIe; progressive scans read-modify-write int16 planes and never set this.
	 * any_escape: some block holds a coefficient beyond a byte (its escape bytes are in use: MIJ_FLAG_HAS_ESCAPES) */
	int compact, any_escape;
	/* 1: a progressive file staged for a batch that packs on the device: the per-block L1 bound is taken there (MIJ_FLAG_L1_ON_DEVICE), not in a pass over the planes here */
	int defer_l1;
	/* largest per-block sum of |de-quantised coefficient| seen (baseline), for MIJ_FLAG_WIDE_IDCT */
	int32_t max_block_l1;
	const char *reason; /* short failure reason, reference wording */
} mjh_decoder;

/* SOI + markers up to and including SOF (codec/jpeg.c:1670-1699).  mode = MJH_SCAN_*.
 * Returns 1 / 0 (d->reason set). */
int mjh_decode_header(mjh_decoder *d, mjh_reader *r, int mode);

/* Fills an mij_image_desc from a parsed header and the caller's req_comp, following
 * load_jpeg_image (codec/jpeg.c:2241-2249): n_out, colour mode, which components the GPU needs.
 * Returns 1, or 0 for "bad req_comp". */
int mjh_describe(const mjh_decoder *d, int req_comp, mij_image_desc *out);

/* The colour mode (MIJ_COLOR_*) as the markers seen so far decide it: ask again after mjh_decode_scans, because
 * APP0 / APP14 segments may follow SOF (codec/jpeg.c:2244 is evaluated after the whole file has been parsed). */
int mjh_color_mode(const mjh_decoder *d, int n_out);

/* Everything after SOF until EOI (codec/jpeg.c:1713-1755): scans are entropy-decoded into
 * d->comp[i].plane (which the caller must have pointed at zero-filled tile-layout planes).
 * For progressive files every block's L1 is computed at the end.  Returns 1 / 0. */
int mjh_decode_scans(mjh_decoder *d);

/* Non-zero if the finished image needs MIJ_FLAG_WIDE_IDCT. */
int mjh_needs_wide_idct(const mjh_decoder *d);

/* Points a decoder whose header has been parsed at an image's staging region (mij_image_coef_bytes(desc) bytes) and clears what the
 * walk will use.  want_compact != 0 and a baseline file: compact planes (returns 1, d->compact set; clears mij_compact_main_bytes);
 * otherwise int16 tile-layout planes (returns 0; clears them all). */
int mjh_attach_staging(mjh_decoder *d, const mij_image_desc *desc, uint8_t *region, int want_compact);
/* MIJ_FLAG_* the finished walk raises: WIDE_IDCT, and STAGED_COMPACT / HAS_ESCAPES for compact staging */
uint32_t mjh_stage_flags(const mjh_decoder *d);

/* the one-call memory forms and the batch front end are declared in include/mij_host.h */

#ifdef __cplusplus
}
#endif

#endif
