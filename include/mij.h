/*
 * mij.h -- C-ABI of the MI355X (gfx950) JPEG back end: "MI JPEG".
 *
 * This is the drop-in boundary for the reference's per-block / per-row kernel seam
 *     stbi__jpeg::idct_block_kernel        (codec/jpeg.c:83, called :1178 :1217 :1342)
 *     stbi__jpeg::resample_row_hv_2_kernel (codec/jpeg.c:85, called :2287/:2308)
 *     stbi__jpeg::YCbCr_to_RGB_kernel      (codec/jpeg.c:84, called :2338 :2357 :2369)
 * coarsened from "one block / one row per call" to "one batch of images per submit":
 * the host entropy decoder (the part of codec/jpeg.c:1155-1317 that stays on the CPU) writes
 * the *quantised* coefficients of every block straight into pinned staging memory owned by a
 * batch; one submit then runs, on the GPU, for every image of the batch
 *     de-quantisation   codec/jpeg.c:325,345,365 (baseline) / :1319-1324 (progressive)
 *     8x8 integer IDCT  codec/jpeg.c:615-679
 *     chroma upsample   codec/jpeg.c:1765-1840,1962-1971 chosen as :2280-2289, rows as :2301-2319
 *     colour + output   codec/jpeg.c:1976-2018 and the branches of :2320-2431
 * and leaves n_out*width*height interleaved bytes per image in device memory (fetch = D2H).
 *
 * Plain C: no HIP, C++ or torch types cross this boundary.  All functions return 0 on
 * success or a negative MIJ_E_* code; mij_last_error() gives a thread-local message.
 * A batch is owned by one host thread at a time; different batches may be driven from
 * different threads concurrently (each batch has its own HIP stream).
 */
#ifndef MIJ_H
#define MIJ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIJ_ABI_VERSION 1

enum {
	MIJ_OK = 0,
	MIJ_E_NODEVICE = -1, /* no usable HIP device / HIP runtime error at init */
	MIJ_E_ARG = -2,      /* bad argument or descriptor */
	MIJ_E_NOMEM = -3,    /* batch arenas exhausted or allocation failed */
	MIJ_E_HIP = -4,      /* HIP runtime error (message has the HIP string) */
	MIJ_E_STATE = -5     /* call out of order (e.g. fetch before submit) */
};

/* how the decoded components map to output pixels: the branches of codec/jpeg.c:2320-2431 */
enum {
	MIJ_COLOR_GREY = 0,  /* 1 component (or luma-only decode of YCbCr when n_out < 3, :2246) */
	MIJ_COLOR_YCBCR = 1, /* 3 components, stbi__YCbCr_to_RGB_row (:2338) */
	MIJ_COLOR_RGB = 2,   /* 3 components tagged RGB: copy (:2325-2335), luma via stbi__compute_y for n_out<3 (:2382-2395) */
	MIJ_COLOR_CMYK = 3,  /* 4 components, Adobe transform 0 (:2343-2354, :2396-2408) */
	MIJ_COLOR_YCCK = 4,  /* 4 components, Adobe transform 2 (:2355-2366, :2409-2417) */
	MIJ_COLOR_YCBCRA = 5 /* 4 components, other transform: YCbCr, 4th ignored (:2367-2370); n_out<3: luma (:2418-2430) */
};

/* descriptor flags */
#define MIJ_FLAG_WIDE_IDCT 1u /* host could not prove that every first-pass IDCT output fits int16:
                                 run the exact 32-bit second pass (see MIJ_BLOCK_L1_LIMIT) */
#define MIJ_FLAG_SKIP 2u      /* the host stage rejected this image after its slot was taken: upload and
                                 launch ignore the slot (its output is undefined) */
#define MIJ_FLAG_STAGED_COMPACT 4u /* the host stage wrote the slot's staging as COMPACT planes itself (mij_compact_offsets; the
                                      baseline Huffman walk does: mjh_decode_memory_fmt): upload copies them as they are, no pack */
#define MIJ_FLAG_HAS_ESCAPES 8u    /* ... and at least one block holds a coefficient beyond a byte: the escape region goes up too */
#define MIJ_FLAG_L1_ON_DEVICE 16u  /* a progressive file whose per-block L1 bound (MIJ_BLOCK_L1_LIMIT) the host did NOT compute: the pack
                                      kernel, which reads every coefficient of the int16 staging anyway, takes the maximum and
                                      mij_batch_upload raises MIJ_FLAG_WIDE_IDCT from it (one small copy back and a wait inside upload) */

/* per component geometry, exactly the reference's img_comp[] fields (codec/jpeg.c:48-62, :1624-1655) */
typedef struct {
	int32_t h, v;   /* sampling factors 1..4 */
	int32_t tq;     /* quantisation table index 0..3 */
	int32_t x, y;   /* effective size in samples (:1627-1628) */
	int32_t bw, bh; /* padded size in 8x8 blocks: w2/8, h2/8 (:1636-1637, = coeff_w, coeff_h :1649-1650) */
} mij_comp_desc;

typedef struct {
	int32_t width, height; /* img_x, img_y */
	int32_t ncomp;         /* components decoded by the entropy stage: 1, 3 or 4 */
	int32_t n_out;         /* output bytes per pixel, 1..4 (:2241) */
	int32_t color;         /* MIJ_COLOR_* */
	uint32_t flags;        /* MIJ_FLAG_* */
	int32_t h_max, v_max;  /* img_h_max, img_v_max (:1616-1617) */
	int32_t mcu_x, mcu_y;  /* img_mcu_x, img_mcu_y (:1621-1622) */
	mij_comp_desc comp[4];
	uint16_t dequant[4][64]; /* natural (de-zigzagged) order, as stored at codec/jpeg.c:1376 */
} mij_image_desc;

/*
 * Coefficient staging layout ("tile layout").  Per component the blocks are numbered
 * L = bx + by*bw (the reference's coefficient indexing, codec/jpeg.c:1252,:1296,:1340); 64
 * consecutive blocks form one 8 KiB tile, and inside a tile the 16-byte chunk c (0..7) of
 * block lane l (= L & 63) sits at byte (c*64 + l)*16, so a 64-lane wavefront that owns one
 * block per lane reads each chunk as one fully coalesced 1 KiB access.  Chunk c holds
 * column c of the block as the four int16 pairs (r0,r4) (r2,r6) (r1,r3) (r5,r7) -- the
 * operand pairs of the first (column) IDCT pass -- so coefficient (row,col) has in-block
 * position P = 8*col + mij_rowslot[row].
 *
 * mij_coef_index(L, P) is the int16 index of that coefficient inside its component plane;
 * mij_zigzag_pos[k] is P for the k-th coefficient in zigzag (bitstream) order.
 */
static const uint8_t mij_rowslot[8] = {0, 4, 2, 5, 1, 6, 3, 7};

static const uint8_t mij_zigzag_pos[64] = {
	/* P of natural index dezigzag[k]; natural = 8*row+col -> 8*col + rowslot[row] */
	0, 8, 4, 2, 12, 16, 24, 20,
	10, 5, 1, 13, 18, 28, 32, 40,
	36, 26, 21, 9, 6, 3, 14, 17,
	29, 34, 44, 48, 56, 52, 42, 37,
	25, 22, 11, 7, 15, 19, 30, 33,
	45, 50, 60, 58, 53, 41, 38, 27,
	23, 31, 35, 46, 49, 61, 57, 54,
	43, 39, 47, 51, 62, 59, 55, 63};

/*
 * Compact coefficient planes: the DEFAULT format of the coefficients in HBM (the staging above stays int16: the
 * host walk and the progressive scans read-modify-write it, and mij_batch_upload packs it on the device).  Per
 * component and 64-block tile: 4 KiB of low bytes in the same chunk order (chunk c of block lane l at byte
 * (c*64 + l)*8, eight bytes = column c in mij_rowslot order), then behind all tiles an int16 DC array (one entry
 * per block; the byte at in-block position 0 holds the block's flags instead) and 64 "escape" bytes per block in
 * position order P.  coefficient == sext8(low byte) + 256 * (int8)escape byte  (mod 2^16 -- all that
 * (short)(coef * dequant), codec/jpeg.c:325-365, depends on); the escape bytes of a block are only defined (and
 * only read) when bit 0 of its flags byte is set, i.e. when one of its coefficients lies outside -128..127.  The
 * reference's coefficients have up to 15 magnitude bits (codec/jpeg.c:250-265), so this is exact for every stream;
 * a typical q=90 photograph reads 4.1 KiB per tile instead of 8.
 */
#define MIJ_TILE_COMPACT_BYTES (4096 + 128 + 4096)
enum { MIJ_COEF_INT16 = 0, MIJ_COEF_COMPACT = 1 };

static inline size_t mij_coef_index(uint32_t L, uint32_t P)
{
	return ((size_t)(L >> 6) << 12) + ((size_t)(P >> 3) << 9) + ((size_t)(L & 63u) << 3) + (P & 7u);
}
/* int16 elements in the plane of a component with nblocks blocks (whole tiles) */
static inline size_t mij_plane_elems(uint32_t nblocks) { return ((size_t)(nblocks + 63u) >> 6) << 12; }

/* Where the compact planes of component `comp` lie inside an image's coefficient region (the same in pinned staging and in HBM),
 * in bytes from the region's start: first, for every component in turn, its low-byte tiles followed by its int16 DC array -- the
 * MAIN part, mij_compact_main_bytes() in all, which is all that has to cross PCIe for an image without escaped blocks -- then the
 * escape bytes of every component (64 per block, position order P).  n_tiles = whole 64-block tiles of the component. */
static inline size_t mij_comp_tiles(const mij_comp_desc *c) { return ((size_t)(c->bw * c->bh) + 63u) >> 6; }
static inline size_t mij_compact_main_bytes(const mij_image_desc *d)
{
	size_t t = 0;
	int c;
	for (c = 0; c < d->ncomp; ++c)
		t += mij_comp_tiles(&d->comp[c]) * (4096 + 128);
	return t;
}
/* bytes of an image's coefficient region: room for either format (= mij_image_coef_bytes) */
static inline size_t mij_image_region_bytes(const mij_image_desc *d)
{
	size_t t = 0;
	int c;
	for (c = 0; c < d->ncomp; ++c)
		t += mij_comp_tiles(&d->comp[c]) * MIJ_TILE_COMPACT_BYTES;
	return t;
}
static inline void mij_compact_offsets(const mij_image_desc *d, int comp, size_t *lo, size_t *dc, size_t *hi)
{
	size_t main = 0, esc = mij_compact_main_bytes(d);
	int c;
	for (c = 0; c < comp; ++c) {
		main += mij_comp_tiles(&d->comp[c]) * (4096 + 128);
		esc += mij_comp_tiles(&d->comp[c]) * 4096;
	}
	*lo = main;
	*dc = main + mij_comp_tiles(&d->comp[comp]) * 4096;
	*hi = esc;
}

/*
 * Fast/exact IDCT contract.  The second IDCT pass runs on packed int16 first-pass outputs
 * (v_dot2_i32_i16) when the host guarantees they fit; a sufficient condition is that for every
 * block the sum of |de-quantised coefficient| is <= MIJ_BLOCK_L1_LIMIT (max |weight| of the
 * first pass is 5683, so |sum| + 512 < 2^25 and (..)>>10 fits int16).  Otherwise the host sets
 * MIJ_FLAG_WIDE_IDCT and the kernel computes the second pass in full 32-bit (wrapping)
 * arithmetic, like the reference's int math on such streams.
 */
#define MIJ_BLOCK_L1_LIMIT 5903

typedef struct mij_ctx mij_ctx;     /* one per (process, device) */
typedef struct mij_batch mij_batch; /* staging + device arenas + one HIP stream */

const char *mij_last_error(void);
int mij_abi_version(void);
int mij_device_count(void);

/* device < 0: use the HIP current device.  Fails with MIJ_E_NODEVICE when no GPU is present. */
int mij_ctx_create(int device, mij_ctx **out);
void mij_ctx_destroy(mij_ctx *ctx);
int mij_ctx_device(const mij_ctx *ctx);
/* "gfx950", CU count, bytes of device memory -- for logs and the bench header */
int mij_ctx_info(const mij_ctx *ctx, char *arch, size_t arch_len, int *cu_count, size_t *total_mem);

/*
 * A batch holds up to max_images images.  stage_bytes = pinned host staging for coefficients
 * (0: none, images can then only be added as clones or filled on the device), coef_bytes /
 * out_bytes = device arenas.  mij_image_coef_bytes / mij_image_out_bytes give an image's needs.
 */
int mij_batch_create(mij_ctx *ctx, int max_images, size_t stage_bytes, size_t coef_bytes, size_t out_bytes, mij_batch **out);
void mij_batch_destroy(mij_batch *b);
int mij_batch_reset(mij_batch *b); /* forget all images; arenas are reused */

size_t mij_image_coef_bytes(const mij_image_desc *d); /* sum over components of whole tiles, MIJ_TILE_COMPACT_BYTES each: room for either format */
size_t mij_image_out_bytes(const mij_image_desc *d);  /* n_out*width*height, rounded up to 256 */

/* Adds an image; returns its slot (>= 0) or a negative error.  Its staging planes are zeroed. */
int mij_batch_add(mij_batch *b, const mij_image_desc *d);
/* The same without the clearing, for a caller that writes every element of the planes itself (mjh_decode_memory
 * clears them on its own thread: the batch front ends add all images up front, in order, on one thread). */
int mij_batch_add_uncleared(mij_batch *b, const mij_image_desc *d);
/* Adds an image that shares descriptor and coefficients with slot src but gets its own device
 * coefficient and output buffers (filled device-to-device at upload).  For benchmarks that need
 * many resident images from a few distinct inputs. */
int mij_batch_add_clone(mij_batch *b, int src_slot);
/* Pinned host plane of component c of a slot (tile layout, int16, zero-filled). */
int16_t *mij_batch_coef(mij_batch *b, int slot, int comp);
/* The slot's whole staging region (mij_image_coef_bytes of pinned host memory) for a host stage that writes COMPACT planes itself
 * (mij_compact_offsets) and then raises MIJ_FLAG_STAGED_COMPACT with mij_batch_set_flags; NULL for clones and slots without staging. */
uint8_t *mij_batch_stage_region(mij_batch *b, int slot, size_t *bytes);
/* MIJ_COEF_COMPACT / MIJ_COEF_INT16: the format host-staged planes get in HBM (a host stage asked for int16 planes stages int16) */
int mij_batch_coef_format(const mij_batch *b);
/* MIJ_FLAG_* of a slot as they stand (after mij_batch_upload: with MIJ_FLAG_WIDE_IDCT where the device found it, MIJ_FLAG_L1_ON_DEVICE cleared) */
uint32_t mij_batch_slot_flags(const mij_batch *b, int slot);
/* May be called after the entropy stage to raise flags it only knows late (e.g. WIDE_IDCT). */
int mij_batch_set_flags(mij_batch *b, int slot, uint32_t flags);

/* May be called after the entropy stage when a marker behind SOF changed the colour branch (the reference decides
 * is_rgb / CMYK / YCCK after the last marker, codec/jpeg.c:2244); the new mode must fit the slot's component count. */
int mij_batch_set_color(mij_batch *b, int slot, int color);

/* submit = upload + launch.  All asynchronous on the batch's stream; wait blocks. */
int mij_batch_upload(mij_batch *b); /* H2D of staged coefficients (+ D2D for clones) + descriptors */
int mij_batch_launch(mij_batch *b); /* the decode kernels over every image of the batch */
int mij_batch_submit(mij_batch *b);
int mij_batch_wait(mij_batch *b);

/* D2H of one image's pixels into dst (dst_bytes >= n_out*width*height); waits for the batch. */
int mij_batch_fetch(mij_batch *b, int slot, uint8_t *dst, size_t dst_bytes);
/* The whole output arena in one asynchronous D2H on the batch's stream (image `slot` starts at byte
 * mij_batch_out_offset(b, slot) of dst; mij_batch_out_bytes(b) in total); mij_batch_wait() completes it.
 * Full PCIe rate needs a pinned destination: mij_host_alloc / mij_host_free. */
int mij_batch_fetch_all_async(mij_batch *b, uint8_t *dst, size_t dst_bytes);
size_t mij_batch_out_offset(const mij_batch *b, int slot);
size_t mij_batch_out_bytes(const mij_batch *b);
void *mij_host_alloc(size_t bytes);
void mij_host_free(void *p);
/* Device address of an image's pixels (valid until reset/destroy) for device-resident consumers. */
void *mij_batch_device_out(mij_batch *b, int slot);
int mij_batch_image_count(const mij_batch *b);

/*
 * Measurement hooks (used by bench.py): HIP events recorded on the batch's own stream.
 * begin/end bracket whatever was enqueued between them; elapsed waits for the end event.
 */
int mij_batch_timer_begin(mij_batch *b);
int mij_batch_timer_end(mij_batch *b);
int mij_batch_timer_elapsed_ms(mij_batch *b, float *ms);
/* FNV-1a 64 of an image's output computed on the device copy (D2H + hash on host); for parity checks of big batches */
int mij_batch_hash_out(mij_batch *b, int slot, uint64_t *hash);

/* Parity of big batches without bringing the pixels to the host: *ndiff = number of 16-byte words in which the
 * device images of the slot pairs (sa[i], sb[i]) differ (same sizes required); e.g. every clone against its source. */
int mij_batch_diff_slots(mij_batch *b, const int *sa, const int *sb, int n, uint64_t *ndiff);

/* which kernel family the last upload chose for a slot: 0 none (skipped), 1 fused 4:2:0, 2 generic two-pass,
 * 3 fused 4:4:4, 4 fused 4:2:2, 5 fused grey, 6 fused 4:4:0 */
int mij_batch_slot_path(const mij_batch *b, int slot);
/* parity tests compare the kernel families: on = 1 sends every image of the batch down the two-pass path (IDCT to sample
 * planes, then resampling + colour; its pass 2 compiled per resampler where the layout allows), on = 2 also insists on
 * the run-time-general pass 2 (k_resample_color), on = 0 is the default choice */
int mij_batch_force_generic(mij_batch *b, int on);

/* ---- GPU entropy stage (experimental): the baseline Huffman walk itself on the GPU, for single-scan interleaved
 * baseline files, restart intervals included.  The host only parses headers and removes
 * the 0xFF00 byte stuffing (mjh_extract_scan, mij_host.h); coefficients never cross PCIe.  Any stream the GPU
 * walk does not like (invalid code, run past coefficient 63, early end, no convergence) is reported back and
 * must be re-done with the host walk, whose behaviour on malformed input is the reference's. ---- */
typedef struct { /* stbi__huffman without code[] (codec/jpeg.c:21-32) */
	uint8_t fast[512];
	uint8_t size[256];
	uint8_t values[256];
	uint32_t maxcode[18];
	int32_t delta[18];
} mjg_huff;

typedef struct {
	mij_image_desc desc;
	uint32_t nblocks, blocks_per_mcu;
	uint8_t blk_comp[12], blk_dx[12], blk_dy[12]; /* block inside the MCU -> component, block offset inside the MCU */
	uint8_t dc_tab[4], ac_tab[4];                 /* component -> index into huff[] (0..3 DC tables, 4..7 AC tables) */
	mjg_huff huff[8];
	uint16_t qz[4][64];                           /* per component, zigzag order */
	/* Restart intervals (DRI): every interval is walked on its own (DC prediction starts at 0 in each, codec/jpeg.c
	 * :1142-1153).  n_seg intervals of restart_mcus MCUs (the last one shorter); their unstuffed bytes lie at
	 * seg[k].off .. + seg[k].len of the stream buffer, 4-byte aligned and 32 zero bytes apart, where seg = the
	 * table of n_seg {uint32 off, uint32 len} pairs at byte seg_table_off of the same buffer.  n_seg == 0: no
	 * restart interval, the whole stream is one segment at offset 0. */
	uint32_t n_seg, restart_mcus, seg_table_off, reserved;
} mjg_scan;

/* pinned + device arenas for stream_bytes of unstuffed entropy data; once per batch.  Device memory: about 9 bytes per stream byte (the write
 * pass's record arena takes 8.1 of them; streams beyond about 1 GiB per batch fall back to a form that needs 64 bytes per block instead). */
int mij_batch_entropy_reserve(mij_batch *b, size_t stream_bytes);
/* pinned region where the caller writes streams (capacity as reserved; reset by mij_batch_reset) */
uint8_t *mij_batch_entropy_stage(mij_batch *b, size_t *capacity);
/* new slot whose coefficients the GPU walk will produce; stream = what mjh_extract_scan wrote (segments, each
 * followed by 32 zero bytes, and the segment table), 4-byte aligned inside the pinned region.  Returns the slot
 * or a negative code (MIJ_E_NOMEM also when the image has more restart intervals than the arena has room for). */
int mij_batch_add_stream(mij_batch *b, const mjg_scan *scan, uint8_t *stream, size_t stream_len); /* stream_len: everything mjh_extract_scan wrote, segment table included */
/* H2D of the streams, the five kernels, D2H of the verdicts; waits.  fallback[0..*n_fallback) = slots the host
 * walk must redo (mij_batch_fallback_prepare, then decode into mij_batch_coef as usual). */
int mij_batch_entropy_run(mij_batch *b, int *fallback, int cap, int *n_fallback);
/* The same in two halves: launch queues everything on the batch's stream and returns at once (the host can
 * parse the next batch's headers meanwhile), finish waits and reports. */
int mij_batch_entropy_launch(mij_batch *b);
int mij_batch_entropy_finish(mij_batch *b, int *fallback, int cap, int *n_fallback);
int mij_batch_fallback_prepare(mij_batch *b, int slot);
/* tests: the coefficient planes of a slot as they sit in HBM, after entropy_run or upload, always returned in the
 * int16 tile layout (compact planes are expanded on the host); dst_elems >= sum of mij_plane_elems */
int mij_batch_fetch_coef(mij_batch *b, int slot, int16_t *dst, size_t dst_elems);
/* The format new coefficient planes of this batch get in HBM: MIJ_COEF_COMPACT (default; environment
 * MIJ_COEF_FORMAT=int16 flips the default) or MIJ_COEF_INT16.  Applies to slots added or uploaded afterwards. */
int mij_batch_set_coef_format(mij_batch *b, int fmt);
/* 1 if the slot's coefficients sit in HBM as compact planes (after upload or the GPU walk) */
int mij_batch_slot_coef_bytes(const mij_batch *b, int slot);
/* number of escaped blocks of a slot (blocks holding a coefficient outside -128..127), read back from HBM; tests */
int mij_batch_slot_escapes(mij_batch *b, int slot);

/* Measurement only: the decode kernels transform a wavefront's 64 blocks with the cheapest IDCT that covers all of them
 * (class 0: DC only -- the reference's own shortcut, codec/jpeg.c:625-633, taken per block instead of per column; 1: non-zeros
 * inside the top-left 2x2; 2: inside the 4x4; 3: the full transform).  mij_batch_count_idct_classes(b, 1) after mij_batch_upload
 * clears the device counters and makes the batch's launches count wavefronts per class; mij_batch_idct_class_counts reads them
 * (out[class]); (b, 0) switches the counting off again.  Counting costs an atomic per wavefront: never on in timed launches. */
/* Measurement: milliseconds k_pack_c8 (int16 staging -> compact planes) took in the last mij_batch_upload, -1 when nothing was packed */
int mij_batch_pack_ms(mij_batch *b, float *ms);
int mij_batch_count_idct_classes(mij_batch *b, int on);
int mij_batch_idct_class_counts(mij_batch *b, uint64_t out[4]);
/* tests / tuning: synchronisation rounds the last entropy_run needed for its slowest image */
int mij_batch_entropy_rounds(const mij_batch *b);

/*
 * Encoder half (BASELINE config 5): the JPEG writer's colour transform, edge replication, 2x2
 * chroma mean, float AAN forward DCT and quantiser (codec/jpeg_write.c:24-74, :96-118, :283-352)
 * for a batch of images on the GPU.  Input: interleaved 8-bit pixels, comp 1..4 as passed to
 * stbi_write_jpg; output: int16[64] data units in zigzag order, MCU after MCU (4:2:0: Y00 Y01 Y10
 * Y11 U V; 4:4:4: Y U V), bit-identical to mjw_transform_host (mij_host.h), for the host's
 * Huffman stage (mjw_emit).  Algorithmic bytes per 1080p image: 6 220 800 read + 6 266 880 written.
 */
typedef struct mij_encoder mij_encoder;

int mij_enc_create(mij_ctx *ctx, int max_images, size_t pixel_bytes, size_t du_bytes, mij_encoder **out);
void mij_enc_destroy(mij_encoder *e);
int mij_enc_reset(mij_encoder *e);
/* copies the pixels into pinned staging; quality and 4:2:0/4:4:4 choice as stbi_write_jpg; returns the slot */
int mij_enc_add(mij_encoder *e, const void *pixels, int width, int height, int comp, int quality, int flip_vertically);
int mij_enc_add_clone(mij_encoder *e, int src_slot); /* own device buffers, same pixels (benchmarks) */
/* the same slot bookkeeping without the copy: the caller stages the pixels with mij_enc_stage_pixels(slot) before
 * mij_enc_upload (batch front ends fill the slots from several host threads at once, mij_write_jpg_batch) */
int mij_enc_add_uncopied(mij_encoder *e, int width, int height, int comp, int quality, int flip_vertically);
void *mij_enc_staging(mij_encoder *e, int slot);
/* Bytes one picture takes in the pixel arenas (pix_cap of mij_enc_create).  Every picture is staged as packed RGB with rows of whole MCU
 * columns (width rounded up to 16, or to 8 above quality 90): the last pixel of a row repeated -- codec/jpeg_write.c:294-296 -- and the
 * channels picked as the reference picks them (:276-279: grey and grey + alpha pictures become r = g = b = grey, RGBA loses its alpha), both
 * applied on the way in, so that the strip kernels take every width and every comp; rounded up to 256. */
size_t mij_enc_pixel_bytes(int width, int height, int comp, int quality);
/* Copies a picture (comp bytes per pixel, as passed to stbi_write_jpg) into the staging of a slot made by mij_enc_add_uncopied in that layout
 * (callable from several threads for different slots).  mij_enc_staging() returns the same memory: packed RGB, row pitch = padded width x 3. */
int mij_enc_stage_pixels(mij_encoder *e, int slot, const void *pixels);
/* every slot's data units into the encoder's pinned mirror with one device-to-host copy (waits for it); mij_enc_units(slot)
 * points into that mirror until the next mij_enc_fetch_all / mij_enc_destroy */
int mij_enc_fetch_all(mij_encoder *e);
int mij_enc_fetch_all_async(mij_encoder *e); /* queued behind the launch, no wait: mij_enc_wait before reading mij_enc_units */
const int16_t *mij_enc_units(const mij_encoder *e, int slot);
int mij_enc_upload(mij_encoder *e);
int mij_enc_launch(mij_encoder *e);
int mij_enc_wait(mij_encoder *e);
int mij_enc_force_generic(mij_encoder *e, int on); /* tests: per-unit kernels even where the fused 4:2:0 kernel applies; before upload */
/* D2H of a slot's data units (mcu_x*mcu_y*du_per_mcu*64 int16) */
int mij_enc_fetch(mij_encoder *e, int slot, int16_t *dst, size_t dst_elems);
int mij_enc_timer_begin(mij_encoder *e);
int mij_enc_timer_end(mij_encoder *e);
int mij_enc_timer_elapsed_ms(mij_encoder *e, float *ms);

#ifdef __cplusplus
}
#endif

#endif /* MIJ_H */
