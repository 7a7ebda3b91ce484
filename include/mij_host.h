/*
 * mij_host.h -- the host half of the JPEG decode path, as plain C entry points.
 *
 * This is the part of the reference that stays on the CPU -- marker parsing, Huffman tables and
 * the sequential entropy-coded-segment walk (codec/jpeg.c:88-558, :1119-1756) -- restated so
 * that every decoded block lands, still quantised, in the tile-layout staging planes of mij.h
 * instead of being de-quantised and inverse-transformed on the spot
 * (codec/jpeg.c:1178,:1217,:1342).  stbi_load* (image_api.h) is built from exactly these calls.
 */
#ifndef MIJ_HOST_H
#define MIJ_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "mij.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Type test + header + describe (stbi__jpeg_test, then codec/jpeg.c:1670-1699, :2241-2249):
 * fills *desc so the caller can size the coefficient planes.  Returns 1, or 0 with *reason = the
 * reference's short failure string ("unknown image type", "bad req_comp", "only 8-bit", ...).
 */
int mjh_probe_memory(const uint8_t *buf, int len, int req_comp, mij_image_desc *desc, const char **reason);

/*
 * The same plus the scans (codec/jpeg.c:1713-1755) into `arena`: the component planes in tile
 * layout, back to back in component order -- exactly the layout of the staging a mij batch hands
 * out (mij_batch_coef(b, slot, 0)).  The callee zero-fills what it uses.  desc->flags gets
 * MIJ_FLAG_WIDE_IDCT when the fast IDCT's range guarantee does not hold.  Returns 1 / 0 + reason.
 */
int mjh_decode_memory(const uint8_t *buf, int len, int req_comp, mij_image_desc *desc, int16_t *arena, size_t arena_elems, const char **reason);

/*
 * The same into an image's whole staging region (mij_batch_stage_region: mij_image_coef_bytes(desc) bytes), in the format the GPU
 * reads where the file allows it: want_compact != 0 and a baseline file -> COMPACT planes written by the walk itself (mij.h,
 * mij_compact_offsets: low bytes + DC array, escape bytes only for blocks that need them), desc->flags gets
 * MIJ_FLAG_STAGED_COMPACT (and MIJ_FLAG_HAS_ESCAPES): no pack pass on the device and 1.6 instead of 3 bytes per pixel over PCIe.
 * Progressive files (their scans read-modify-write int16 planes, codec/jpeg.c:372-558) and want_compact == 0: int16 tile layout
 * as mjh_decode_memory.  The reference's block decoder being restated either way: codec/jpeg.c:308-370.
 */
int mjh_decode_memory_fmt(const uint8_t *buf, int len, int req_comp, mij_image_desc *desc, uint8_t *region, size_t region_bytes, int want_compact, const char **reason);

/* Header parse + extraction of the entropy segment for the GPU entropy stage (mij.h: mjg_scan,
 * mij_batch_add_stream).  Returns 1: *scan filled, the segment without its 0xFF00 stuffing written to stream
 * (stream_cap must leave 32 spare bytes); 2: a valid header but not a layout the GPU walk takes (use
 * mjh_decode_memory); 0: rejected like mjh_probe_memory, *reason set. */
int mjh_extract_scan(const uint8_t *buf, int len, int req_comp, mjg_scan *scan, uint8_t *stream, size_t stream_cap, size_t *stream_len, const char **reason);

/*
 * Batch front end: n JPEGs into a mij batch.  Images are added to the batch in input order
 * (slots[i] = the slot of image i, or -1 with reasons[i] set when its header is rejected);
 * an image whose entropy data is rejected keeps its slot but is flagged MIJ_FLAG_SKIP (reasons[i]
 * set, slot reported as -1 - slot).  Follow with mij_batch_submit().  Returns the number of
 * images decoded successfully, or a negative MIJ_E_* code when the batch arenas are too small.
 *
 * mjh_decode_batch       the DEFAULT: the Huffman walk itself runs on the GPU wherever it applies (single-scan baseline
 *                        files; mjh_decode_batch_gpu below), the host walk is the fallback for every other layout and for
 *                        any stream the GPU walk reports back.  The batch gets its entropy arena on first use (sized for
 *                        that call; a later, larger call takes the host walk).  Environment MIJ_ENTROPY=host selects the
 *                        host-only front end instead (mjh_gpu_walk_default() says which is in force).  Pictures below
 *                        2200 pixels per host thread (MIJ_GPU_WALK_BATCH_MIN_PIXELS overrides) take the host walk in the same
 *                        call: their few subsequences leave the GPU walk's workgroups mostly idle.
 * mjh_decode_batch_host  the host stage of every image on `threads` host threads (one image per task), straight into the
 *                        batch's pinned staging -- what north_star describes ("the C host keeps the Huffman walk"); its
 *                        end-to-end rate is bounded by the walk (about 0.25-0.5 Gpix/s per host core) and by PCIe.
 */
int mjh_decode_batch(mij_batch *b, const uint8_t *const *bufs, const int *lens, int n, int req_comp, int threads, int *slots, const char **reasons);
int mjh_decode_batch_host(mij_batch *b, const uint8_t *const *bufs, const int *lens, int n, int req_comp, int threads, int *slots, const char **reasons);
int mjh_gpu_walk_default(void);

/*
 * One logical batch over several devices (BASELINE config 3; north_star: "sharded across the 8 GPUs of one node on
 * separate HIP streams ... no RCCL"): batches[k] belongs to device k's context (any contexts will do -- two on one
 * device is what the single-GPU test uses), image i goes to owner[i] = the k-th of n_batches contiguous slices (sizes
 * differing by at most one), slots[i] / reasons[i] as above, relative to that batch.  The host threads are one
 * shared pool; the thread that walks the last image of a slice SUBMITS that batch (upload + kernels, asynchronous on
 * its own stream), so device k works while slice k+1 is still being walked.  Follow with mij_batch_wait() on every
 * batch.  Returns the number of images decoded, or a negative MIJ_E_* code.  No collective, no peer traffic: decoder
 * state is per image (codec/jpeg.c:2445).
 */
int mjh_decode_batch_multi(mij_batch *const *batches, int n_batches, const uint8_t *const *bufs, const int *lens, int n, int req_comp, int threads,
									int *owner, int *slots, const char **reasons);

/* The same contract with the Huffman walk on the GPU where it applies (mij_batch_entropy_reserve must have
 * been called, otherwise this is mjh_decode_batch): the host threads only parse headers and remove byte
 * stuffing, mij_batch_entropy_run walks the streams, and whatever it does not take or refuses is walked on
 * the host as above.  Waits for the GPU walk; follow with mij_batch_submit(). */
int mjh_decode_batch_gpu(mij_batch *b, const uint8_t *const *bufs, const int *lens, int n, int req_comp, int threads, int *slots, const char **reasons);

/* The same in two halves, so that the host can prepare the next batch while the GPU walks this one: begin parses,
 * unstuffs, adds the slots and queues the GPU walk (returns NULL with *rc set on failure; MIJ_E_STATE = no entropy
 * arena reserved); end waits for the walk, host-walks what it handed back and returns the count like above.
 * bufs, lens, slots and reasons must stay valid until end. */
typedef struct mjh_gpu_job mjh_gpu_job;
mjh_gpu_job *mjh_decode_batch_gpu_begin(mij_batch *b, const uint8_t *const *bufs, const int *lens, int n, int req_comp, int threads, int *slots,
													 const char **reasons, int *rc);
int mjh_decode_batch_gpu_end(mjh_gpu_job *job);

/*
 * JPEG writer in three steps (stbi_write_jpg_to_func = plan + transform + emit); step 2 also exists
 * on the GPU (mij_enc_* in mij.h) and must produce the same data units bit for bit.
 *   mjw_plan_init       quality mapping and tables                       codec/jpeg_write.c:220-243
 *   mjw_transform_host  colour transform, edge replication, 2x2 chroma mean, float AAN fDCT, quantiser
 *                       for every data unit                              codec/jpeg_write.c:24-74,96-118,283-352
 *   mjw_emit            headers, Huffman emission, padding, EOI          codec/jpeg_write.c:245-268,120-169,358-363
 * Data units: int16[64] each, zigzag order, MCU after MCU (4:2:0: Y00 Y01 Y10 Y11 U V; 4:4:4: Y U V).
 */
typedef void mjw_write_func(void *context, void *data, int size); /* == stbi_write_func */
typedef struct {
	int width, height, comp; /* comp 1..4 as passed to stbi_write_jpg */
	int subsample;           /* 1: 4:2:0 (quality <= 90), 0: 4:4:4 */
	int mcu_x, mcu_y, du_per_mcu;
	unsigned char ytab[64], ctab[64]; /* quantisation tables, zigzag order (as written to DQT) */
	float fdtbl_y[64], fdtbl_c[64];   /* 1 / (q * aan scale), natural order */
} mjw_plan;

int mjw_plan_init(mjw_plan *p, int width, int height, int comp, int quality); /* 0 on bad arguments */
size_t mjw_plan_du_count(const mjw_plan *p);
void mjw_transform_host(const mjw_plan *p, const void *pixels, int flip_vertically, int16_t *du);
int mjw_emit(const mjw_plan *p, const int16_t *du, mjw_write_func *func, void *context);
int mjw_flip_on_write(void); /* the flag set by stbi_flip_vertically_on_write */

/* the plan (tables, geometry) the GPU encoder built for a slot, for mjw_emit */
int mij_enc_plan(const mij_encoder *e, int slot, mjw_plan *out);

/*
 * stbi_write_jpg_to_func with step 2 on the GPU: same arguments, same byte stream (the data units
 * are bit-identical), 0 on bad arguments or when no gfx950 device is present (no CPU fallback: use
 * stbi_write_jpg_to_func for the host-only writer).
 */
int mij_write_jpg_to_func(mjw_write_func *func, void *context, int x, int y, int comp, const void *data, int quality);

/* A batch of pictures in host memory -> their JPEG byte streams (each what stbi_write_jpg_to_func delivers for that picture,
 * codec/jpeg_write.c:283-366): staging copies on `threads` host threads, ONE GPU launch for every picture's transform, one copy
 * back, Huffman emission on the host threads.  out[i] is a malloc'ed stream of out_len[i] bytes (the caller frees it) or NULL for a
 * picture whose arguments were bad (NULL pixels, sizes or comp mjw_plan_init refuses) -- such pictures, even a whole call of them,
 * are not an error.  Returns the number of streams written, or a negative MIJ_E_* (device or memory failure); on a negative
 * return every out[i] is NULL again and nothing is left for the caller to free. */
int mij_write_jpg_batch(const void *const *pixels, const int *x, const int *y, const int *comp, int n, int quality, int threads,
                        unsigned char **out, size_t *out_len);
/* mjw_emit into memory: bytes written, 0 when `cap` is too small or an argument is bad (cap >= 1024 + 2 bytes per coefficient always fits) */
size_t mjw_emit_to_memory(const mjw_plan *p, const int16_t *du, unsigned char *out, size_t cap);

#ifdef __cplusplus
}
#endif

#endif /* MIJ_HOST_H */
