/*
 * oracle/oracle_jpeg.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement ("port") of the reference's JPEG decode and encode paths, used only as the
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
 * (image-codecs_amd/) never includes, links or calls anything in this directory.
 *
 * Parity status: PINNED.  The restatement is checked bit-for-bit against (a) the golden vectors
 * in tests/golden/ that were produced by the real reference compiled in place (oracle/_ref, see
 * oracle/ref/ and tests/golden/make_golden.py), and (b) the reference library itself whenever
 * oracle/_ref/libstbref.so is present.  The reference ships no tests or fixtures of its own.
 */
#ifndef ORACLE_JPEG_H
#define ORACLE_JPEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* stbi_load_from_memory (convert.c:254 -> codec/jpeg.c:2224): returns malloc'd pixels or NULL;
 * *reason (optional) receives the reference's short failure string. */
unsigned char *orc_load_from_memory(const unsigned char *buf, int len, int *x, int *y, int *comp, int req_comp, const char **reason);
void orc_free(void *p);

/* stbi_info_from_memory (image_api.c:119) */
int orc_info_from_memory(const unsigned char *buf, int len, int *x, int *y, int *comp);

/* stages of the hot path, individually addressable */
void orc_idct_block(unsigned char *out, int out_stride, const short data[64]);              /* codec/jpeg.c:615 */
/* kind: 0 row_1, 1 v_2, 2 h_2, 3 hv_2, 4 generic; writes the resampled row to out, returns its length */
int orc_resample_row(int kind, unsigned char *out, const unsigned char *in_near, const unsigned char *in_far, int w, int hs); /* :1765-1840,:1962 */
void orc_ycbcr_to_rgb_row(unsigned char *out, const unsigned char *y, const unsigned char *pcb, const unsigned char *pcr, int count, int step); /* :1976 */

/* Decode, additionally dumping every block handed to the IDCT (de-quantised, natural order, in the
 * reference's call order) -- the twin of oracle/_ref's ref_decode_capture. */
unsigned char *orc_decode_capture(const unsigned char *buf, int len, int *x, int *y, int *comp, int req_comp, short *coef_out, long coef_cap, long *coef_n);

/* stbi_write_jpg_to_func into memory (codec/jpeg_write.c:368): bytes produced (may exceed cap), -1 on failure */
long orc_encode(unsigned char *dst, long cap, int w, int h, int comp, const void *pixels, int quality);

/* cpu_baseline helper: decode `n` JPEGs (bufs[i], lens[i]) `reps` times on `threads` pthreads
 * (one image per task), discarding the pixels; returns decoded pixels per call, wall seconds in *secs */
long orc_decode_many(const unsigned char *const *bufs, const int *lens, int n, int reps, int threads, int req_comp, double *secs);

#ifdef __cplusplus
}
#endif
#endif
