/*
 * oracle/ref/ref_tu.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * One C translation unit that compiles the reference's JPEG decode + encode path from
 * the sources where they lie under REF_ROOT (default /root/reference), in the include
 * order SURVEY.md Appendix A found to work:
 *     prelude -> common.c -> codec/jpeg.c -> stbi__info_main -> image_api.c -> convert.c
 *             -> codec/write_bmp.c -> codec/jpeg_write.c
 * It must be compiled as C (gnu99), not C++ (the reference relies on tentative
 * definitions).  Output goes to oracle/_ref/ only (git-ignored, never committed).
 *
 * Besides the reference's own public functions (stbi_load_from_memory,
 * stbi_info_from_memory, stbi_write_jpg_to_func, ...) the library exports thin ref_*
 * hooks around the *static* hot-path functions so tests can pin each stage:
 *   ref_idct_block            -> stbi__idct_block            codec/jpeg.c:615
 *   ref_resample_row_*        -> stbi__resample_row_*        codec/jpeg.c:1774-1840,1962
 *   ref_ycbcr_to_rgb_row      -> stbi__YCbCr_to_RGB_row      codec/jpeg.c:1976
 *   ref_decode_capture        -> load_jpeg_image             codec/jpeg.c:2224, with the
 *                                idct_block_kernel seam (codec/jpeg.c:83) wrapped so the
 *                                dequantised blocks and the IDCT planes can be dumped.
 */
#include "ref_prelude.h"

#ifndef REF_ROOT
#define REF_ROOT /root/reference
#endif
#define REF_STR2(x) #x
#define REF_STR(x) REF_STR2(x)
#define REF_FILE(rel) REF_STR(REF_ROOT/rel)

#include REF_FILE(common.c)
#include REF_FILE(codec/jpeg.c)

/* called by image_api.c:90,122,129 but never defined in the reference */
static int stbi__info_main(stbi__context *s, int *x, int *y, int *comp)
{
	if (stbi__jpeg_info(s, x, y, comp))
		return 1;
	return stbi__err("unknown image type", "Image not of any known type, or corrupt");
}

#include REF_FILE(image_api.c)
#include REF_FILE(convert.c)
#include REF_FILE(codec/write_bmp.c)
#include REF_FILE(codec/jpeg_write.c)

/* ------------------------------------------------------------------ hooks */

void ref_idct_block(unsigned char *out, int out_stride, short data[64])
{
	stbi__idct_block(out, out_stride, data);
}

/* kind: 0 = resample_row_1, 1 = v_2, 2 = h_2, 3 = hv_2, 4 = generic.  Returns the row the
 * reference would hand to the colour stage, copied into `out` (2*w or hs*w bytes). */
int ref_resample_row(int kind, unsigned char *out, unsigned char *in_near, unsigned char *in_far, int w, int hs)
{
	unsigned char *r;
	int n;
	switch (kind) {
	case 0: r = resample_row_1(out, in_near, in_far, w, hs); n = w; break;
	case 1: r = stbi__resample_row_v_2(out, in_near, in_far, w, hs); n = w; break;
	case 2: r = stbi__resample_row_h_2(out, in_near, in_far, w, hs); n = 2 * w; break;
	case 3: r = stbi__resample_row_hv_2(out, in_near, in_far, w, hs); n = 2 * w; break;
	default: r = stbi__resample_row_generic(out, in_near, in_far, w, hs); n = hs * w; break;
	}
	if (r != out)
		memmove(out, r, n);
	return n;
}

void ref_ycbcr_to_rgb_row(unsigned char *out, const unsigned char *y, const unsigned char *pcb, const unsigned char *pcr, int count, int step)
{
	stbi__YCbCr_to_RGB_row(out, y, pcb, pcr, count, step);
}

/* capture state for ref_decode_capture (single-threaded test use only) */
static short *ref__cap_coef;
static long ref__cap_coef_cap, ref__cap_coef_n;

static void ref__capturing_idct(stbi_uc *out, int out_stride, short data[64])
{
	if (ref__cap_coef && ref__cap_coef_n + 64 <= ref__cap_coef_cap) {
		memcpy(ref__cap_coef + ref__cap_coef_n, data, 64 * sizeof(short));
	}
	ref__cap_coef_n += 64;
	stbi__idct_block(out, out_stride, data);
}

/* Decodes like stbi_load_from_memory, but (a) every block handed to the IDCT seam is
 * appended (dequantised, natural order, in call order) to coef_out (capacity coef_cap
 * shorts; *coef_n receives the number the decode produced), and (b) nothing else changes.
 * Returns the malloc'd pixels or NULL. */
unsigned char *ref_decode_capture(const unsigned char *buf, int len, int *x, int *y, int *comp, int req_comp,
											 short *coef_out, long coef_cap, long *coef_n)
{
	stbi__context s;
	stbi__jpeg *j;
	unsigned char *result;
	stbi__start_mem(&s, buf, len);
	if (!stbi__jpeg_test(&s)) {
		stbi__err("unknown image type", "Image not of any known type, or corrupt");
		return NULL;
	}
	j = (stbi__jpeg *)stbi__malloc(sizeof(stbi__jpeg));
	j->s = &s;
	stbi__setup_jpeg(j);
	j->idct_block_kernel = ref__capturing_idct;
	ref__cap_coef = coef_out;
	ref__cap_coef_cap = coef_cap;
	ref__cap_coef_n = 0;
	result = load_jpeg_image(j, x, y, comp, req_comp);
	if (coef_n)
		*coef_n = ref__cap_coef_n;
	ref__cap_coef = NULL;
	STBI_FREE(j);
	return result;
}

/* memory-sink writer for ref_encode */
typedef struct { unsigned char *p; long n, cap; } ref__sink;
static void ref__sink_write(void *ctx, void *data, int size)
{
	ref__sink *k = (ref__sink *)ctx;
	if (k->n + size <= k->cap)
		memcpy(k->p + k->n, data, size);
	k->n += size;
}

/* stbi_write_jpg_to_func into a caller buffer; returns bytes produced (may exceed cap: then
 * the caller retries with a bigger buffer), or -1 if the reference returned 0. */
long ref_encode(unsigned char *dst, long cap, int w, int h, int comp, const void *rgb, int quality)
{
	ref__sink k;
	k.p = dst; k.n = 0; k.cap = cap;
	if (!stbi_write_jpg_to_func(ref__sink_write, &k, w, h, comp, rgb, quality))
		return -1;
	return k.n;
}

/* cpu_baseline helper (bench.py): decode n JPEGs `reps` times on `threads` pthreads with the
 * reference's own stbi_load_from_memory, one image per task, discarding the pixels.  The only
 * shared mutable state in the reference on this path is the failure-reason pointer (a benign
 * race).  Returns decoded pixels; *secs = wall time. */
#include <pthread.h>
#include <time.h>

typedef struct {
	const unsigned char *const *bufs;
	const int *lens;
	int n, reps, req_comp;
	long next, pixels;
	pthread_mutex_t lock;
} ref__many;

static void *ref__many_worker(void *arg)
{
	ref__many *m = (ref__many *)arg;
	long total = (long)m->n * m->reps, mine = 0;
	for (;;) {
		long i;
		int x = 0, y = 0, c = 0;
		unsigned char *p;
		pthread_mutex_lock(&m->lock);
		i = m->next++;
		pthread_mutex_unlock(&m->lock);
		if (i >= total)
			break;
		p = stbi_load_from_memory(m->bufs[i % m->n], m->lens[i % m->n], &x, &y, &c, m->req_comp);
		if (p) {
			mine += (long)x * y;
			stbi_image_free(p);
		}
	}
	pthread_mutex_lock(&m->lock);
	m->pixels += mine;
	pthread_mutex_unlock(&m->lock);
	return NULL;
}

long ref_decode_many(const unsigned char *const *bufs, const int *lens, int n, int reps, int threads, int req_comp, double *secs)
{
	ref__many m;
	pthread_t tid[256];
	struct timespec t0, t1;
	int i;
	if (threads < 1) threads = 1;
	if (threads > 256) threads = 256;
	m.bufs = bufs; m.lens = lens; m.n = n; m.reps = reps; m.req_comp = req_comp; m.next = 0; m.pixels = 0;
	pthread_mutex_init(&m.lock, NULL);
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (i = 0; i < threads; ++i) pthread_create(&tid[i], NULL, ref__many_worker, &m);
	for (i = 0; i < threads; ++i) pthread_join(tid[i], NULL);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	pthread_mutex_destroy(&m.lock);
	if (secs) *secs = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
	return m.pixels;
}
