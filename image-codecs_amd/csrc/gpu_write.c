/*
 * gpu_write.c -- mij_write_jpg_to_func: the JPEG writer with its transform stage on the GPU.
 * plan (host) -> mij_enc_* (colour + subsample + fDCT + quantiser, HIP) -> mjw_emit (host Huffman).
 * Produces the same bytes as stbi_write_jpg_to_func / the reference (codec/jpeg_write.c:368).
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "mij.h"
#include "mij_host.h"

static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static mij_ctx *g_ctx = NULL;
static int g_failed = 0;

static mij_ctx *writer_ctx(void)
{
	mij_ctx *c;
	pthread_mutex_lock(&g_lock);
	if (!g_ctx && !g_failed) {
		const char *env = getenv("MIJ_DEVICE");
		if (mij_ctx_create(env ? atoi(env) : -1, &g_ctx) != MIJ_OK) {
			g_ctx = NULL;
			g_failed = 1;
		}
	}
	c = g_ctx;
	pthread_mutex_unlock(&g_lock);
	return c;
}

/* Encoders are kept between calls: creating one costs a handful of device and pinned-host allocations (3-4 ms, more than the whole
 * call for a 512 x 512 picture).  A call takes the smallest pooled encoder that is large enough, or creates one with some slack, and
 * puts it back when it is done; nothing is tied to the calling thread. */
#define MJW_POOL_MAX 16
typedef struct {
	mij_encoder *enc;
	size_t pix_cap, du_cap;
	int max_images;
} pooled_enc;
static pooled_enc g_pool[MJW_POOL_MAX];
static int g_pool_n = 0;

static int pool_take(mij_ctx *ctx, int images, size_t pix, size_t dub, pooled_enc *out)
{
	int i, best = -1;
	pthread_mutex_lock(&g_lock);
	for (i = 0; i < g_pool_n; ++i)
		if (g_pool[i].max_images >= images && g_pool[i].pix_cap >= pix && g_pool[i].du_cap >= dub && (best < 0 || g_pool[i].pix_cap < g_pool[best].pix_cap))
			best = i;
	if (best >= 0) {
		*out = g_pool[best];
		g_pool[best] = g_pool[--g_pool_n];
	}
	pthread_mutex_unlock(&g_lock);
	if (best >= 0) {
		if (mij_enc_reset(out->enc) == MIJ_OK)
			return 1;
		mij_enc_destroy(out->enc);
	}
	out->pix_cap = pix + pix / 4 + 4096;
	out->du_cap = dub + dub / 4 + 4096;
	out->max_images = images;
	out->enc = NULL;
	if (mij_enc_create(ctx, images, out->pix_cap, out->du_cap, &out->enc) != MIJ_OK) {
		out->pix_cap = pix + 256;
		out->du_cap = dub + 256;
		if (mij_enc_create(ctx, images, out->pix_cap, out->du_cap, &out->enc) != MIJ_OK)
			return 0;
	}
	return 1;
}

static void pool_give(pooled_enc *e)
{
	mij_encoder *drop = e->enc;
	pthread_mutex_lock(&g_lock);
	if (g_pool_n < MJW_POOL_MAX) {
		g_pool[g_pool_n++] = *e;
		drop = NULL;
	} else { /* full: keep the larger of this one and the smallest pooled one */
		int i, small = 0;
		for (i = 1; i < g_pool_n; ++i)
			if (g_pool[i].pix_cap < g_pool[small].pix_cap)
				small = i;
		if (g_pool[small].pix_cap < e->pix_cap) {
			drop = g_pool[small].enc;
			g_pool[small] = *e;
		}
	}
	pthread_mutex_unlock(&g_lock);
	if (drop)
		mij_enc_destroy(drop);
}

int mij_write_jpg_to_func(mjw_write_func *func, void *context, int x, int y, int comp, const void *data, int quality)
{
	mjw_plan plan;
	mij_ctx *ctx;
	pooled_enc pe;
	int16_t *du = NULL;
	size_t elems, pix;
	int slot, ok = 0;
	if (!func || !data || !mjw_plan_init(&plan, x, y, comp, quality))
		return 0;
	ctx = writer_ctx();
	if (!ctx)
		return 0; /* no gpu device: this entry point has no host fallback */
	elems = mjw_plan_du_count(&plan) * 64;
	pix = mij_enc_pixel_bytes(x, y, comp, quality);
	if (!pool_take(ctx, 1, pix + 256, elems * 2 + 256, &pe))
		return 0;
	du = (int16_t *)malloc(elems * sizeof(int16_t));
	slot = du ? mij_enc_add(pe.enc, data, x, y, comp, quality, mjw_flip_on_write()) : -1;
	if (slot >= 0 && mij_enc_upload(pe.enc) == MIJ_OK && mij_enc_launch(pe.enc) == MIJ_OK && mij_enc_fetch(pe.enc, slot, du, elems) == MIJ_OK)
		ok = mjw_emit(&plan, du, func, context);
	free(du);
	pool_give(&pe);
	return ok;
}

/* ------------------------------------------------------------------ a batch of pictures (BASELINE config 5 end to end)
 *
 * stbi_write_jpg_to_func once per picture spends its time in the transform (host) or in per-call traffic (GPU).  Here the
 * staging copies of all pictures run on `threads` host threads, ONE launch transforms them, one copy brings every data unit
 * back to pinned memory and the threads emit the streams (codec/jpeg_write.c:120-169) side by side. */
typedef struct {
	mij_encoder *enc;
	const void *const *pixels;
	const int *x, *y, *comp, *slot;
	unsigned char **out;
	size_t *out_len;
	int lo, hi, next, phase, ok, stage_failed;
	pthread_mutex_t lock;
} wb_job;

static void *wb_worker(void *arg)
{
	wb_job *j = (wb_job *)arg;
	int good = 0, bad = 0;
	for (;;) {
		int i;
		pthread_mutex_lock(&j->lock);
		i = j->next++;
		pthread_mutex_unlock(&j->lock);
		if (i >= j->hi)
			break;
		if (j->slot[i] < 0)
			continue;
		if (j->phase == 0) { /* pixels -> the slot's pinned staging */
			if (mij_enc_stage_pixels(j->enc, j->slot[i], j->pixels[i]) != MIJ_OK)
				++bad;
		} else { /* data units -> byte stream */
			mjw_plan plan;
			const int16_t *du = mij_enc_units(j->enc, j->slot[i]);
			if (du && mij_enc_plan(j->enc, j->slot[i], &plan) == MIJ_OK) {
				/* a coefficient takes at most 27 bits and every output byte may be a stuffed 0xFF: under 7 bytes each (untouched pages cost nothing) */
				const size_t cap = 2048 + mjw_plan_du_count(&plan) * 64 * 7;
				unsigned char *buf = (unsigned char *)malloc(cap);
				const size_t len = buf ? mjw_emit_to_memory(&plan, du, buf, cap) : 0;
				if (len) {
					unsigned char *fit = (unsigned char *)realloc(buf, len);
					j->out[i] = fit ? fit : buf;
					j->out_len[i] = len;
					++good;
				} else
					free(buf);
			}
		}
	}
	pthread_mutex_lock(&j->lock);
	j->ok += good;
	j->stage_failed += bad;
	pthread_mutex_unlock(&j->lock);
	return NULL;
}

/* phase 0 (staging) or 1 (emission) of pictures lo .. hi-1 on `threads` threads, the caller's included */
static void wb_run(wb_job *j, mij_encoder *enc, int phase, int lo, int hi, int threads)
{
	pthread_t th[64];
	int t, started = 0;
	j->enc = enc;
	j->phase = phase;
	j->lo = lo;
	j->hi = hi;
	j->next = lo;
	if (threads > 64)
		threads = 64;
	if (threads > hi - lo)
		threads = hi - lo;
	for (t = 1; t < threads; ++t)
		if (pthread_create(&th[started], NULL, wb_worker, j) == 0)
			++started;
	wb_worker(j);
	for (t = 0; t < started; ++t)
		pthread_join(th[t], NULL);
}

/* Chunks of pictures go through two encoders in turn: while the GPU and the PCIe link work on chunk c (upload, one launch, the copy
 * back -- all queued without a wait), the host threads emit chunk c-1 and stage chunk c+1. */
#define WB_CHUNK_BYTES ((size_t)200 << 20)
int mij_write_jpg_batch(const void *const *pixels, const int *x, const int *y, const int *comp, int n, int quality, int threads,
								unsigned char **out, size_t *out_len)
{
	mij_ctx *ctx;
	pooled_enc pe[2];
	wb_job j;
	size_t pix_max = 0, dub_max = 0;
	int i, *slot, *cend, nchunk = 0, rc = MIJ_OK, have[2] = {0, 0}, queued[2] = {0, 0}, img_max = 0, c;
	if (!pixels || !x || !y || !comp || !out || !out_len || n < 0)
		return MIJ_E_ARG;
	for (i = 0; i < n; ++i) {
		out[i] = NULL;
		out_len[i] = 0;
	}
	if (n == 0)
		return 0;
	ctx = writer_ctx();
	if (!ctx)
		return MIJ_E_NODEVICE;
	slot = (int *)malloc(sizeof(int) * (size_t)n);
	cend = (int *)malloc(sizeof(int) * (size_t)n);
	if (!slot || !cend) {
		free(slot);
		free(cend);
		return MIJ_E_NOMEM;
	}
	/* chunk boundaries: about WB_CHUNK_BYTES of pixels each; the encoders are sized for the largest chunk */
	{
		size_t pix = 0, dub = 0;
		int first = 0;
		for (i = 0; i < n; ++i) {
			mjw_plan plan;
			slot[i] = -1;
			if (pixels[i] && mjw_plan_init(&plan, x[i], y[i], comp[i], quality)) {
				pix += mij_enc_pixel_bytes(x[i], y[i], comp[i], quality);
				dub += (mjw_plan_du_count(&plan) * 128 + 255) / 256 * 256;
				slot[i] = 0;
			}
			if (pix >= WB_CHUNK_BYTES || i == n - 1) {
				cend[nchunk++] = i + 1;
				pix_max = pix > pix_max ? pix : pix_max;
				dub_max = dub > dub_max ? dub : dub_max;
				img_max = i + 1 - first > img_max ? i + 1 - first : img_max;
				first = i + 1;
				pix = dub = 0;
			}
		}
	}
	for (c = 0; c < (nchunk > 1 ? 2 : 1); ++c) {
		if (!pool_take(ctx, img_max, pix_max + 256, dub_max + 256, &pe[c])) {
			rc = MIJ_E_NOMEM;
			break;
		}
		have[c] = 1;
	}
	memset(&j, 0, sizeof j);
	j.pixels = pixels;
	j.x = x;
	j.y = y;
	j.comp = comp;
	j.slot = slot;
	j.out = out;
	j.out_len = out_len;
	pthread_mutex_init(&j.lock, NULL);
	if (threads < 1)
		threads = 1;
	for (c = 0; c <= nchunk && rc == MIJ_OK; ++c) {
		mij_encoder *cur = c < nchunk ? pe[c & 1].enc : NULL, *prev = c > 0 && queued[(c - 1) & 1] ? pe[(c - 1) & 1].enc : NULL;
		const int lo = c < nchunk ? (c ? cend[c - 1] : 0) : 0, hi = c < nchunk ? cend[c] : 0;
		if (cur) { /* stage chunk c and queue its GPU work */
			int added = 0;
			queued[c & 1] = 0;
			if (c >= 2)
				rc = mij_enc_reset(cur); /* its previous chunk was emitted in the last round */
			for (i = lo; i < hi && rc == MIJ_OK; ++i)
				if (slot[i] == 0) {
					slot[i] = mij_enc_add_uncopied(cur, x[i], y[i], comp[i], quality, mjw_flip_on_write());
					if (slot[i] < 0)
						rc = slot[i];
					else
						++added;
				}
			/* a chunk in which every picture was refused (NULL pixels, bad arguments) has nothing to upload: its outputs stay NULL */
			if (rc == MIJ_OK && added) {
				wb_run(&j, cur, 0, lo, hi, threads);
				if (j.stage_failed)
					rc = MIJ_E_ARG;
				if (rc == MIJ_OK)
					rc = mij_enc_upload(cur);
				if (rc == MIJ_OK)
					rc = mij_enc_launch(cur);
				if (rc == MIJ_OK)
					rc = mij_enc_fetch_all_async(cur);
				if (rc == MIJ_OK)
					queued[c & 1] = 1;
			}
		}
		if (prev && rc == MIJ_OK) { /* emit chunk c-1 while chunk c is in flight */
			rc = mij_enc_wait(prev);
			if (rc == MIJ_OK)
				wb_run(&j, prev, 1, c > 1 ? cend[c - 2] : 0, cend[c - 1], threads);
		}
	}
	pthread_mutex_destroy(&j.lock);
	for (c = 0; c < 2; ++c)
		if (have[c]) {
			(void)mij_enc_wait(pe[c].enc);
			pool_give(&pe[c]);
		}
	free(slot);
	free(cend);
	if (rc != MIJ_OK) /* an error return hands nothing over: the streams of earlier chunks are released here, not leaked */
		for (i = 0; i < n; ++i) {
			free(out[i]);
			out[i] = NULL;
			out_len[i] = 0;
		}
	return rc == MIJ_OK ? j.ok : rc;
}
