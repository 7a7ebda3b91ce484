/*
 * gpu_write.c -- mij_write_jpg_to_func: the JPEG writer with its transform stage on the GPU.
 * plan (host) -> mij_enc_* (colour + subsample + fDCT + quantiser, HIP) -> mjw_emit (host Huffman).
 * Produces the same bytes as stbi_write_jpg_to_func / the reference (codec/jpeg_write.c:368).
 */
#include <pthread.h>
#include <stdlib.h>

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
} pooled_enc;
static pooled_enc g_pool[MJW_POOL_MAX];
static int g_pool_n = 0;

static int pool_take(mij_ctx *ctx, size_t pix, size_t dub, pooled_enc *out)
{
	int i, best = -1;
	pthread_mutex_lock(&g_lock);
	for (i = 0; i < g_pool_n; ++i)
		if (g_pool[i].pix_cap >= pix && g_pool[i].du_cap >= dub && (best < 0 || g_pool[i].pix_cap < g_pool[best].pix_cap))
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
	out->enc = NULL;
	if (mij_enc_create(ctx, 1, out->pix_cap, out->du_cap, &out->enc) != MIJ_OK) {
		out->pix_cap = pix + 256;
		out->du_cap = dub + 256;
		if (mij_enc_create(ctx, 1, out->pix_cap, out->du_cap, &out->enc) != MIJ_OK)
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
	pix = (size_t)x * (size_t)y * (size_t)comp;
	if (!pool_take(ctx, pix + 256, elems * 2 + 256, &pe))
		return 0;
	du = (int16_t *)malloc(elems * sizeof(int16_t));
	slot = du ? mij_enc_add(pe.enc, data, x, y, comp, quality, mjw_flip_on_write()) : -1;
	if (slot >= 0 && mij_enc_upload(pe.enc) == MIJ_OK && mij_enc_launch(pe.enc) == MIJ_OK && mij_enc_fetch(pe.enc, slot, du, elems) == MIJ_OK)
		ok = mjw_emit(&plan, du, func, context);
	free(du);
	pool_give(&pe);
	return ok;
}
