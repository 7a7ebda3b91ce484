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

int mij_write_jpg_to_func(mjw_write_func *func, void *context, int x, int y, int comp, const void *data, int quality)
{
	mjw_plan plan;
	mij_ctx *ctx;
	mij_encoder *enc = NULL;
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
	if (mij_enc_create(ctx, 1, pix + 256, elems * 2 + 256, &enc) != MIJ_OK)
		return 0;
	du = (int16_t *)malloc(elems * sizeof(int16_t));
	slot = du ? mij_enc_add(enc, data, x, y, comp, quality, mjw_flip_on_write()) : -1;
	if (slot >= 0 && mij_enc_upload(enc) == MIJ_OK && mij_enc_launch(enc) == MIJ_OK && mij_enc_fetch(enc, slot, du, elems) == MIJ_OK)
		ok = mjw_emit(&plan, du, func, context);
	free(du);
	mij_enc_destroy(enc);
	return ok;
}
