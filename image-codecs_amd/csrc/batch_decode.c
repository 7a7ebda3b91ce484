/*
 * batch_decode.c -- mjh_decode_batch (include/mij_host.h): the host stage of many JPEGs on a
 * pthread pool, one image per task, writing straight into a mij batch's pinned staging.
 *
 * The reference is single-threaded and holds all decoder state per call (stbi__jpeg is malloc'd
 * per image, codec/jpeg.c:2445), so images are independent; the only process-global it touches on
 * this path is the failure-reason pointer, which here is per image (reasons[i]).
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "jpeg_entropy.h"
#include "mij.h"
#include "mij_host.h"

typedef struct {
	mij_batch *b;
	const uint8_t *const *bufs;
	const int *lens;
	int n, req_comp;
	int *slots;
	const char **reasons;
	mij_image_desc *descs;
	int next, ok;
	pthread_mutex_t lock;
} pool_t;

static void *worker(void *arg)
{
	pool_t *p = (pool_t *)arg;
	int good = 0;
	for (;;) {
		int i, slot;
		const char *why = NULL;
		mij_image_desc d;
		pthread_mutex_lock(&p->lock);
		i = p->next++;
		pthread_mutex_unlock(&p->lock);
		if (i >= p->n)
			break;
		slot = p->slots[i];
		if (slot < 0)
			continue;
		d = p->descs[i];
		{
			int16_t *arena = mij_batch_coef(p->b, slot, 0);
			size_t elems = mij_image_coef_bytes(&d) / sizeof(int16_t);
			if (arena && mjh_decode_memory(p->bufs[i], p->lens[i], p->req_comp, &d, arena, elems, &why)) {
				if (d.flags)
					mij_batch_set_flags(p->b, slot, d.flags);
				++good;
			} else {
				mij_batch_set_flags(p->b, slot, MIJ_FLAG_SKIP);
				p->reasons[i] = why ? why : "decode failed";
				p->slots[i] = -1 - slot;
			}
		}
	}
	pthread_mutex_lock(&p->lock);
	p->ok += good;
	pthread_mutex_unlock(&p->lock);
	return NULL;
}

int mjh_decode_batch(mij_batch *b, const uint8_t *const *bufs, const int *lens, int n, int req_comp, int threads, int *slots, const char **reasons)
{
	pool_t p;
	pthread_t tid[256];
	int i, started = 0;
	if (!b || !bufs || !lens || !slots || !reasons || n < 0)
		return MIJ_E_ARG;
	if (threads < 1)
		threads = 1;
	if (threads > 256)
		threads = 256;
	p.descs = (mij_image_desc *)malloc(sizeof(mij_image_desc) * (size_t)(n > 0 ? n : 1));
	if (!p.descs)
		return MIJ_E_NOMEM;
	/* headers first, in order: slot assignment must be deterministic */
	for (i = 0; i < n; ++i) {
		const char *why = NULL;
		reasons[i] = NULL;
		if (!mjh_probe_memory(bufs[i], lens[i], req_comp, &p.descs[i], &why)) {
			slots[i] = -1;
			reasons[i] = why;
			continue;
		}
		slots[i] = mij_batch_add(b, &p.descs[i]);
		if (slots[i] < 0) {
			int rc = slots[i];
			free(p.descs);
			return rc; /* arenas too small: the caller sizes them from the headers */
		}
	}
	p.b = b;
	p.bufs = bufs;
	p.lens = lens;
	p.n = n;
	p.req_comp = req_comp;
	p.slots = slots;
	p.reasons = reasons;
	p.next = 0;
	p.ok = 0;
	pthread_mutex_init(&p.lock, NULL);
	if (threads > n)
		threads = n > 0 ? n : 1;
	for (i = 1; i < threads; ++i)
		if (pthread_create(&tid[started], NULL, worker, &p) == 0)
			++started;
	worker(&p);
	for (i = 0; i < started; ++i)
		pthread_join(tid[i], NULL);
	pthread_mutex_destroy(&p.lock);
	free(p.descs);
	return p.ok;
}
