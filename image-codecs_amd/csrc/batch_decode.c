/*
 * batch_decode.c -- mjh_decode_batch (include/mij_host.h): the host stage of many JPEGs on a
 * pthread pool, one image per task, writing straight into a mij batch's pinned staging.
 *
 * The reference is single-threaded and holds all decoder state per call (stbi__jpeg is malloc'd
 * per image, codec/jpeg.c:2445), so images are independent; the only process-global it touches on
 * this path is the failure-reason pointer, which here is per image (reasons[i]).
 */
#include <pthread.h>
#include <stdio.h>
#include <time.h>
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
	/* GPU entropy front end */
	mjg_scan *scans;
	uint8_t *stage;
	size_t *off, *cap, *slen;
	int *status;      /* mjh_extract_scan's verdict per image */
	size_t small_px;  /* default front end: pictures below this many pixels go to the host walk (0: none do) */
	const char *todo; /* per image: 1 = the host walk has to do it */
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
		if (slot < 0 || (p->todo && !p->todo[i]))
			continue;
		d = p->descs[i];
		{
			/* baseline files are staged as compact planes by the walk itself (no pack pass, half the bytes over PCIe) unless the
			 * batch was asked for int16 planes; progressive files stay int16 (mjh_decode_memory_fmt) */
			size_t bytes = 0;
			uint8_t *region = mij_batch_stage_region(p->b, slot, &bytes);
			if (region && mjh_decode_memory_fmt(p->bufs[i], p->lens[i], p->req_comp, &d, region, bytes, mij_batch_coef_format(p->b) == MIJ_COEF_COMPACT, &why)) {
				if (d.flags)
					mij_batch_set_flags(p->b, slot, d.flags);
				if (d.color != p->descs[i].color) /* a JFIF / Adobe marker behind SOF changed the colour branch */
					mij_batch_set_color(p->b, slot, d.color);
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

static void run_pool(pool_t *p, void *(*fn)(void *), int threads);

/* the host-only front end: every Huffman walk on the host threads */
int mjh_decode_batch_host(mij_batch *b, const uint8_t *const *bufs, const int *lens, int n, int req_comp, int threads, int *slots, const char **reasons)
{
	pool_t p;
	int i;
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
		slots[i] = mij_batch_add_uncleared(b, &p.descs[i]); /* the worker's mjh_decode_memory clears the planes */
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
	p.todo = NULL;
	pthread_mutex_init(&p.lock, NULL);
	run_pool(&p, worker, threads);
	pthread_mutex_destroy(&p.lock);
	free(p.descs);
	return p.ok;
}

/* ------------------------------------------------------------------ the same with the Huffman walk on the GPU */

/* pictures below this many pixels take the host walk although the GPU walk could do them: set by the default front end
 * (mjh_decode_batch) around its call, see there */
static __thread size_t t_small_px = 0;

static void *extract_worker(void *arg)
{
	pool_t *p = (pool_t *)arg;
	for (;;) {
		int i;
		const char *why = NULL;
		pthread_mutex_lock(&p->lock);
		i = p->next++;
		pthread_mutex_unlock(&p->lock);
		if (i >= p->n)
			break;
		if (p->small_px && (size_t)(p->lens[i] > 0 ? p->lens[i] : 0) < (p->small_px / 4 < 8192 ? p->small_px / 4 : 8192)) {
			/* a file this short is a small picture whatever its quality (a quarter of a byte per pixel and less; never more than 8 KiB,
			 * so that a large flat picture of a many-threaded call still has its size looked at): header only,
			 * the host walk takes it (status 2) without the stream being unstuffed for a GPU walk that will not happen */
			p->slen[i] = 0;
			p->status[i] = mjh_probe_memory(p->bufs[i], p->lens[i], p->req_comp, &p->scans[i].desc, &why) ? 2 : 0;
		} else
			p->status[i] = mjh_extract_scan(p->bufs[i], p->lens[i], p->req_comp, &p->scans[i], p->stage + p->off[i], p->cap[i], &p->slen[i], &why);
		p->reasons[i] = why;
	}
	return NULL;
}

/* (documented rather than changed: the pool runs ONE job at a time -- callers of the batch front ends on different threads queue
 * on job_lock for the length of a host stage, which is what a machine whose cores one job already fills wants; the helpers are detached and
 * live until the process ends, so this library must not be dlclose()d once a batch front end has run -- INTEGRATION.md says so.) */
/* A process-wide pool of helper threads, created on first use and kept: spawning and joining 15 threads per call cost
 * more than the work of a 128-image chunk's header stage.  One job at a time (callers queue on job_lock); a job is
 * "run fn(arg) on up to `want` helpers besides the caller" -- fn pulls work items itself, so helpers that wake up late
 * simply find nothing left, and tickets nobody took are cancelled once the caller's own fn(arg) has returned. */
static struct {
	pthread_mutex_t job_lock, m;
	pthread_cond_t cv_work, cv_done;
	void *(*fn)(void *);
	void *arg;
	int tickets, running, nthreads;
} g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, NULL, 0, 0, 0};

static void *pool_helper(void *unused)
{
	(void)unused;
	pthread_mutex_lock(&g_pool.m);
	for (;;) {
		void *(*fn)(void *);
		void *arg;
		while (g_pool.tickets == 0)
			pthread_cond_wait(&g_pool.cv_work, &g_pool.m);
		--g_pool.tickets;
		++g_pool.running;
		fn = g_pool.fn;
		arg = g_pool.arg;
		pthread_mutex_unlock(&g_pool.m);
		fn(arg);
		pthread_mutex_lock(&g_pool.m);
		if (--g_pool.running == 0 && g_pool.tickets == 0)
			pthread_cond_signal(&g_pool.cv_done);
	}
	return NULL;
}

static void run_on_pool(void *(*fn)(void *), void *arg, int threads)
{
	int want = threads - 1;
	if (want > 255)
		want = 255;
	if (want <= 0) {
		fn(arg);
		return;
	}
	pthread_mutex_lock(&g_pool.job_lock);
	pthread_mutex_lock(&g_pool.m);
	while (g_pool.nthreads < want) {
		pthread_t t;
		pthread_attr_t a;
		pthread_attr_init(&a);
		pthread_attr_setdetachstate(&a, PTHREAD_CREATE_DETACHED);
		if (pthread_create(&t, &a, pool_helper, NULL) != 0) {
			pthread_attr_destroy(&a);
			break;
		}
		pthread_attr_destroy(&a);
		++g_pool.nthreads;
	}
	if (want > g_pool.nthreads)
		want = g_pool.nthreads;
	g_pool.fn = fn;
	g_pool.arg = arg;
	g_pool.tickets = want;
	pthread_cond_broadcast(&g_pool.cv_work);
	pthread_mutex_unlock(&g_pool.m);
	fn(arg);
	pthread_mutex_lock(&g_pool.m);
	g_pool.tickets = 0; /* helpers that have not started yet would find no work left */
	while (g_pool.running > 0)
		pthread_cond_wait(&g_pool.cv_done, &g_pool.m);
	pthread_mutex_unlock(&g_pool.m);
	pthread_mutex_unlock(&g_pool.job_lock);
}

static void run_pool(pool_t *p, void *(*fn)(void *), int threads)
{
	p->next = 0;
	if (threads > p->n)
		threads = p->n > 0 ? p->n : 1;
	run_on_pool(fn, p, threads);
}

struct mjh_gpu_job {
	pool_t p;
	int threads, max_slot, *fb;
	char *todo;
};

static void job_free(mjh_gpu_job *j)
{
	if (!j)
		return;
	free(j->p.scans);
	free(j->p.descs);
	free(j->p.off);
	free(j->p.status);
	free(j->todo);
	free(j->fb);
	free(j);
}

mjh_gpu_job *mjh_decode_batch_gpu_begin(mij_batch *b, const uint8_t *const *bufs, const int *lens, int n, int req_comp, int threads, int *slots,
													 const char **reasons, int *rc_out)
{
#ifdef MIJ_TIMING_BUILD /* -DMIJ_TIMING_BUILD: per-stage wall times of this call on stderr (development builds only) */
	double t0_ = 0, t1_ = 0, t2_ = 0, t3_ = 0;
	struct timespec ts_;
#define NOW_(v) do { clock_gettime(CLOCK_MONOTONIC, &ts_); v = ts_.tv_sec * 1e3 + ts_.tv_nsec * 1e-6; } while (0)
#else
#define NOW_(v) do { } while (0)
#endif
	mjh_gpu_job *j = NULL;
	pool_t *p;
	size_t cap = 0, used = 0, cnt = (size_t)(n > 0 ? n : 1);
	uint8_t *stage;
	int i, rc = MIJ_OK;
	if (rc_out)
		*rc_out = MIJ_OK;
	if (!b || !bufs || !lens || !slots || !reasons || n < 0) {
		rc = MIJ_E_ARG;
		goto fail;
	}
	stage = mij_batch_entropy_stage(b, &cap);
	if (!stage) {
		rc = MIJ_E_STATE; /* no entropy arena reserved */
		goto fail;
	}
	if (threads < 1)
		threads = 1;
	if (threads > 256)
		threads = 256;
	j = (mjh_gpu_job *)calloc(1, sizeof *j);
	if (!j) {
		rc = MIJ_E_NOMEM;
		goto fail;
	}
	p = &j->p;
	j->threads = threads;
	j->max_slot = -1;
	p->b = b;
	p->bufs = bufs;
	p->lens = lens;
	p->n = n;
	p->req_comp = req_comp;
	p->slots = slots;
	p->reasons = reasons;
	p->stage = stage;
	p->small_px = t_small_px; /* 0 from the explicit GPU-walk entries: they walk what they are given */
	p->scans = (mjg_scan *)malloc(sizeof(mjg_scan) * cnt);
	p->descs = (mij_image_desc *)malloc(sizeof(mij_image_desc) * cnt);
	p->off = (size_t *)malloc(sizeof(size_t) * 3 * cnt);
	p->status = (int *)malloc(sizeof(int) * cnt);
	j->todo = (char *)calloc(cnt, 1);
	j->fb = (int *)malloc(sizeof(int) * cnt);
	if (!p->scans || !p->descs || !p->off || !p->status || !j->todo || !j->fb) {
		rc = MIJ_E_NOMEM;
		goto fail;
	}
	p->cap = p->off + n;
	p->slen = p->cap + n;
	/* every image gets a region a little larger than its file: the unstuffed data cannot be longer, and each
	 * restart interval adds about 40 bytes of padding and table (files with intervals shorter than ~350 bytes
	 * do not fit and take the host walk) */
	for (i = 0; i < n; ++i) {
		const size_t len = (size_t)(lens[i] > 0 ? lens[i] : 0);
		const size_t need = (len + len / 8 + 4096 + 255) / 256 * 256;
		p->off[i] = used;
		p->cap[i] = need;
		used += need;
	}
	if (used > cap) {
		rc = MIJ_E_NOMEM;
		goto fail;
	}
	NOW_(t0_);
	pthread_mutex_init(&p->lock, NULL);
	run_pool(p, extract_worker, threads);
	pthread_mutex_destroy(&p->lock);
	NOW_(t1_);
	/* slots in input order */
	const size_t small_px = p->small_px;
	for (i = 0; i < n; ++i) {
		p->descs[i] = p->scans[i].desc;
		if (p->status[i] == 0) {
			slots[i] = -1;
			continue;
		}
		reasons[i] = NULL;
		/* small pictures are quicker on the host threads: a scan of a few subsequences leaves most of the 256 lanes of its workgroup
		 * idle in every pass of the GPU walk.  Measured per call, 16 host threads (tools/bench_batch_sizes.py): 64 x 64 0.73 against
		 * 1.83 Gpix/s, 128 x 128 2.6 against 4.7, 256 x 256 9.9 against 5.6 -- the GPU walk's rate grows with the picture, the host's
		 * with the threads, and they cross near 2200 pixels per thread. */
		if (p->status[i] == 1 && (size_t)p->descs[i].width * (size_t)p->descs[i].height < small_px)
			p->status[i] = 2;
		if (p->status[i] == 1) {
			slots[i] = mij_batch_add_stream(b, &p->scans[i], stage + p->off[i], p->slen[i]);
			if (slots[i] == MIJ_E_NOMEM) /* e.g. more restart intervals than the entropy arena has scans for: host walk */
				p->status[i] = 2;
		}
		if (p->status[i] != 1) {
			slots[i] = mij_batch_add_uncleared(b, &p->descs[i]);
			j->todo[i] = 1;
		}
		if (slots[i] < 0) {
			rc = slots[i];
			goto fail;
		}
		if (slots[i] > j->max_slot)
			j->max_slot = slots[i];
	}
	NOW_(t2_);
	rc = mij_batch_entropy_launch(b);
	NOW_(t3_);
#ifdef MIJ_TIMING_BUILD
	fprintf(stderr, "gpu_begin: extract %.3f ms, add_stream %.3f ms, launch %.3f ms\n", t1_ - t0_, t2_ - t1_, t3_ - t2_);
#endif
	if (rc != MIJ_OK)
		goto fail;
	return j;
fail:
	job_free(j);
	if (rc_out)
		*rc_out = rc;
	return NULL;
}

int mjh_decode_batch_gpu_end(mjh_gpu_job *j)
{
	pool_t *p;
	int i, rc, n_fb = 0, gpu_ok = 0, *img_of_slot = NULL;
	if (!j)
		return MIJ_E_ARG;
	p = &j->p;
	rc = mij_batch_entropy_finish(p->b, j->fb, p->n > 0 ? p->n : 1, &n_fb);
	if (rc != MIJ_OK)
		goto out;
	if (n_fb > 0) {
		img_of_slot = (int *)malloc(sizeof(int) * (size_t)(j->max_slot + 1));
		if (!img_of_slot) {
			rc = MIJ_E_NOMEM;
			goto out;
		}
		for (i = 0; i <= j->max_slot; ++i)
			img_of_slot[i] = -1;
		for (i = 0; i < p->n; ++i)
			if (p->slots[i] >= 0)
				img_of_slot[p->slots[i]] = i;
		for (i = 0; i < n_fb; ++i) {
			const int img = j->fb[i] >= 0 && j->fb[i] <= j->max_slot ? img_of_slot[j->fb[i]] : -1;
			if (img < 0) { /* a scan of the arena that is not one of this call's pictures (mjh_decode_batch_gpu_begin on a batch that was not reset) */
				rc = MIJ_E_STATE;
				goto out;
			}
			if (mij_batch_fallback_prepare(p->b, j->fb[i]) != MIJ_OK) {
				/* no staging planes left for the host walk of this one image: it alone fails, the batch goes on */
				mij_batch_set_flags(p->b, j->fb[i], MIJ_FLAG_SKIP);
				p->reasons[img] = "outofmem";
				p->slots[img] = -1 - j->fb[i];
				p->status[img] = 0;
				continue;
			}
			j->todo[img] = 1;
		}
	}
	for (i = 0; i < p->n; ++i)
		if (p->status[i] == 1 && !j->todo[i])
			++gpu_ok;
	/* the host walk for what the GPU did not take (usually nothing) */
	p->todo = j->todo;
	p->ok = 0;
	for (i = 0; i < p->n; ++i)
		if (p->slots[i] >= 0 && j->todo[i])
			break;
	if (i < p->n) {
		pthread_mutex_init(&p->lock, NULL);
		run_pool(p, worker, j->threads);
		pthread_mutex_destroy(&p->lock);
	}
	rc = gpu_ok + p->ok;
out:
	free(img_of_slot);
	job_free(j);
	return rc;
}

int mjh_decode_batch_gpu(mij_batch *b, const uint8_t *const *bufs, const int *lens, int n, int req_comp, int threads, int *slots, const char **reasons)
{
	int rc = MIJ_OK;
	mjh_gpu_job *j = mjh_decode_batch_gpu_begin(b, bufs, lens, n, req_comp, threads, slots, reasons, &rc);
	if (!j) {
		if (rc == MIJ_E_STATE) /* no entropy arena reserved: the host-only front end */
			return mjh_decode_batch_host(b, bufs, lens, n, req_comp, threads, slots, reasons);
		return rc;
	}
	return mjh_decode_batch_gpu_end(j);
}

/* bytes of entropy arena a list of files needs (mjh_decode_batch_gpu_begin gives every image a region a little larger
 * than its file) */
static size_t entropy_bytes_for(const int *lens, int n)
{
	size_t used = 0;
	int i;
	for (i = 0; i < n; ++i) {
		const size_t len = (size_t)(lens[i] > 0 ? lens[i] : 0);
		used += (len + len / 8 + 4096 + 255) / 256 * 256;
	}
	return used;
}

/* 1: the Huffman walk runs on the GPU by default, host walk as the fallback (MIJ_ENTROPY=host turns that off) */
int mjh_gpu_walk_default(void)
{
	const char *e = getenv("MIJ_ENTROPY");
	return !(e && !strcmp(e, "host"));
}

/* The default front end: the Huffman walk on the GPU wherever it applies.  The batch gets its entropy arena on first
 * use, sized for this call; a later call that needs more than the arena holds takes the host walk. */
int mjh_decode_batch(mij_batch *b, const uint8_t *const *bufs, const int *lens, int n, int req_comp, int threads, int *slots, const char **reasons)
{
	size_t cap = 0;
	if (!b || !bufs || !lens || !slots || !reasons || n < 0)
		return MIJ_E_ARG;
	/* a batch that already holds pictures of an earlier call (no reset in between) is extended through the host walk: the GPU walk's
	 * finish step looks at every scan of the arena, and only this call's slots have an owner here */
	if (!mjh_gpu_walk_default() || n == 0 || mij_batch_image_count(b) != 0)
		return mjh_decode_batch_host(b, bufs, lens, n, req_comp, threads, slots, reasons);
	{
		/* a call made of nothing but files so short that each would go to the host walk unseen (extract_worker's rule) is the host front end's:
		 * no entropy arena, no second pass over the list -- a batch of thumbnails (64 x 64: 1.3 -> 1.9 Gpix/s per call) */
		const char *small_env = getenv("MIJ_GPU_WALK_BATCH_MIN_PIXELS");
		const size_t px = small_env ? (size_t)strtoull(small_env, NULL, 10) : (size_t)2200 * (size_t)(threads < 1 ? 1 : (threads > 256 ? 256 : threads));
		const size_t shortest = px / 4 < 8192 ? px / 4 : 8192;
		int i;
		for (i = 0; i < n; ++i)
			if ((size_t)(lens[i] > 0 ? lens[i] : 0) >= shortest)
				break;
		if (i == n)
			return mjh_decode_batch_host(b, bufs, lens, n, req_comp, threads, slots, reasons);
	}
	if (!mij_batch_entropy_stage(b, &cap)) {
		if (mij_batch_image_count(b) != 0 || mij_batch_entropy_reserve(b, entropy_bytes_for(lens, n) + 4096) != MIJ_OK)
			return mjh_decode_batch_host(b, bufs, lens, n, req_comp, threads, slots, reasons);
		(void)mij_batch_entropy_stage(b, &cap);
	}
	if (entropy_bytes_for(lens, n) > cap)
		return mjh_decode_batch_host(b, bufs, lens, n, req_comp, threads, slots, reasons);
	{
		const char *small_env = getenv("MIJ_GPU_WALK_BATCH_MIN_PIXELS");
		int rc;
		t_small_px = small_env ? (size_t)strtoull(small_env, NULL, 10) : (size_t)2200 * (size_t)(threads < 1 ? 1 : (threads > 256 ? 256 : threads));
		rc = mjh_decode_batch_gpu(b, bufs, lens, n, req_comp, threads, slots, reasons);
		t_small_px = 0;
		return rc;
	}
}

/* ------------------------------------------------------------------ one logical batch over several devices
 *
 * BASELINE config 3 / north_star: "a batch of independent images is sharded across the 8 GPUs of one node on
 * separate HIP streams (embarrassingly parallel, so no RCCL)".  Decoder state is per call in the reference
 * (stbi__jpeg is malloc'd per image, codec/jpeg.c:2445), so image i simply goes to batch
 * owner(i) = the contiguous slice it falls into; every batch has its own context (device), stream and arenas, the
 * host threads are one shared pool.  Slices are walked in order, and the thread that finishes the last image of a
 * slice submits that batch at once: device k uploads and decodes while the pool walks slice k+1.
 */
typedef struct {
	pool_t p;            /* b unused: per image batches below */
	mij_batch *const *batches;
	const int *owner;
	int n_batches;
	int *remaining;      /* per batch: images not yet walked */
	int *submit_rc;      /* per batch: result of its submit */
} multi_t;

static void *multi_worker(void *arg)
{
	multi_t *m = (multi_t *)arg;
	pool_t *p = &m->p;
	int good = 0;
	for (;;) {
		int i, slot, k, last;
		const char *why = NULL;
		mij_image_desc d;
		pthread_mutex_lock(&p->lock);
		i = p->next++;
		pthread_mutex_unlock(&p->lock);
		if (i >= p->n)
			break;
		k = m->owner[i];
		slot = p->slots[i];
		if (slot >= 0) {
			mij_batch *b = m->batches[k];
			size_t bytes = 0;
			uint8_t *region = mij_batch_stage_region(b, slot, &bytes);
			d = p->descs[i];
			if (region && mjh_decode_memory_fmt(p->bufs[i], p->lens[i], p->req_comp, &d, region, bytes, mij_batch_coef_format(b) == MIJ_COEF_COMPACT, &why)) {
				if (d.flags)
					mij_batch_set_flags(b, slot, d.flags);
				if (d.color != p->descs[i].color)
					mij_batch_set_color(b, slot, d.color);
				++good;
			} else {
				mij_batch_set_flags(b, slot, MIJ_FLAG_SKIP);
				p->reasons[i] = why ? why : "decode failed";
				p->slots[i] = -1 - slot;
			}
		}
		pthread_mutex_lock(&p->lock);
		last = --m->remaining[k] == 0;
		pthread_mutex_unlock(&p->lock);
		if (last && mij_batch_image_count(m->batches[k]) > 0)
			m->submit_rc[k] = mij_batch_submit(m->batches[k]); /* asynchronous on that batch's own stream */
	}
	pthread_mutex_lock(&p->lock);
	p->ok += good;
	pthread_mutex_unlock(&p->lock);
	return NULL;
}

int mjh_decode_batch_multi(mij_batch *const *batches, int n_batches, const uint8_t *const *bufs, const int *lens, int n, int req_comp, int threads,
									int *owner, int *slots, const char **reasons)
{
	multi_t m;
	pool_t *p = &m.p;
	int i, k, rc = MIJ_OK;
	if (!batches || n_batches < 1 || n_batches > 64 || !bufs || !lens || !owner || !slots || !reasons || n < 0)
		return MIJ_E_ARG;
	for (k = 0; k < n_batches; ++k)
		if (!batches[k])
			return MIJ_E_ARG;
	if (threads < 1)
		threads = 1;
	if (threads > 256)
		threads = 256;
	memset(&m, 0, sizeof m);
	p->descs = (mij_image_desc *)malloc(sizeof(mij_image_desc) * (size_t)(n > 0 ? n : 1));
	m.remaining = (int *)calloc((size_t)n_batches, sizeof(int));
	m.submit_rc = (int *)calloc((size_t)n_batches, sizeof(int));
	if (!p->descs || !m.remaining || !m.submit_rc) {
		rc = MIJ_E_NOMEM;
		goto out;
	}
	/* contiguous slices, sizes differing by at most one (the same rule as image-codecs_amd/sharding.py shard_range);
	 * headers in input order, so slot assignment inside every batch is deterministic */
	{
		const int base = n / n_batches, rem = n % n_batches;
		int lo = 0;
		for (k = 0; k < n_batches; ++k) {
			const int cnt = base + (k < rem ? 1 : 0);
			for (i = lo; i < lo + cnt; ++i)
				owner[i] = k;
			m.remaining[k] = cnt;
			lo += cnt;
		}
	}
	for (i = 0; i < n; ++i) {
		const char *why = NULL;
		reasons[i] = NULL;
		if (!mjh_probe_memory(bufs[i], lens[i], req_comp, &p->descs[i], &why)) {
			slots[i] = -1;
			reasons[i] = why;
			continue;
		}
		slots[i] = mij_batch_add_uncleared(batches[owner[i]], &p->descs[i]);
		if (slots[i] < 0) {
			rc = slots[i];
			goto out;
		}
	}
	p->bufs = bufs;
	p->lens = lens;
	p->n = n;
	p->req_comp = req_comp;
	p->slots = slots;
	p->reasons = reasons;
	m.batches = batches;
	m.owner = owner;
	m.n_batches = n_batches;
	pthread_mutex_init(&p->lock, NULL);
	run_pool(p, multi_worker, threads);
	pthread_mutex_destroy(&p->lock);
	rc = p->ok;
	for (k = 0; k < n_batches; ++k)
		if (m.submit_rc[k] != MIJ_OK)
			rc = m.submit_rc[k];
out:
	free(p->descs);
	free(m.remaining);
	free(m.submit_rc);
	return rc;
}
