/*
 * image_api.c -- the stb-compatible load / info surface of include/image_api.h on top of the
 * host entropy decoder (jpeg_entropy.c) and the GPU back end (mij.h).
 *
 * Mirrors, for JPEG only, the reference's dispatch and post-processing:
 *   stbi__load_main                     image_api.c:3-56    (type test, then load)
 *   stbi__load_and_postprocess_8bit/16  convert.c:78-133    (8<->16 conversion, vertical flip)
 *   stbi_load* / stbi_info* / is_16_bit / is_hdr            convert.c:188-266,345-398 image_api.c:74-145
 *   load_jpeg_image's argument checks and outputs           codec/jpeg.c:2224-2249,2293,2433-2438
 * The pixel work itself (IDCT, upsample, colour) runs on the GPU; there is no CPU fallback.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "image_api.h"
#include "jpeg_entropy.h"
#include "mij.h"

static __thread const char *t_reason = NULL;
static int g_flip_on_load = 0; /* process-global like the reference's stbi__vertically_flip_on_load */

const char *stbi_failure_reason(void) { return t_reason; }
void stbi_image_free(void *p) { free(p); }
void stbi_set_flip_vertically_on_load(int flag) { g_flip_on_load = flag; }

static unsigned char *fail_ptr(const char *why)
{
	t_reason = why;
	return NULL;
}
static int fail_int(const char *why)
{
	t_reason = why;
	return 0;
}

/* ------------------------------------------------------------------ per-thread GPU batch cache */

static pthread_mutex_t g_ctx_lock = PTHREAD_MUTEX_INITIALIZER;
static mij_ctx *g_ctx = NULL;
static int g_ctx_failed = 0;

static mij_ctx *shared_ctx(void)
{
	mij_ctx *c;
	pthread_mutex_lock(&g_ctx_lock);
	if (!g_ctx && !g_ctx_failed) {
		const char *env = getenv("MIJ_DEVICE");
		int dev = env ? atoi(env) : -1;
		if (mij_ctx_create(dev, &g_ctx) != MIJ_OK) {
			g_ctx = NULL;
			g_ctx_failed = 1;
		}
	}
	c = g_ctx;
	pthread_mutex_unlock(&g_ctx_lock);
	return c;
}

typedef struct {
	mij_batch *b;
	size_t coef_cap, out_cap, stream_cap;
	uint8_t *bounce; /* pinned host buffer of out_cap bytes, allocated on first need (fetch_pixels) */
} tl_batch;

/* The calling thread's one-picture batch lives in thread-local storage while the thread lives.  When the thread ends, the batch
 * goes back to a process-wide pool instead of being destroyed: a pthread-key destructor runs AFTER the C++ thread_local
 * destructors of the thread, the HIP runtime keeps its per-thread state in such objects, and calling it from there corrupts the
 * heap (seen as "malloc_consolidate(): unaligned fastbin chunk" at process exit after a few dozen short-lived threads had each
 * made stbi_load calls).  A new thread takes a pooled batch before it creates one -- which also spares thread-pool style callers
 * the 70 MB of allocations per new thread. */
#define MIJ_POOL_MAX 64
static __thread tl_batch t_batch = {NULL, 0, 0, 0, NULL};
static pthread_key_t g_batch_key;
static pthread_once_t g_batch_key_once = PTHREAD_ONCE_INIT;
static pthread_mutex_t g_pool_lock = PTHREAD_MUTEX_INITIALIZER;
static tl_batch g_pool[MIJ_POOL_MAX];
static int g_pool_n = 0;

static void batch_key_dtor(void *p)
{
	tl_batch *t = (tl_batch *)p;
	if (!t)
		return;
	pthread_mutex_lock(&g_pool_lock);
	if (t->b && g_pool_n < MIJ_POOL_MAX)
		g_pool[g_pool_n++] = *t; /* no HIP call here; a pool that is full leaks the batch rather than risk one */
	pthread_mutex_unlock(&g_pool_lock);
	free(t);
}
static void batch_key_init(void) { pthread_key_create(&g_batch_key, batch_key_dtor); }

/* the key's value mirrors t_batch for the destructor (heap copy: the thread's static TLS block is not ours to read then) */
static void remember_thread_batch(void)
{
	tl_batch *t;
	pthread_once(&g_batch_key_once, batch_key_init);
	t = (tl_batch *)pthread_getspecific(g_batch_key);
	if (!t) {
		t = (tl_batch *)calloc(1, sizeof(*t));
		if (!t)
			return;
		pthread_setspecific(g_batch_key, t);
	}
	*t = t_batch;
}

/* a one-image batch big enough for `d` (and, stream_bytes > 0, with an entropy arena for a file of that many bytes: the GPU
 * Huffman walk), grown geometrically and reused across calls */
static mij_batch *thread_batch(mij_ctx *ctx, const mij_image_desc *d, size_t stream_bytes)
{
	size_t cb = mij_image_coef_bytes(d), ob = mij_image_out_bytes(d);
	if (!t_batch.b) { /* a batch a finished thread left behind, the roomiest one */
		int i, best = -1;
		pthread_mutex_lock(&g_pool_lock);
		for (i = 0; i < g_pool_n; ++i)
			if (best < 0 || g_pool[i].coef_cap > g_pool[best].coef_cap)
				best = i;
		if (best >= 0) {
			t_batch = g_pool[best];
			g_pool[best] = g_pool[--g_pool_n];
		}
		pthread_mutex_unlock(&g_pool_lock);
		if (t_batch.b)
			remember_thread_batch();
	}
	if (t_batch.b && cb <= t_batch.coef_cap && ob <= t_batch.out_cap && stream_bytes <= t_batch.stream_cap) {
		if (mij_batch_reset(t_batch.b) != MIJ_OK)
			return NULL;
		return t_batch.b;
	}
	if (t_batch.b) {
		if (stream_bytes < t_batch.stream_cap)
			stream_bytes = t_batch.stream_cap; /* keep what earlier calls needed */
		mij_batch_destroy(t_batch.b);
		if (t_batch.bounce)
			mij_host_free(t_batch.bounce);
		t_batch.b = NULL;
		t_batch.bounce = NULL;
		t_batch.coef_cap = t_batch.out_cap = t_batch.stream_cap = 0;
		remember_thread_batch();
	}
	{
		size_t ccap = cb + cb / 4 + 4096, ocap = ob + ob / 4 + 4096;
		mij_batch *b = NULL;
		if (mij_batch_create(ctx, 1, ccap, ccap, ocap, &b) != MIJ_OK) {
			/* retry without slack before giving up */
			if (mij_batch_create(ctx, 1, cb, cb, ob, &b) != MIJ_OK)
				return NULL;
			ccap = cb;
			ocap = ob;
		}
		if (stream_bytes) {
			stream_bytes += stream_bytes / 4 + 8192;
			if (mij_batch_entropy_reserve(b, stream_bytes) != MIJ_OK)
				stream_bytes = 0; /* no arena: the host walk does it */
		}
		t_batch.b = b;
		t_batch.coef_cap = ccap;
		t_batch.out_cap = ocap;
		t_batch.stream_cap = stream_bytes;
		remember_thread_batch();
	}
	return t_batch.b;
}

/* Pixels of the thread batch's one picture into the caller's malloc block.  A device-to-host copy into pageable memory goes
 * through one staging path inside the HIP runtime: calls from several threads queue up behind each other (eight threads decoding
 * 1080p pictures: 4 Gpix/s in all, tools/bench_threads.py).  With more than one stbi_load call in flight the pixels therefore
 * cross PCIe into the batch's own pinned buffer and the calling thread copies them out itself. */
static int g_calls_in_flight = 0;

static int fetch_pixels(mij_batch *b, int slot, unsigned char *pixels, size_t nbytes)
{
	const char *env = getenv("MIJ_FETCH_BOUNCE"); /* 0: never, 1: always; default: when other calls are in flight */
	const int bounce = env ? atoi(env) : (__atomic_load_n(&g_calls_in_flight, __ATOMIC_RELAXED) > 1);
	if (bounce && t_batch.b == b) {
		if (!t_batch.bounce) {
			t_batch.bounce = (uint8_t *)mij_host_alloc(t_batch.out_cap);
			remember_thread_batch();
		}
		if (t_batch.bounce && mij_batch_out_bytes(b) <= t_batch.out_cap) {
			const size_t off = mij_batch_out_offset(b, slot);
			if (off != (size_t)-1 && off + nbytes <= t_batch.out_cap && mij_batch_fetch_all_async(b, t_batch.bounce, t_batch.out_cap) == MIJ_OK &&
				 mij_batch_wait(b) == MIJ_OK) {
				memcpy(pixels, t_batch.bounce + off, nbytes);
				return MIJ_OK;
			}
		}
	}
	return mij_batch_fetch(b, slot, pixels, nbytes);
}

/* ------------------------------------------------------------------ load */

/* convert.c:37-61 */
static void vertical_flip(void *image, int w, int h, int bytes_per_pixel)
{
	size_t row_bytes = (size_t)w * (size_t)bytes_per_pixel;
	unsigned char *bytes = (unsigned char *)image, tmp[2048];
	int row;
	for (row = 0; row < (h >> 1); ++row) {
		unsigned char *a = bytes + (size_t)row * row_bytes, *b = bytes + (size_t)(h - row - 1) * row_bytes;
		size_t left = row_bytes;
		while (left) {
			size_t n = left < sizeof(tmp) ? left : sizeof(tmp);
			memcpy(tmp, a, n);
			memcpy(a, b, n);
			memcpy(b, tmp, n);
			a += n;
			b += n;
			left -= n;
		}
	}
}

static size_t gpu_walk_min_pixels(void)
{
	const char *e = getenv("MIJ_GPU_WALK_MIN_PIXELS");
	/* measured per call on an MI355X box (tools/bench_single.py): the host walk costs 2.0 ms per megapixel; the GPU walk of a one-picture
	 * batch took 1.5-2.2 ms + 0.2 ms per megapixel in round 2 (crossing between 1 and 1.3 megapixels: the threshold was 1280 x 1024) and takes
	 * 0.9-1.1 ms up to a megapixel since round 3's walk: 720 x 576 0.87 against 1.03 ms, 800 x 600 1.00 against 0.93, 1024 x 768 1.59
	 * against 1.09 (profiles/r03_single_call.json) */
	return e ? (size_t)strtoull(e, NULL, 10) : (size_t)800 * 600;
}

/* One image through the GPU Huffman walk: NULL when the file is not a layout the walk takes, when the walk reports the
 * stream back, or on any resource problem -- the caller then takes the host walk. */
static unsigned char *load_gpu_walk(mij_ctx *ctx, const uint8_t *buf, int len, int req_comp, const mij_image_desc *desc)
{
	mjg_scan *scan;
	mij_batch *b;
	uint8_t *stage;
	size_t cap = 0, slen = 0, nbytes;
	const char *why = NULL;
	unsigned char *pixels = NULL;
	int slot, nfb = 0, fb[1];
	if (len <= 0)
		return NULL;
	b = thread_batch(ctx, desc, (size_t)len + (size_t)len / 8 + 8192);
	if (!b || !(stage = mij_batch_entropy_stage(b, &cap)))
		return NULL;
	scan = (mjg_scan *)malloc(sizeof(*scan));
	if (!scan)
		return NULL;
	if (mjh_extract_scan(buf, len, req_comp, scan, stage, cap, &slen, &why) != 1)
		goto out;
	slot = mij_batch_add_stream(b, scan, stage, slen);
	if (slot < 0)
		goto out;
	nbytes = (size_t)scan->desc.n_out * (size_t)scan->desc.width * (size_t)scan->desc.height;
	if (nbytes > 0x7fffffffu - 1)
		goto out;
	if (mij_batch_entropy_launch(b) != MIJ_OK)
		goto out;
	/* while the GPU walks: the caller's pixel block, and its pages touched once -- a fresh 50 MB block costs 12 000 page faults, which
	 * otherwise land in the copy-out at the end of the call (round 3: the pixels are copied out of a pinned buffer, not DMA'd in) */
	pixels = (unsigned char *)malloc(nbytes + 1); /* codec/jpeg.c:2293: n * x * y + 1 bytes */
	if (pixels && nbytes >= ((size_t)1 << 20)) {
		size_t i;
		for (i = 0; i < nbytes; i += 4096)
			pixels[i] = 0;
	}
	if (mij_batch_entropy_finish(b, fb, 1, &nfb) != MIJ_OK || nfb != 0 || !pixels) {
		free(pixels);
		pixels = NULL;
		goto out;
	}
	if (mij_batch_submit(b) != MIJ_OK || fetch_pixels(b, slot, pixels, nbytes) != MIJ_OK) {
		free(pixels);
		pixels = NULL;
	}
out:
	free(scan);
	return pixels;
}

/* stbi__load_main (image_api.c:3-56) + stbi__jpeg_load / load_jpeg_image (codec/jpeg.c:2224-2452) */
static unsigned char *load_main_counted(mjh_reader *r, int *x, int *y, int *comp, int req_comp)
{
	mjh_decoder *d;
	mij_image_desc desc;
	mij_ctx *ctx;
	mij_batch *b;
	unsigned char *pixels = NULL;
	int slot;
	size_t nbytes;

	d = (mjh_decoder *)calloc(1, sizeof(*d));
	if (!d)
		return fail_ptr("outofmem");

	/* stbi__jpeg_test: SOI present?  (every other codec of the reference is out of scope) */
	if (!mjh_decode_header(d, r, MJH_SCAN_TYPE)) {
		mjh_reader_rewind(r);
		free(d);
		return fail_ptr("unknown image type");
	}
	mjh_reader_rewind(r);

	if (req_comp < 0 || req_comp > 4) {
		free(d);
		return fail_ptr("bad req_comp");
	}
	if (!mjh_decode_header(d, r, MJH_SCAN_LOAD)) {
		const char *why = d->reason;
		free(d);
		return fail_ptr(why);
	}
	if (!mjh_describe(d, req_comp, &desc)) {
		free(d);
		return fail_ptr("bad req_comp");
	}

	ctx = shared_ctx();
	if (!ctx) {
		free(d);
		return fail_ptr("no gpu device");
	}
	/* Large pictures from memory: the Huffman walk itself on the GPU (mij.h, "GPU entropy stage") when the file is a layout
	 * it takes; whatever it does not take, or reports back, is walked below exactly as before, so failure reasons and
	 * the treatment of damaged streams stay the reference's.  Small pictures are quicker on the host (the GPU walk is some
	 * thirty launches and two waits); callback and FILE sources are walked as they arrive. */
	if (r->io.read == NULL && mjh_gpu_walk_default() && (size_t)desc.width * (size_t)desc.height >= gpu_walk_min_pixels()) {
		unsigned char *px = load_gpu_walk(ctx, r->orig, (int)(r->orig_end - r->orig), req_comp, &desc);
		if (px) {
			*x = desc.width;
			*y = desc.height;
			if (comp)
				*comp = d->img_n >= 3 ? 3 : 1;
			free(d);
			return px;
		}
	}
	b = thread_batch(ctx, &desc, 0);
	if (!b) {
		free(d);
		return fail_ptr("outofmem");
	}
	slot = mij_batch_add_uncleared(b, &desc); /* mjh_attach_staging clears what the walk uses */
	if (slot < 0) {
		free(d);
		return fail_ptr("outofmem");
	}
	{
		/* baseline: the walk writes compact planes itself (what the kernels read: no pack pass, half the bytes over PCIe);
		 * progressive: int16 planes, packed on the device */
		uint8_t *region = mij_batch_stage_region(b, slot, NULL);
		if (!region) {
			free(d);
			return fail_ptr("outofmem");
		}
		mjh_attach_staging(d, &desc, region, mij_batch_coef_format(b) == MIJ_COEF_COMPACT);
	}

	if (!mjh_decode_scans(d)) {
		const char *why = d->reason;
		free(d);
		return fail_ptr(why);
	}
	if (mjh_stage_flags(d))
		mij_batch_set_flags(b, slot, mjh_stage_flags(d));
	/* is_rgb / CMYK / YCCK are decided once every marker has been seen (codec/jpeg.c:2244): APP0 / APP14 may follow SOF */
	if (mjh_color_mode(d, desc.n_out) != desc.color) {
		desc.color = mjh_color_mode(d, desc.n_out);
		if (mij_batch_set_color(b, slot, desc.color) != MIJ_OK) {
			free(d);
			return fail_ptr("gpu decode failed");
		}
	}

	/* codec/jpeg.c:2293: n * x * y + 1 bytes */
	nbytes = (size_t)desc.n_out * (size_t)desc.width * (size_t)desc.height;
	if (nbytes > 0x7fffffffu - 1) {
		free(d);
		return fail_ptr("outofmem");
	}
	pixels = (unsigned char *)malloc(nbytes + 1);
	if (!pixels) {
		free(d);
		return fail_ptr("outofmem");
	}
	if (mij_batch_submit(b) != MIJ_OK || fetch_pixels(b, slot, pixels, nbytes) != MIJ_OK) {
		free(pixels);
		free(d);
		return fail_ptr("gpu decode failed");
	}
	*x = desc.width;
	*y = desc.height;
	if (comp)
		*comp = d->img_n >= 3 ? 3 : 1;
	free(d);
	return pixels;
}

static unsigned char *load_main(mjh_reader *r, int *x, int *y, int *comp, int req_comp)
{
	unsigned char *p;
	__atomic_add_fetch(&g_calls_in_flight, 1, __ATOMIC_RELAXED);
	p = load_main_counted(r, x, y, comp, req_comp);
	__atomic_sub_fetch(&g_calls_in_flight, 1, __ATOMIC_RELAXED);
	return p;
}

/* convert.c:78-104 */
static unsigned char *load_and_postprocess_8bit(mjh_reader *r, int *x, int *y, int *comp, int req_comp)
{
	int file_comp = 0;
	unsigned char *result = load_main(r, x, y, &file_comp, req_comp);
	if (!result)
		return NULL;
	if (comp)
		*comp = file_comp;
	if (g_flip_on_load) {
		int channels = req_comp ? req_comp : file_comp;
		vertical_flip(result, *x, *y, channels);
	}
	return result;
}

/* convert.c:106-133 with stbi__convert_8_to_16 (convert.c:18-33) */
static stbi_us *load_and_postprocess_16bit(mjh_reader *r, int *x, int *y, int *comp, int req_comp)
{
	int file_comp = 0, channels, n, i;
	stbi_us *wide;
	unsigned char *result = load_main(r, x, y, &file_comp, req_comp);
	if (!result)
		return NULL;
	if (comp)
		*comp = file_comp;
	channels = req_comp == 0 ? file_comp : req_comp;
	n = *x * *y * channels;
	wide = (stbi_us *)malloc((size_t)n * 2);
	if (!wide) {
		free(result);
		return (stbi_us *)fail_ptr("outofmem");
	}
	for (i = 0; i < n; ++i)
		wide[i] = (stbi_us)((result[i] << 8) + result[i]);
	free(result);
	if (g_flip_on_load)
		vertical_flip(wide, *x, *y, channels * (int)sizeof(stbi_us));
	return wide;
}

stbi_uc *stbi_load_from_memory(stbi_uc const *buffer, int len, int *x, int *y, int *comp, int req_comp)
{
	mjh_reader r;
	mjh_reader_mem(&r, buffer, len);
	return load_and_postprocess_8bit(&r, x, y, comp, req_comp);
}

stbi_uc *stbi_load_from_callbacks(stbi_io_callbacks const *clbk, void *user, int *x, int *y, int *comp, int req_comp)
{
	mjh_reader r;
	mjh_reader_callbacks(&r, clbk, user);
	return load_and_postprocess_8bit(&r, x, y, comp, req_comp);
}

stbi_uc *stbi_load_from_file(FILE *f, int *x, int *y, int *comp, int req_comp)
{
	mjh_reader r;
	unsigned char *result;
	mjh_reader_file(&r, f);
	result = load_and_postprocess_8bit(&r, x, y, comp, req_comp);
	if (result)
		fseek(f, -(long)mjh_reader_unread(&r), SEEK_CUR); /* convert.c:208 */
	return result;
}

stbi_uc *stbi_load(char const *filename, int *x, int *y, int *comp, int req_comp)
{
	FILE *f = fopen(filename, "rb");
	unsigned char *result;
	if (!f)
		return fail_ptr("can't fopen");
	result = stbi_load_from_file(f, x, y, comp, req_comp);
	fclose(f);
	return result;
}

stbi_us *stbi_load_16_from_memory(stbi_uc const *buffer, int len, int *x, int *y, int *comp, int req_comp)
{
	mjh_reader r;
	mjh_reader_mem(&r, buffer, len);
	return load_and_postprocess_16bit(&r, x, y, comp, req_comp);
}

stbi_us *stbi_load_16_from_callbacks(stbi_io_callbacks const *clbk, void *user, int *x, int *y, int *comp, int req_comp)
{
	mjh_reader r;
	mjh_reader_callbacks(&r, clbk, user);
	return load_and_postprocess_16bit(&r, x, y, comp, req_comp);
}

stbi_us *stbi_load_from_file_16(FILE *f, int *x, int *y, int *comp, int req_comp)
{
	mjh_reader r;
	stbi_us *result;
	mjh_reader_file(&r, f);
	result = load_and_postprocess_16bit(&r, x, y, comp, req_comp);
	if (result)
		fseek(f, -(long)mjh_reader_unread(&r), SEEK_CUR);
	return result;
}

stbi_us *stbi_load_16(char const *filename, int *x, int *y, int *comp, int req_comp)
{
	FILE *f = fopen(filename, "rb");
	stbi_us *result;
	if (!f)
		return (stbi_us *)fail_ptr("can't fopen");
	result = stbi_load_from_file_16(f, x, y, comp, req_comp);
	fclose(f);
	return result;
}

/* ------------------------------------------------------------------ info (host only) */

/* stbi__info_main -> stbi__jpeg_info (codec/jpeg.c:2466-2490) */
static int info_main(mjh_reader *r, int *x, int *y, int *comp)
{
	mjh_decoder *d = (mjh_decoder *)calloc(1, sizeof(*d));
	int ok;
	if (!d)
		return fail_int("outofmem");
	ok = mjh_decode_header(d, r, MJH_SCAN_HEADER);
	if (!ok) {
		mjh_reader_rewind(r);
		free(d);
		return fail_int("unknown image type");
	}
	if (x)
		*x = d->img_x;
	if (y)
		*y = d->img_y;
	if (comp)
		*comp = d->img_n >= 3 ? 3 : 1;
	free(d);
	return 1;
}

int stbi_info_from_memory(stbi_uc const *buffer, int len, int *x, int *y, int *comp)
{
	mjh_reader r;
	mjh_reader_mem(&r, buffer, len);
	return info_main(&r, x, y, comp);
}

int stbi_info_from_callbacks(stbi_io_callbacks const *clbk, void *user, int *x, int *y, int *comp)
{
	mjh_reader r;
	mjh_reader_callbacks(&r, clbk, user);
	return info_main(&r, x, y, comp);
}

int stbi_info_from_file(FILE *f, int *x, int *y, int *comp)
{
	mjh_reader r;
	long pos = ftell(f);
	int ok;
	mjh_reader_file(&r, f);
	ok = info_main(&r, x, y, comp);
	fseek(f, pos, SEEK_SET);
	return ok;
}

int stbi_info(char const *filename, int *x, int *y, int *comp)
{
	FILE *f = fopen(filename, "rb");
	int ok;
	if (!f)
		return fail_int("can't fopen");
	ok = stbi_info_from_file(f, x, y, comp);
	fclose(f);
	return ok;
}

/* JPEG is never 16-bit (image_api.c:58-71 only asks PNG and PSD) and never HDR */
int stbi_is_16_bit_from_memory(stbi_uc const *buffer, int len)
{
	(void)buffer;
	(void)len;
	return 0;
}
int stbi_is_16_bit_from_callbacks(stbi_io_callbacks const *clbk, void *user)
{
	(void)clbk;
	(void)user;
	return 0;
}
int stbi_is_16_bit_from_file(FILE *f)
{
	(void)f;
	return 0;
}
int stbi_is_16_bit(char const *filename)
{
	FILE *f = fopen(filename, "rb");
	if (!f)
		return fail_int("can't fopen");
	fclose(f);
	return 0;
}
int stbi_is_hdr_from_memory(stbi_uc const *buffer, int len)
{
	(void)buffer;
	(void)len;
	return 0;
}
int stbi_is_hdr_from_callbacks(stbi_io_callbacks const *clbk, void *user)
{
	(void)clbk;
	(void)user;
	return 0;
}
int stbi_is_hdr_from_file(FILE *f)
{
	(void)f;
	return 0;
}
int stbi_is_hdr(char const *filename)
{
	FILE *f = fopen(filename, "rb");
	if (f)
		fclose(f);
	return 0;
}
