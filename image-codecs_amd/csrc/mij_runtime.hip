/*
 * mij_runtime.hip -- host side of the C-ABI in include/mij.h: contexts, batches (pinned staging,
 * device arenas, one HIP stream each), upload / launch / fetch, measurement hooks.
 * The kernels are in mij_kernels.h.  No exceptions and no C++ types cross the ABI.
 */
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "mij.h"
#include "mij_kernels.h"

using namespace mij;

/* ------------------------------------------------------------------ errors */

static thread_local char g_err[256] = "";

static int set_err(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}

#define HIP_TRY(expr)                                                                         \
	do {                                                                                       \
		hipError_t e_ = (expr);                                                                 \
		if (e_ != hipSuccess)                                                                   \
			return set_err(MIJ_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));            \
	} while (0)

extern "C" const char *mij_last_error(void) { return g_err; }
extern "C" int mij_abi_version(void) { return MIJ_ABI_VERSION; }

extern "C" int mij_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

/* ------------------------------------------------------------------ context */

struct mij_ctx {
	int device;
	hipDeviceProp_t prop;
	int max_dyn_lds;
	/* pinned bounce buffer of the one-slot fetches (d2h_bounced) */
	std::mutex bounce_lock;
	uint8_t *bounce = nullptr;
	hipEvent_t bounce_ev[2] = {nullptr, nullptr};
};
static const size_t MIJ_BOUNCE_BYTES = (size_t)8 << 20;

/* Device -> caller-owned host memory for the one-slot fetches (mij_batch_fetch, mij_enc_fetch, mij_batch_fetch_coef).  The caller's
 * buffer is ordinary pageable memory, often never touched before; handing it to hipMemcpyAsync makes the runtime pin, DMA into and
 * unpin pages this library does not own (or stage through a path shared by every stream of the process).  The
 * copy therefore lands in a pinned buffer of the context, 8 MiB at a time (two halves: the next chunk's DMA runs while the calling thread
 * copies this one out): the DMA engine only ever writes memory the library allocated with hipHostMalloc.  (Round 3: the one input of the
 * twice-seen encoder-leg mismatch that was not a pure function of its arguments was a DMA into a fresh numpy buffer, DESIGN.md
 * section 8.)  Throughput paths do not come here: they copy whole arenas into pinned memory the caller got from mij_host_alloc. */
static int d2h_bounced(mij_ctx *ctx, hipStream_t st, void *dst, const void *src_dev, size_t bytes);

extern "C" int mij_ctx_create(int device, mij_ctx **out)
{
	if (!out)
		return set_err(MIJ_E_ARG, "mij_ctx_create: out is NULL");
	*out = nullptr;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0)
		return set_err(MIJ_E_NODEVICE, "no gpu device (%s)", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
	if (device < 0) {
		if (hipGetDevice(&device) != hipSuccess)
			device = 0;
	}
	if (device >= n)
		return set_err(MIJ_E_ARG, "device %d out of range (%d devices)", device, n);
	mij_ctx *c = new (std::nothrow) mij_ctx();
	if (!c)
		return set_err(MIJ_E_NOMEM, "out of host memory");
	c->device = device;
	if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&c->prop, device) != hipSuccess) {
		delete c;
		return set_err(MIJ_E_NODEVICE, "cannot open device %d", device);
	}
	if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
		/* the kernels are built for gfx950 only; any other device cannot load the code object */
		set_err(MIJ_E_NODEVICE, "device %d is %s, this library is built for gfx950 only", device, c->prop.gcnArchName);
		delete c;
		return MIJ_E_NODEVICE;
	}
	c->max_dyn_lds = 160 * 1024;
	/* allow the fused band kernels to use the whole 160 KiB of LDS for wide images */
#define MIJ_LDS_ATTR(K) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, c->max_dyn_lds)
#define MIJ_LDS_ATTR8(K)                                                                                                           \
	MIJ_LDS_ATTR((K<3, false, false>)); MIJ_LDS_ATTR((K<3, true, false>)); MIJ_LDS_ATTR((K<4, false, false>)); MIJ_LDS_ATTR((K<4, true, false>)); \
	MIJ_LDS_ATTR((K<3, false, true>)); MIJ_LDS_ATTR((K<3, true, true>)); MIJ_LDS_ATTR((K<4, false, true>)); MIJ_LDS_ATTR((K<4, true, true>))
	MIJ_LDS_ATTR8(k_fused420);
	MIJ_LDS_ATTR8(k_fused420w);
	MIJ_LDS_ATTR8(k_fused420x);
	MIJ_LDS_ATTR8(k_fused420s);
	MIJ_LDS_ATTR8(k_fused420t);
	MIJ_LDS_ATTR8(k_fused440);
	MIJ_LDS_ATTR8(k_fused440w);
	MIJ_LDS_ATTR8(k_fused420c);
	MIJ_LDS_ATTR8(k_fused440c);
	MIJ_LDS_ATTR8(k_fused422);
	MIJ_LDS_ATTR8(k_fused422w);
	MIJ_LDS_ATTR8(k_fused422x);
	MIJ_LDS_ATTR8(k_fused422s);
	MIJ_LDS_ATTR8(k_fused422t);
#undef MIJ_LDS_ATTR8
#undef MIJ_LDS_ATTR
	(void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_encode420), hipFuncAttributeMaxDynamicSharedMemorySize, MIJ_ENC_LDS);
	(void)hipGetLastError();
	*out = c;
	return MIJ_OK;
}

/* HIP gives a process four hardware queues by default and maps every further stream onto one of them; two batches that share
 * a queue run their kernels one after the other.  A ring of four batches plus one idle per-thread batch of stbi_load already loses
 * 30 % of its throughput that way (profiles/r02v_hw_queues.txt).  Unless the user chose a value, ask for eight -- this has to
 * happen before the HIP runtime initialises, hence a constructor of this library (a process that touched HIP earlier keeps its setting). */
__attribute__((constructor)) static void mij_hip_defaults() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

extern "C" void mij_ctx_destroy(mij_ctx *ctx)
{
	if (!ctx)
		return;
	if (ctx->bounce) {
		(void)hipSetDevice(ctx->device);
		(void)hipHostFree(ctx->bounce);
		for (int i = 0; i < 2; ++i)
			if (ctx->bounce_ev[i])
				(void)hipEventDestroy(ctx->bounce_ev[i]);
	}
	delete ctx;
}

static int d2h_bounced(mij_ctx *ctx, hipStream_t st, void *dst, const void *src_dev, size_t bytes)
{
	std::lock_guard<std::mutex> guard(ctx->bounce_lock);
	if (!ctx->bounce) {
		HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ctx->bounce), 2 * MIJ_BOUNCE_BYTES, hipHostMallocDefault));
		HIP_TRY(hipEventCreateWithFlags(&ctx->bounce_ev[0], hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&ctx->bounce_ev[1], hipEventDisableTiming));
	}
	if (!bytes) {
		HIP_TRY(hipStreamSynchronize(st));
		return MIJ_OK;
	}
	/* two halves: the DMA of chunk k+1 runs while the calling thread copies chunk k out (a 4096 x 4096 picture is seven chunks) */
	const size_t nchunk = (bytes + MIJ_BOUNCE_BYTES - 1) / MIJ_BOUNCE_BYTES;
	for (size_t k = 0; k <= nchunk; ++k) {
		if (k < nchunk) {
			const size_t off = k * MIJ_BOUNCE_BYTES, n = bytes - off < MIJ_BOUNCE_BYTES ? bytes - off : MIJ_BOUNCE_BYTES;
			HIP_TRY(hipMemcpyAsync(ctx->bounce + (k & 1) * MIJ_BOUNCE_BYTES, static_cast<const uint8_t *>(src_dev) + off, n, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipEventRecord(ctx->bounce_ev[k & 1], st));
		}
		if (k > 0) {
			const size_t off = (k - 1) * MIJ_BOUNCE_BYTES, n = bytes - off < MIJ_BOUNCE_BYTES ? bytes - off : MIJ_BOUNCE_BYTES;
			HIP_TRY(hipEventSynchronize(ctx->bounce_ev[(k - 1) & 1]));
			memcpy(static_cast<uint8_t *>(dst) + off, ctx->bounce + ((k - 1) & 1) * MIJ_BOUNCE_BYTES, n);
		}
	}
	return MIJ_OK;
}
extern "C" int mij_ctx_device(const mij_ctx *ctx) { return ctx ? ctx->device : -1; }

extern "C" int mij_ctx_info(const mij_ctx *ctx, char *arch, size_t arch_len, int *cu_count, size_t *total_mem)
{
	if (!ctx)
		return set_err(MIJ_E_ARG, "ctx is NULL");
	if (arch && arch_len) {
		strncpy(arch, ctx->prop.gcnArchName, arch_len - 1);
		arch[arch_len - 1] = 0;
	}
	if (cu_count)
		*cu_count = ctx->prop.multiProcessorCount;
	if (total_mem)
		*total_mem = ctx->prop.totalGlobalMem;
	return MIJ_OK;
}

/* ------------------------------------------------------------------ batch */

struct Slot {
	mij_image_desc desc;
	DevImage dev;
	size_t stage_off;  /* byte offset in the staging arena (clones: the source's) */
	size_t coef_base;  /* byte offset of this image's region in the coefficient arena */
	size_t coef_bytes; /* bytes of that region (mij_image_coef_bytes: room for either format) */
	int clone_of;      /* -1: own staging */
	int dev_coef;      /* 1: the GPU entropy stage wrote the coefficient planes in HBM; nothing to upload */
	int es_index;      /* index into the entropy arena's scan list, or -1 */
	int coef_bytes_fmt; /* 1: compact planes in HBM (low bytes + escapes + DC array), 0: int16 tile layout */
	int path;          /* 0 none, 1 fused 4:2:0, 2 two-pass, 3 fused 4:4:4, 4 fused 4:2:2, 5 fused grey, 6 fused 4:4:0 */
};

/* kernel families of a launch plan, in launch order */
/* MK_RS_FAST + RS_*: pass 2 compiled per resampler (k_resample_fast); list index [n_out == 4][YCbCr colour][0] */
enum { MK_PLANES = 0, MK_RESAMPLE, MK_RS_FAST, MK_420 = MK_RS_FAST + RS_KINDS, MK_422, MK_444, MK_GREY, MK_440, MK_420W, MK_440W /* k_fused420w / k_fused440w: 512 threads, wide pictures */, MK_420X /* 1024 threads: one workgroup per CU */, MK_420S, MK_420T /* 128 / 64 threads: narrow pictures */, MK_422W, MK_422X, MK_422S, MK_422T /* k_fused422 with 512 / 1024 / 128 / 64 threads */, MK_1X1C /* k_fused1x1c: RGB-tagged / CMYK / YCCK at 1x1 */, MK_420C, MK_440C /* column segments: a row of MCUs beyond a CU's LDS */, MK_KINDS };
struct Work4 { /* WorkBand and WorkIdct are both four u32 */
	uint32_t a, b, c, d;
};

struct mij_batch {
	mij_ctx *ctx;
	hipStream_t stream;
	hipEvent_t ev_begin, ev_end;
	hipEvent_t ev_pack0, ev_pack1; /* around k_pack_c8 in the last upload (mij_batch_pack_ms); created on first use */
	bool pack_timed;
	int max_images;
	/* arenas */
	uint8_t *stage;
	size_t stage_cap, stage_used;
	uint8_t *d_coef;
	size_t coef_cap, coef_used;
	uint8_t *d_out;
	size_t out_cap, out_used;
	uint8_t *d_planes;
	size_t planes_cap;
	uint8_t *d_up16; /* upload scratch: int16 planes on their way into compact planes (k_pack_c8), stage_cap bytes */
	uint32_t *d_l1max, *h_l1max; /* per slot: largest per-block L1 the pack kernel saw (MIJ_FLAG_L1_ON_DEVICE); max_images entries, created on first use */
	/* descriptors + work lists (pinned host mirror + device copy) */
	DevImage *h_imgs, *d_imgs;
	Work4 *h_work, *d_work;
	size_t work_cap;
	std::vector<Slot> slots;
	/* launch plan built by upload */
	struct Launch {
		int kind, nout, wide, b8;
		size_t first, count, lds;
	};
	std::vector<Launch> launches;
	bool uploaded, launched;
	int force_generic; /* 0: fused kernels where they apply; 1: two-pass path for every image; 2: and its run-time-general pass 2 */
	int coef_fmt;  /* format new coefficient planes get in HBM: 1 compact (default), 0 int16 (MIJ_COEF_FORMAT=int16, mij_batch_set_coef_format) */
	int band_rows; /* MCU rows per fused workgroup; 0 = automatic */
	struct EsArena *es; /* GPU entropy stage, allocated by mij_batch_entropy_reserve */
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static void es_free_fwd(struct EsArena *e);
static void es_reset_fwd(struct EsArena *e);

static inline size_t comp_tiles(const mij_comp_desc &cp) { return ((size_t)(cp.bw * cp.bh) + 63) >> 6; }

/* room for either format: int16 tile layout needs 8192 B per 64-block tile, compact planes 4096 (low bytes) + 128 (DC)
 * + 4096 (escape bytes) = MIJ_TILE_COMPACT_BYTES */
extern "C" size_t mij_image_coef_bytes(const mij_image_desc *d)
{
	size_t total = 0;
	for (int c = 0; c < d->ncomp; ++c)
		total += comp_tiles(d->comp[c]) * MIJ_TILE_COMPACT_BYTES;
	return total;
}

extern "C" size_t mij_image_out_bytes(const mij_image_desc *d) { return align_up((size_t)d->n_out * d->width * d->height, 256); }

extern "C" int mij_batch_create(mij_ctx *ctx, int max_images, size_t stage_bytes, size_t coef_bytes, size_t out_bytes, mij_batch **out)
{
	if (!ctx || !out || max_images <= 0)
		return set_err(MIJ_E_ARG, "mij_batch_create: bad argument");
	*out = nullptr;
	HIP_TRY(hipSetDevice(ctx->device));
	mij_batch *b = new (std::nothrow) mij_batch();
	if (!b)
		return set_err(MIJ_E_NOMEM, "out of host memory");
	b->ctx = ctx;
	b->max_images = max_images;
	b->stage = nullptr;
	b->d_coef = b->d_out = b->d_planes = nullptr;
	b->h_imgs = b->d_imgs = nullptr;
	b->h_work = b->d_work = nullptr;
	b->d_up16 = nullptr;
	b->d_l1max = b->h_l1max = nullptr;
	b->stage_cap = stage_bytes;
	b->coef_cap = coef_bytes;
	b->out_cap = out_bytes;
	b->stage_used = b->coef_used = b->out_used = 0;
	b->planes_cap = 0;
	b->work_cap = 0;
	b->uploaded = b->launched = false;
	b->force_generic = 0;
	b->band_rows = 0;
	{
		const char *fmt = getenv("MIJ_COEF_FORMAT");
		b->coef_fmt = (fmt && (!strcmp(fmt, "int16") || !strcmp(fmt, "0"))) ? 0 : 1;
	}
	b->es = nullptr;
	b->stream = nullptr;
	b->ev_begin = b->ev_end = nullptr;
	b->ev_pack0 = b->ev_pack1 = nullptr;
	b->pack_timed = false;
	const char *env = getenv("MIJ_BAND_ROWS");
	if (env)
		b->band_rows = atoi(env);

	hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
	if (e == hipSuccess)
		e = hipEventCreate(&b->ev_begin);
	if (e == hipSuccess)
		e = hipEventCreate(&b->ev_end);
	if (e == hipSuccess && stage_bytes)
		e = hipHostMalloc(reinterpret_cast<void **>(&b->stage), stage_bytes, hipHostMallocDefault);
	if (e == hipSuccess && coef_bytes)
		e = hipMalloc(reinterpret_cast<void **>(&b->d_coef), coef_bytes);
	if (e == hipSuccess && out_bytes)
		e = hipMalloc(reinterpret_cast<void **>(&b->d_out), out_bytes);
	if (e == hipSuccess)
		e = hipHostMalloc(reinterpret_cast<void **>(&b->h_imgs), sizeof(DevImage) * (size_t)max_images, hipHostMallocDefault);
	if (e == hipSuccess)
		e = hipMalloc(reinterpret_cast<void **>(&b->d_imgs), sizeof(DevImage) * (size_t)max_images);
	if (e != hipSuccess) {
		int code = (e == hipErrorOutOfMemory) ? MIJ_E_NOMEM : MIJ_E_HIP;
		set_err(code, "mij_batch_create: %s", hipGetErrorString(e));
		mij_batch_destroy(b);
		return code;
	}
	b->slots.reserve((size_t)max_images);
	*out = b;
	return MIJ_OK;
}

extern "C" void mij_batch_destroy(mij_batch *b)
{
	if (!b)
		return;
	(void)hipSetDevice(b->ctx->device);
	if (b->stream)
		(void)hipStreamSynchronize(b->stream);
	if (b->stage)
		(void)hipHostFree(b->stage);
	if (b->d_coef)
		(void)hipFree(b->d_coef);
	if (b->d_out)
		(void)hipFree(b->d_out);
	if (b->d_planes)
		(void)hipFree(b->d_planes);
	if (b->h_imgs)
		(void)hipHostFree(b->h_imgs);
	if (b->d_imgs)
		(void)hipFree(b->d_imgs);
	if (b->h_work)
		(void)hipHostFree(b->h_work);
	if (b->d_work)
		(void)hipFree(b->d_work);
	if (b->d_up16)
		(void)hipFree(b->d_up16);
	if (b->d_l1max)
		(void)hipFree(b->d_l1max);
	if (b->h_l1max)
		(void)hipHostFree(b->h_l1max);
	if (b->es)
		es_free_fwd(b->es);
	if (b->ev_begin)
		(void)hipEventDestroy(b->ev_begin);
	if (b->ev_end)
		(void)hipEventDestroy(b->ev_end);
	if (b->ev_pack0)
		(void)hipEventDestroy(b->ev_pack0);
	if (b->ev_pack1)
		(void)hipEventDestroy(b->ev_pack1);
	if (b->stream)
		(void)hipStreamDestroy(b->stream);
	delete b;
}

extern "C" int mij_batch_reset(mij_batch *b)
{
	if (!b)
		return set_err(MIJ_E_ARG, "batch is NULL");
	HIP_TRY(hipSetDevice(b->ctx->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	b->slots.clear();
	b->stage_used = b->coef_used = b->out_used = 0;
	b->uploaded = b->launched = false;
	b->launches.clear();
	es_reset_fwd(b->es);
	return MIJ_OK;
}

static int check_desc(const mij_image_desc *d)
{
	if (!d)
		return set_err(MIJ_E_ARG, "descriptor is NULL");
	if (d->width <= 0 || d->height <= 0 || d->width > 65535 || d->height > 65535)
		return set_err(MIJ_E_ARG, "bad image size %dx%d", d->width, d->height);
	if (!(d->ncomp == 1 || d->ncomp == 3 || d->ncomp == 4))
		return set_err(MIJ_E_ARG, "bad component count %d", d->ncomp);
	if (d->n_out < 1 || d->n_out > 4)
		return set_err(MIJ_E_ARG, "bad n_out %d", d->n_out);
	if (d->color < MIJ_COLOR_GREY || d->color > MIJ_COLOR_YCBCRA)
		return set_err(MIJ_E_ARG, "bad colour mode %d", d->color);
	if ((d->color == MIJ_COLOR_YCBCR || d->color == MIJ_COLOR_RGB) && d->ncomp != 3)
		return set_err(MIJ_E_ARG, "colour mode %d needs 3 components", d->color);
	if (d->color >= MIJ_COLOR_CMYK && d->ncomp != 4)
		return set_err(MIJ_E_ARG, "colour mode %d needs 4 components", d->color);
	if (d->h_max < 1 || d->h_max > 4 || d->v_max < 1 || d->v_max > 4 || d->mcu_x <= 0 || d->mcu_y <= 0)
		return set_err(MIJ_E_ARG, "bad MCU geometry");
	for (int c = 0; c < d->ncomp; ++c) {
		const mij_comp_desc &cp = d->comp[c];
		if (cp.h < 1 || cp.h > 4 || cp.v < 1 || cp.v > 4 || cp.tq < 0 || cp.tq > 3)
			return set_err(MIJ_E_ARG, "bad sampling/table for component %d", c);
		if (cp.h > d->h_max || cp.v > d->v_max)
			return set_err(MIJ_E_ARG, "component %d sampling exceeds h_max/v_max", c);
		if (cp.bw != d->mcu_x * cp.h || cp.bh != d->mcu_y * cp.v)
			return set_err(MIJ_E_ARG, "component %d block grid does not match the MCU grid", c);
		if (cp.x <= 0 || cp.y <= 0 || cp.x > cp.bw * 8 || cp.y > cp.bh * 8)
			return set_err(MIJ_E_ARG, "component %d effective size out of range", c);
	}
	/* the MCU grid covers the picture (codec/jpeg.c:1618-1622): the kernels address pixels from MCU coordinates */
	if ((int64_t)d->width > (int64_t)d->mcu_x * 8 * d->h_max || (int64_t)d->height > (int64_t)d->mcu_y * 8 * d->v_max)
		return set_err(MIJ_E_ARG, "image larger than its MCU grid");
	return MIJ_OK;
}

/* where the component planes of a slot lie inside its region of the coefficient arena, for its format */
static void layout_coef(Slot &s)
{
	size_t off = s.coef_base;
	if (s.coef_bytes_fmt)
		s.dev.flags |= MIJ_DEV_COEF_BYTES;
	else
		s.dev.flags &= ~(int32_t)MIJ_DEV_COEF_BYTES;
	for (int c = 0; c < s.desc.ncomp; ++c) {
		const size_t nt = comp_tiles(s.desc.comp[c]);
		DevComp &dc = s.dev.comp[c];
		if (s.coef_bytes_fmt) { /* mij.h, mij_compact_offsets: low bytes + DC of every component first, the escape bytes behind them */
			size_t lo, dcv, hi;
			mij_compact_offsets(&s.desc, c, &lo, &dcv, &hi);
			dc.coef_off = s.coef_base + lo;
			dc.dc_off = s.coef_base + dcv;
			dc.hi_off = s.coef_base + hi;
		} else {
			dc.coef_off = off;
			dc.dc_off = dc.hi_off = 0;
			off += nt << 13;
		}
	}
}

static void fill_dev_image(Slot &s, size_t out_off)
{
	const mij_image_desc &d = s.desc;
	DevImage &v = s.dev;
	memset(&v, 0, sizeof(v));
	v.width = d.width;
	v.height = d.height;
	v.n_out = d.n_out;
	v.color = d.color;
	v.ncomp = d.ncomp;
	v.flags = (int32_t)d.flags;
	v.mcu_x = d.mcu_x;
	v.mcu_y = d.mcu_y;
	v.out_off = out_off;
	size_t plane_off = 0;
	for (int c = 0; c < d.ncomp; ++c) {
		DevComp &dc = v.comp[c];
		const mij_comp_desc &cp = d.comp[c];
		dc.h = cp.h;
		dc.v = cp.v;
		dc.x = cp.x;
		dc.y = cp.y;
		dc.bw = cp.bw;
		dc.bh = cp.bh;
		dc.hs = d.h_max / cp.h;
		dc.vs = d.v_max / cp.v;
		dc.plane_off = plane_off; /* relative; rebased at launch */
		plane_off += align_up((size_t)cp.bw * 8 * cp.bh * 8, 256);
		/* quantisation table, natural order -> in-block position order P = 8*col + rowslot[row] */
		uint16_t q[64];
		for (int row = 0; row < 8; ++row)
			for (int col = 0; col < 8; ++col)
				q[8 * col + mij_rowslot[row]] = d.dequant[cp.tq][8 * row + col];
		for (int i = 0; i < 32; ++i)
			v.dq[c][i] = (uint32_t)q[2 * i] | ((uint32_t)q[2 * i + 1] << 16);
	}
	v.plane_bytes_total = plane_off;
}

#define MIJ_NO_STAGE ((size_t)-1)

/* lazy_stage: a slot of the GPU entropy stage -- its staging planes are only needed if the host walk has to
 * redo it, so they are neither required nor cleared here (mij_batch_fallback_prepare does that) */
static int add_common(mij_batch *b, const mij_image_desc *d, int clone_of, bool lazy_stage = false, bool clear = true)
{
	if ((int)b->slots.size() >= b->max_images)
		return set_err(MIJ_E_NOMEM, "batch is full (%d images)", b->max_images);
	const size_t cbytes = mij_image_coef_bytes(d), obytes = mij_image_out_bytes(d);
	if (b->coef_used + cbytes > b->coef_cap)
		return set_err(MIJ_E_NOMEM, "coefficient arena exhausted");
	if (b->out_used + obytes > b->out_cap)
		return set_err(MIJ_E_NOMEM, "output arena exhausted");
	Slot s;
	s.desc = *d;
	s.clone_of = clone_of;
	s.dev_coef = 0;
	s.es_index = -1;
	s.coef_bytes_fmt = clone_of >= 0 ? b->slots[(size_t)clone_of].coef_bytes_fmt : 0;
	s.coef_bytes = cbytes;
	s.path = 0;
	if (clone_of < 0) {
		if (b->stage_used + cbytes > b->stage_cap) {
			if (!lazy_stage)
				return set_err(MIJ_E_NOMEM, "staging arena exhausted");
			s.stage_off = MIJ_NO_STAGE;
		} else {
			s.stage_off = b->stage_used;
			if (!lazy_stage && clear)
				memset(b->stage + s.stage_off, 0, cbytes);
			b->stage_used += cbytes;
		}
	} else {
		s.stage_off = b->slots[(size_t)clone_of].stage_off;
	}
	s.coef_base = b->coef_used;
	fill_dev_image(s, b->out_used);
	layout_coef(s); /* host-staged slots: int16 until upload decides; clones: their source's format */
	b->coef_used += cbytes;
	b->out_used += obytes;
	b->slots.push_back(s);
	b->uploaded = b->launched = false;
	return (int)b->slots.size() - 1;
}

extern "C" int mij_batch_add(mij_batch *b, const mij_image_desc *d)
{
	if (!b)
		return set_err(MIJ_E_ARG, "batch is NULL");
	int rc = check_desc(d);
	if (rc != MIJ_OK)
		return rc;
	return add_common(b, d, -1);
}

extern "C" int mij_batch_add_uncleared(mij_batch *b, const mij_image_desc *d)
{
	if (!b)
		return set_err(MIJ_E_ARG, "batch is NULL");
	int rc = check_desc(d);
	if (rc != MIJ_OK)
		return rc;
	return add_common(b, d, -1, false, false);
}

extern "C" int mij_batch_add_clone(mij_batch *b, int src_slot)
{
	if (!b || src_slot < 0 || src_slot >= (int)b->slots.size())
		return set_err(MIJ_E_ARG, "bad source slot");
	int root = b->slots[(size_t)src_slot].clone_of >= 0 ? b->slots[(size_t)src_slot].clone_of : src_slot;
	mij_image_desc d = b->slots[(size_t)root].desc;
	return add_common(b, &d, root);
}

extern "C" int16_t *mij_batch_coef(mij_batch *b, int slot, int comp)
{
	if (!b || slot < 0 || slot >= (int)b->slots.size()) {
		set_err(MIJ_E_ARG, "bad slot");
		return nullptr;
	}
	const Slot &s = b->slots[(size_t)slot];
	if (comp < 0 || comp >= s.desc.ncomp || s.clone_of >= 0 || !b->stage || s.stage_off == MIJ_NO_STAGE) {
		set_err(MIJ_E_ARG, "bad component, or slot has no staging of its own");
		return nullptr;
	}
	size_t off = s.stage_off;
	for (int c = 0; c < comp; ++c)
		off += mij_plane_elems((uint32_t)(s.desc.comp[c].bw * s.desc.comp[c].bh)) * sizeof(int16_t);
	return reinterpret_cast<int16_t *>(b->stage + off);
}

extern "C" uint8_t *mij_batch_stage_region(mij_batch *b, int slot, size_t *bytes)
{
	if (!b || slot < 0 || slot >= (int)b->slots.size()) {
		set_err(MIJ_E_ARG, "bad slot");
		return nullptr;
	}
	const Slot &s = b->slots[(size_t)slot];
	if (s.clone_of >= 0 || !b->stage || s.stage_off == MIJ_NO_STAGE) {
		set_err(MIJ_E_ARG, "slot has no staging of its own");
		return nullptr;
	}
	if (bytes)
		*bytes = s.coef_bytes;
	return b->stage + s.stage_off;
}

extern "C" int mij_batch_coef_format(const mij_batch *b) { return b ? b->coef_fmt : MIJ_COEF_COMPACT; }
extern "C" uint32_t mij_batch_slot_flags(const mij_batch *b, int slot)
{
	if (!b || slot < 0 || slot >= (int)b->slots.size())
		return 0;
	return b->slots[(size_t)slot].desc.flags;
}

extern "C" int mij_batch_set_flags(mij_batch *b, int slot, uint32_t flags)
{
	if (!b || slot < 0 || slot >= (int)b->slots.size())
		return set_err(MIJ_E_ARG, "bad slot");
	b->slots[(size_t)slot].desc.flags = flags;
	b->slots[(size_t)slot].dev.flags = (int32_t)flags | (b->slots[(size_t)slot].coef_bytes_fmt ? MIJ_DEV_COEF_BYTES : 0);
	b->uploaded = b->launched = false;
	return MIJ_OK;
}

extern "C" int mij_batch_set_color(mij_batch *b, int slot, int color)
{
	if (!b || slot < 0 || slot >= (int)b->slots.size())
		return set_err(MIJ_E_ARG, "bad slot");
	Slot &s = b->slots[(size_t)slot];
	mij_image_desc d = s.desc;
	d.color = color;
	int rc = check_desc(&d);
	if (rc != MIJ_OK)
		return rc;
	s.desc.color = color;
	s.dev.color = color;
	b->uploaded = b->launched = false;
	return MIJ_OK;
}

extern "C" int mij_batch_image_count(const mij_batch *b) { return b ? (int)b->slots.size() : 0; }
extern "C" int mij_batch_slot_path(const mij_batch *b, int slot)
{
	if (!b || slot < 0 || slot >= (int)b->slots.size())
		return 0;
	return b->slots[(size_t)slot].path;
}
extern "C" int mij_batch_force_generic(mij_batch *b, int on)
{
	if (!b)
		return set_err(MIJ_E_ARG, "batch is NULL");
	b->force_generic = on < 0 ? 0 : (on > 2 ? 2 : on);
	b->uploaded = b->launched = false;
	return MIJ_OK;
}

/* can the fused h2v2 kernel take this image? */
static bool fused420_ok(const mij_batch *b, const mij_image_desc &d)
{
	if (b->force_generic || (d.flags & MIJ_FLAG_SKIP))
		return false;
	if (d.ncomp != 3 || d.color != MIJ_COLOR_YCBCR || (d.n_out != 3 && d.n_out != 4))
		return false;
	if (d.comp[0].h != 2 || d.comp[0].v != 2)
		return false;
	for (int c = 1; c < 3; ++c)
		if (d.comp[c].h != 1 || d.comp[c].v != 1)
			return false;
	return true; /* any width: a row of MCUs beyond the LDS of a CU goes in column segments (fused420_kind) */
}

/* single-component images: IDCT straight into the pixel buffer */
/* h1v2 (4:4:0): luma 1x2, chroma 1x1 -- the band kernel with H2 = false, 304 * mcu_x bytes of LDS */
static size_t fused440_lds(const mij_image_desc &d) { return (size_t)d.mcu_x * (16 * 8 + 2 * 8 * 8 + 2 * 8 + 4 * 8); }
static bool fused440_ok(const mij_batch *b, const mij_image_desc &d)
{
	if (b->force_generic || (d.flags & MIJ_FLAG_SKIP))
		return false;
	if (d.ncomp != 3 || d.color != MIJ_COLOR_YCBCR || (d.n_out != 3 && d.n_out != 4))
		return false;
	if (d.comp[0].h != 1 || d.comp[0].v != 2)
		return false;
	for (int c = 1; c < 3; ++c)
		if (d.comp[c].h != 1 || d.comp[c].v != 1)
			return false;
	return true; /* any width, see fused420_ok */
}

static bool fused_grey_ok(const mij_batch *b, const mij_image_desc &d)
{
	if (b->force_generic || (d.flags & MIJ_FLAG_SKIP))
		return false;
	/* one component -- or the luma of a YCbCr file asked for as grey (req_comp 1 / 2: the reference resamples only component 0
	 * then, codec/jpeg.c:2246,:2380-2430), when the luma plane has the picture's own resolution: the chroma planes are not even
	 * transformed */
	const bool luma_only = d.ncomp == 3 && d.comp[0].h == d.h_max && d.comp[0].v == d.v_max && d.n_out < 3;
	return (d.ncomp == 1 || luma_only) && d.color == MIJ_COLOR_GREY && d.n_out >= 1 && d.n_out <= 4 && (uint64_t)d.width * d.height * d.n_out < 0xfffffff0ull;
}

/* can the fused h2v1 kernel take this image? */
static bool fused422_ok(const mij_batch *b, const mij_image_desc &d)
{
	if (b->force_generic || (d.flags & MIJ_FLAG_SKIP))
		return false;
	if (d.ncomp != 3 || d.color != MIJ_COLOR_YCBCR || (d.n_out != 3 && d.n_out != 4))
		return false;
	if (d.comp[0].h != 2 || d.comp[0].v != 1)
		return false;
	for (int c = 1; c < 3; ++c)
		if (d.comp[c].h != 1 || d.comp[c].v != 1)
			return false;
	return (size_t)d.mcu_x * 256 + 16 <= (size_t)b->ctx->max_dyn_lds;
}

/* can the register-resident 4:4:4 kernel take this image? */
/* Which specialised pass 2 (RS_*, mij_kernels.h) serves an image of the two-pass path, or -1 for the run-time-general
 * k_resample_color: component 0 (and 3) at full resolution, components 1 and 2 sharing factors that divide, W % 4 == 0,
 * three or four output channels.  *ycc: YCbCr colour (stbi__YCbCr_to_RGB_row) as opposed to RGB-tagged / CMYK / YCCK. */
static int resample_fast_kind(const mij_batch *b, const mij_image_desc &d, int *ycc)
{
	if (b->force_generic >= 2 || (d.n_out != 3 && d.n_out != 4) || (d.width & 3) || d.ncomp < 3)
		return -1;
	const bool four = d.color == MIJ_COLOR_CMYK || d.color == MIJ_COLOR_YCCK;
	if (d.color != MIJ_COLOR_YCBCR && d.color != MIJ_COLOR_YCBCRA && d.color != MIJ_COLOR_RGB && !four)
		return -1;
	if ((four && d.ncomp != 4) || d.comp[0].h != d.h_max || d.comp[0].v != d.v_max)
		return -1;
	if (four && (d.comp[3].h != d.h_max || d.comp[3].v != d.v_max))
		return -1;
	if (d.comp[1].h != d.comp[2].h || d.comp[1].v != d.comp[2].v || d.h_max % d.comp[1].h || d.v_max % d.comp[1].v)
		return -1;
	const int hs = d.h_max / d.comp[1].h, vs = d.v_max / d.comp[1].v;
	*ycc = (d.color == MIJ_COLOR_YCBCR || d.color == MIJ_COLOR_YCBCRA) ? 1 : 0;
	if (hs == 1)
		return vs == 2 ? RS_V2 : RS_ROW1;
	if (hs == 2)
		return vs == 1 ? RS_H2 : (vs == 2 ? RS_HV2 : RS_GEN2);
	return hs == 4 ? RS_GEN4 : -1;
}

static bool all_1x1(const mij_image_desc &d)
{
	for (int c = 0; c < d.ncomp; ++c)
		if (d.comp[c].h != 1 || d.comp[c].v != 1)
			return false;
	return (uint64_t)d.width * d.height * d.n_out < 0xfffffff0ull;
}
static bool fused444_ok(const mij_batch *b, const mij_image_desc &d)
{
	if (b->force_generic || (d.flags & MIJ_FLAG_SKIP) || (d.n_out != 3 && d.n_out != 4))
		return false;
	/* three-component YCbCr, or four components whose transform is YCbCr with the fourth ignored (codec/jpeg.c:2367-2370): the kernel only touches components 0-2 */
	if (!((d.ncomp == 3 && d.color == MIJ_COLOR_YCBCR) || (d.ncomp == 4 && d.color == MIJ_COLOR_YCBCRA)))
		return false;
	return all_1x1(d);
}
/* k_fused1x1c: RGB-tagged (codec/jpeg.c:2325-2335), Adobe CMYK (:2343-2354), YCCK (:2355-2366) with every component at 1x1 */
static bool fused1x1c_ok(const mij_batch *b, const mij_image_desc &d)
{
	if (b->force_generic || (d.flags & MIJ_FLAG_SKIP) || (d.n_out != 3 && d.n_out != 4))
		return false;
	if (!((d.ncomp == 3 && d.color == MIJ_COLOR_RGB) || (d.ncomp == 4 && (d.color == MIJ_COLOR_CMYK || d.color == MIJ_COLOR_YCCK))))
		return false;
	return all_1x1(d);
}

/* Small tables (image descriptors, work lists, Huffman tables) go to the device by a copy KERNEL reading the
 * pinned host buffer, not by hipMemcpyAsync: with several batches in flight the runtime's copy path made the host
 * wait behind other streams' large transfers (measured: 8 ms for a 100 KB copy, one call in four).  Sizes are
 * multiples of 4; the buffers come from hipHostMalloc, which maps them for the device. */
__global__ __launch_bounds__(256) void k_copy_words(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint32_t n)
{
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
		dst[i] = src[i];
}

__global__ __launch_bounds__(256) void k_copy_words16(uint4 *__restrict__ dst, const uint4 *__restrict__ src, uint32_t n)
{
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
		dst[i] = src[i];
}

static hipError_t copy_table(void *dst, const void *src_pinned, size_t bytes, hipStream_t st)
{
	void *dsrc = nullptr;
	if (bytes == 0)
		return hipSuccess;
	if ((bytes & 3u) || bytes > (1u << 30) || hipHostGetDevicePointer(&dsrc, const_cast<void *>(src_pinned), 0) != hipSuccess || !dsrc) {
		(void)hipGetLastError();
		return hipMemcpyAsync(dst, src_pinned, bytes, hipMemcpyHostToDevice, st);
	}
	if (!((uintptr_t)dst & 15u) && !((uintptr_t)dsrc & 15u) && !(bytes & 15u) && bytes >= 65536) { /* the entropy streams: tens of MB */
		const uint32_t n = (uint32_t)(bytes / 16);
		const unsigned grid = (n + 1023u) / 1024u > 512u ? 512u : (n + 1023u) / 1024u;
		hipLaunchKernelGGL(k_copy_words16, dim3(grid ? grid : 1u), dim3(256), 0, st, static_cast<uint4 *>(dst), static_cast<const uint4 *>(dsrc), n);
		return hipGetLastError();
	}
	const uint32_t n = (uint32_t)(bytes / 4);
	const unsigned grid = (n + 1023u) / 1024u > 1024u ? 1024u : (n + 1023u) / 1024u;
	hipLaunchKernelGGL(k_copy_words, dim3(grid ? grid : 1u), dim3(256), 0, st, static_cast<uint32_t *>(dst), static_cast<const uint32_t *>(dsrc), n);
	return hipGetLastError();
}

/* the other direction for small results (verdict words): the kernel writes the pinned host buffer through its device mapping */
static hipError_t copy_table_to_host(void *dst_pinned, const void *src_dev, size_t bytes, hipStream_t st)
{
	void *ddst = nullptr;
	if (bytes == 0)
		return hipSuccess;
	if ((bytes & 3u) || hipHostGetDevicePointer(&ddst, dst_pinned, 0) != hipSuccess || !ddst) {
		(void)hipGetLastError();
		return hipMemcpyAsync(dst_pinned, src_dev, bytes, hipMemcpyDeviceToHost, st);
	}
	const uint32_t n = (uint32_t)(bytes / 4);
	const unsigned grid = (n + 1023u) / 1024u > 1024u ? 1024u : (n + 1023u) / 1024u;
	hipLaunchKernelGGL(k_copy_words, dim3(grid ? grid : 1u), dim3(256), 0, st, static_cast<uint32_t *>(ddst), static_cast<const uint32_t *>(src_dev), n);
	return hipGetLastError();
}

template <typename T>
static int grow_pair(T *&h, T *&d, size_t &cap, size_t need)
{
	if (need <= cap)
		return MIJ_OK;
	size_t ncap = need + need / 2 + 64;
	if (h)
		(void)hipHostFree(h);
	if (d)
		(void)hipFree(d);
	h = nullptr;
	d = nullptr;
	cap = 0;
	HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h), sizeof(T) * ncap, hipHostMallocDefault));
	HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), sizeof(T) * ncap));
	cap = ncap;
	return MIJ_OK;
}

/* experiment knobs (environment, read once): MIJ_SEG_MIN = MCU columns beyond which a picture goes in column segments although its row
 * fits the LDS; MIJ_SEG_COLS = widest segment in MCU columns (0: whatever fills the LDS) */
static int seg_min_cols()
{
	static int v = -1;
	if (v < 0) {
		const char *e = getenv("MIJ_SEG_MIN");
		v = e ? atoi(e) : 1 << 30;
	}
	return v;
}
static int seg_max_cols()
{
	static int v = -1;
	if (v < 0) {
		const char *e = getenv("MIJ_SEG_COLS");
		v = e ? atoi(e) : 0;
	}
	return v;
}
static size_t fused420_lds(const mij_image_desc &d) { return (size_t)d.mcu_x * (16 * 16 + 2 * 8 * 8 + 2 * 16 + 4 * 8); }
/* fewer than three workgroups of the band kernel fit a CU's LDS: the eight-wave form (k_fused420w) */
/* which form of the band kernel a 4:2:0 picture takes (-1: none): by the workgroups of its width that fit a CU's LDS, and for narrow
 * pictures by how many waves a row of MCUs keeps busy (mij_kernels.h, k_fused420w / x / s / t) */
static int fused420_kind(const mij_batch *b, const mij_image_desc &d)
{
	if (!fused420_ok(b, d))
		return -1;
	const size_t lds = fused420_lds(d), cap = (size_t)b->ctx->max_dyn_lds;
	if (lds > cap || d.mcu_x > seg_min_cols())
		return MK_420C;
	if (2 * lds > cap)
		return MK_420X;
	if (3 * lds > cap)
		return MK_420W;
	return d.mcu_x <= 24 ? MK_420T : (d.mcu_x <= 56 ? MK_420S : MK_420);
}
#ifndef MIJ_422T_MAX /* MCU columns up to which k_fused422 runs with one / two waves (A/B: 0 switches a form off) */
#define MIJ_422T_MAX 24
#endif
#ifndef MIJ_422S_MAX
#define MIJ_422S_MAX 56
#endif
static int band_threads(int kind)
{
	switch (kind) {
	case MK_420X: case MK_422X: return MIJ_F420X_NT;
	case MK_420C: case MK_440C: return MIJ_F420C_NT;
	case MK_420W: case MK_440W: case MK_422W: return MIJ_F420W_NT;
	case MK_420S: case MK_422S: return MIJ_F420S_NT;
	case MK_420T: case MK_422T: return MIJ_F420T_NT;
	default: return MIJ_F420_NT;
	}
}
static bool fused440_wide(const mij_batch *b, const mij_image_desc &d) { return fused440_ok(b, d) && 3 * fused440_lds(d) > (size_t)b->ctx->max_dyn_lds; }
static int fused440_kind(const mij_batch *b, const mij_image_desc &d)
{
	return !fused440_ok(b, d) ? -1 : (fused440_lds(d) > (size_t)b->ctx->max_dyn_lds ? MK_440C : (fused440_wide(b, d) ? MK_440W : MK_440));
}
/* Column segments of a picture too wide for one workgroup's LDS (bytes_per_col per MCU column): the fewest segments, of equal width, of
 * which two fit a CU with their two halo columns each.  Returns the segment count; segment k spans MCU columns [mcu_x * k / n, mcu_x * (k + 1) / n). */
static int band_segments(const mij_batch *b, int mcu_x, size_t bytes_per_col)
{
	const int most = (int)((size_t)b->ctx->max_dyn_lds / bytes_per_col) - 2; /* what fits at all */
	int fit = (int)((size_t)b->ctx->max_dyn_lds / (2 * bytes_per_col)) - 2;   /* two workgroups per CU (mij_kernels.h, k_fused420c) */
	if (seg_max_cols() > 0)
		fit = seg_max_cols() < most ? seg_max_cols() : most;
	if (fit < 1)
		return mcu_x; /* cannot happen with 160 KiB of LDS: one column per segment */
	return (mcu_x + fit - 1) / fit;
}
static size_t band_segment_lds(int mcu_x, int nseg, size_t bytes_per_col)
{
	int widest = 0;
	for (int k = 0; k < nseg; ++k) {
		const int w = (int)((long)mcu_x * (k + 1) / nseg - (long)mcu_x * k / nseg);
		widest = w > widest ? w : widest;
	}
	return (size_t)(widest + 2) * bytes_per_col;
}

extern "C" int mij_batch_upload(mij_batch *b)
{
	if (!b)
		return set_err(MIJ_E_ARG, "batch is NULL");
	HIP_TRY(hipSetDevice(b->ctx->device));
	const size_t n = b->slots.size();
	if (n == 0)
		return set_err(MIJ_E_STATE, "batch is empty");

	/* ---- format of the planes in HBM.  Slots the GPU entropy stage wrote keep theirs; host-staged slots get the
	 * batch's format (compact by default: uploaded as int16 into the scratch, packed by k_pack_c8); clones follow
	 * their source (which precedes them). */
	bool need_pack = false;
	for (size_t i = 0; i < n; ++i) {
		Slot &s = b->slots[i];
		if (s.clone_of >= 0)
			s.coef_bytes_fmt = b->slots[(size_t)s.clone_of].coef_bytes_fmt;
		else if (!s.dev_coef) /* planes the host staged compact keep that format whatever the batch's default */
			s.coef_bytes_fmt = ((b->coef_fmt || (s.desc.flags & (MIJ_FLAG_STAGED_COMPACT | MIJ_FLAG_L1_ON_DEVICE))) && !(s.desc.flags & MIJ_FLAG_SKIP)) ? 1 : 0;
		layout_coef(s);
		if (s.clone_of < 0 && !s.dev_coef && s.coef_bytes_fmt && !(s.desc.flags & MIJ_FLAG_STAGED_COMPACT))
			need_pack = true;
	}
	if (need_pack && !b->d_up16) {
		hipError_t e = hipMalloc(reinterpret_cast<void **>(&b->d_up16), b->stage_cap ? b->stage_cap : 16);
		if (e != hipSuccess)
			return set_err(e == hipErrorOutOfMemory ? MIJ_E_NOMEM : MIJ_E_HIP, "upload scratch: %s", hipGetErrorString(e));
	}

	/* ---- progressive files whose L1 bound the host left to the device (MIJ_FLAG_L1_ON_DEVICE): their planes go up and are packed NOW, the
	 * pack kernel takes every block's L1 on the way, the maxima come back, and MIJ_FLAG_WIDE_IDCT is set before the launch plan below sorts
	 * the images by it.  Afterwards these slots hold finished compact planes in HBM (dev_coef: later uploads leave them alone). */
	{
		std::vector<Work4> pre;
		for (size_t i = 0; i < n; ++i) {
			const Slot &s = b->slots[i];
			if (s.clone_of < 0 && !s.dev_coef && (s.desc.flags & MIJ_FLAG_L1_ON_DEVICE) && !(s.desc.flags & MIJ_FLAG_SKIP))
				for (int c = 0; c < s.desc.ncomp; ++c)
					for (uint32_t f = 0, nb = (uint32_t)comp_tiles(s.desc.comp[c]) * 64u; f < nb; f += 256)
						pre.push_back(Work4{(uint32_t)i, (uint32_t)c, f, 0u});
		}
		if (!pre.empty()) {
			if (!b->d_l1max) {
				HIP_TRY(hipMalloc(reinterpret_cast<void **>(&b->d_l1max), sizeof(uint32_t) * (size_t)b->max_images));
				HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->h_l1max), sizeof(uint32_t) * (size_t)b->max_images, hipHostMallocDefault));
			}
			if (pre.size() > b->work_cap)
				HIP_TRY(hipStreamSynchronize(b->stream));
			int rc0;
			if ((rc0 = grow_pair(b->h_work, b->d_work, b->work_cap, pre.size())) != MIJ_OK)
				return rc0;
			memcpy(b->h_work, pre.data(), pre.size() * sizeof(Work4));
			for (size_t i = 0; i < n; ++i) {
				Slot &s = b->slots[i];
				s.dev.src16_off = s.stage_off == MIJ_NO_STAGE ? 0 : s.stage_off;
				b->h_imgs[i] = s.dev;
				if (s.clone_of < 0 && !s.dev_coef && (s.desc.flags & MIJ_FLAG_L1_ON_DEVICE))
					b->h_imgs[i].flags |= MIJ_DEV_L1_MAX;
			}
			HIP_TRY(copy_table(b->d_imgs, b->h_imgs, sizeof(DevImage) * n, b->stream));
			HIP_TRY(copy_table(b->d_work, b->h_work, sizeof(Work4) * pre.size(), b->stream));
			HIP_TRY(hipMemsetAsync(b->d_l1max, 0, sizeof(uint32_t) * n, b->stream));
			for (size_t i = 0; i < n; ++i) {
				const Slot &s = b->slots[i];
				if (s.clone_of < 0 && !s.dev_coef && (s.desc.flags & MIJ_FLAG_L1_ON_DEVICE) && !(s.desc.flags & MIJ_FLAG_SKIP)) {
					size_t bytes16 = 0;
					for (int c = 0; c < s.desc.ncomp; ++c)
						bytes16 += comp_tiles(s.desc.comp[c]) << 13;
					HIP_TRY(hipMemcpyAsync(b->d_up16 + s.stage_off, b->stage + s.stage_off, bytes16, hipMemcpyHostToDevice, b->stream));
				}
			}
			hipLaunchKernelGGL(k_pack_c8, dim3((unsigned)pre.size()), dim3(256), 0, b->stream, b->d_imgs, reinterpret_cast<const WorkIdct *>(b->d_work), b->d_up16, b->d_coef, b->d_l1max);
			HIP_TRY(hipGetLastError());
			HIP_TRY(copy_table_to_host(b->h_l1max, b->d_l1max, sizeof(uint32_t) * n, b->stream));
			HIP_TRY(hipStreamSynchronize(b->stream));
			for (size_t i = 0; i < n; ++i) {
				Slot &s = b->slots[i];
				if (s.clone_of < 0 && !s.dev_coef && (s.desc.flags & MIJ_FLAG_L1_ON_DEVICE) && !(s.desc.flags & MIJ_FLAG_SKIP)) {
					s.desc.flags &= ~(uint32_t)MIJ_FLAG_L1_ON_DEVICE;
					if (b->h_l1max[i] > (uint32_t)MIJ_BLOCK_L1_LIMIT)
						s.desc.flags |= MIJ_FLAG_WIDE_IDCT;
					s.dev.flags = (int32_t)s.desc.flags | MIJ_DEV_COEF_BYTES;
					s.dev_coef = 1;
				}
			}
			for (size_t i = 0; i < n; ++i) { /* clones take their source's verdict */
				Slot &s = b->slots[i];
				if (s.clone_of >= 0 && (s.desc.flags & MIJ_FLAG_L1_ON_DEVICE)) {
					s.desc.flags = b->slots[(size_t)s.clone_of].desc.flags;
					s.dev.flags = (int32_t)s.desc.flags | MIJ_DEV_COEF_BYTES;
				}
			}
		}
	}

	/* ---- plan: one work list per (kernel family, n_out, wide IDCT, plane format) */
	std::vector<Work4> lists[MK_KINDS][2][2][2];
	size_t lds_need[MK_KINDS][2][2][2];
	memset(lds_need, 0, sizeof(lds_need));
	std::vector<Work4> pack;
	size_t planes_need = 0, planes_off = 0;
	const int cu = b->ctx->prop.multiProcessorCount > 0 ? b->ctx->prop.multiProcessorCount : 256;
	/* Automatic band count per image.  The grid runs in "rounds" of (CUs x workgroups per CU by LDS)
	 * co-resident workgroups; the last round of a launch is only as full as the remainder, and every
	 * band re-does two chroma block rows of IDCT as halo.  Pick the bands-per-image (1..16) that
	 * minimises  rounds x (1 + halo share)  per unit of work; measured on MI355X: 1024 x 1080p ->
	 * 6 bands (6144 workgroups = 8.0 rounds of 768) beats 4 (5.33 rounds) by ~1.5 %. */
#ifndef MIJ_BAND_CAP
#define MIJ_BAND_CAP 64
#endif
	auto auto_bands = [&](int (*kind_of)(const mij_batch *, const mij_image_desc &), int kind, size_t (*lds_of)(const mij_image_desc &)) -> int {
		const int nt = band_threads(kind);
		size_t n_fused = 0, mcu_rows_sum = 0, lds_max = 0;
		for (size_t i = 0; i < n; ++i)
			if (kind_of(b, b->slots[i].desc) == kind) {
				const mij_image_desc &d = b->slots[i].desc;
				++n_fused;
				mcu_rows_sum += (size_t)d.mcu_y;
				if (lds_of(d) > lds_max)
					lds_max = lds_of(d);
			}
		int nb_best = 1;
		if (n_fused) {
			size_t per_cu = lds_max ? (size_t)b->ctx->max_dyn_lds / lds_max : 1;
			const size_t by_waves = 4 * MIJ_F420_WAVES / ((size_t)nt / 64); /* waves per SIMD by registers x four SIMDs */
			per_cu = per_cu < 1 ? 1 : (per_cu > by_waves ? by_waves : per_cu);
			const size_t slots = (size_t)cu * per_cu;
			const double avg_rows = (double)mcu_rows_sum / (double)n_fused;
			double best = 1e30;
			/* up to 16 bands per picture; up to MIJ_BAND_CAP for 4:2:0 batches so small that sixteen bands each leave workgroup slots
			 * empty (a lone picture from stbi_load, a handful): 16 x 1080p 0.078 -> 0.058 ms.  Not for 4:4:0, whose halo is a larger share
			 * of a band (0.069 -> 0.093 ms), and not once the slots are full (36 x 5120 x 2880: 0.70 -> 0.73 ms with the higher cap). */
			const int cap = (kind != MK_440 && kind != MK_440W && kind != MK_440C && n_fused * 16 <= slots) ? MIJ_BAND_CAP : 16;
			for (int nb = 1; nb <= cap && nb <= (int)avg_rows; ++nb) {
				const size_t wgs = n_fused * (size_t)nb;
				const size_t rounds = (wgs + slots - 1) / slots;
				/* every inner band edge re-transforms two chroma block rows (4 of an MCU row's 6 blocks' worth), the IDCT being ~45 %
				 * of the work; a launch also pays about 0.3 band lengths of ramp-up and tail whatever its shape -- without that term
				 * the model took 3 bands for 1024 x 1080p where 6 measure 1.5 % faster, and 3 for 256 images where 12 measure 4 % faster
				 * (interleaved runs, profiles/r02z_band_count.txt) */
				const double halo = 1.0 + 0.45 * 4.0 * (nb - 1) / (6.0 * avg_rows);
				const double cost = ((double)rounds + 0.3) * (avg_rows / nb) * halo; /* time ~ (rounds + ramp) x band length */
				if (cost < best * 0.999) {
					best = cost;
					nb_best = nb;
				}
			}
		}
		return nb_best;
	};
	int auto_nb[MK_KINDS];
	for (int k = 0; k < MK_KINDS; ++k)
		auto_nb[k] = 1;
	for (int k : {MK_420, MK_420W, MK_420X, MK_420S, MK_420T, MK_420C})
		auto_nb[k] = auto_bands(fused420_kind, k, fused420_lds);
	for (int k : {MK_440, MK_440W, MK_440C})
		auto_nb[k] = auto_bands(fused440_kind, k, fused440_lds);

	for (size_t i = 0; i < n; ++i) {
		Slot &s = b->slots[i];
		const mij_image_desc &d = s.desc;
		const int wide = (d.flags & MIJ_FLAG_WIDE_IDCT) ? 1 : 0, b8 = s.coef_bytes_fmt ? 1 : 0, o4 = d.n_out == 4 ? 1 : 0;
		if (d.flags & MIJ_FLAG_SKIP) { /* rejected by the host stage after it got a slot */
			s.path = 0;
			continue;
		}
		if (s.clone_of < 0 && !s.dev_coef && b8 && !(d.flags & MIJ_FLAG_STAGED_COMPACT)) /* int16 planes in the scratch -> compact planes */
			for (int c = 0; c < d.ncomp; ++c)
				for (uint32_t f = 0, nb = (uint32_t)comp_tiles(d.comp[c]) * 64u; f < nb; f += 256)
					pack.push_back(Work4{(uint32_t)i, (uint32_t)c, f, 0u});
		auto per_blocks = [&](std::vector<Work4> &L, int comp) {
			const uint32_t nblk = (uint32_t)(d.comp[comp].bw * d.comp[comp].bh);
			for (uint32_t f = 0; f < nblk; f += 256)
				L.push_back(Work4{(uint32_t)i, (uint32_t)comp, f, 0u});
		};
		if (fused420_ok(b, d)) {
			s.path = 1;
			/* split mcu_y into nb equal-ish bands */
			const int mk = fused420_kind(b, d);
			int nb = b->band_rows > 0 ? (d.mcu_y + b->band_rows - 1) / b->band_rows : auto_nb[mk];
			if (nb > d.mcu_y)
				nb = d.mcu_y;
			if (nb < 1)
				nb = 1;
			const int nseg = mk == MK_420C ? band_segments(b, d.mcu_x, fused420_lds(d) / (size_t)d.mcu_x) : 1;
			for (int k = 0; k < nb; ++k)
				for (int g = 0; g < nseg; ++g) /* cols: only the column-segmented form reads it */
					lists[mk][o4][wide][b8].push_back(Work4{(uint32_t)i, (uint32_t)((long)d.mcu_y * k / nb), (uint32_t)((long)d.mcu_y * (k + 1) / nb),
																		 (uint32_t)((long)d.mcu_x * g / nseg) | (uint32_t)((long)d.mcu_x * (g + 1) / nseg) << 16});
			size_t &l = lds_need[mk][o4][wide][b8];
			const size_t need = mk == MK_420C ? band_segment_lds(d.mcu_x, nseg, fused420_lds(d) / (size_t)d.mcu_x) : fused420_lds(d);
			l = need > l ? need : l;
		} else if (fused_grey_ok(b, d)) {
			s.path = 5;
			per_blocks(lists[MK_GREY][0][wide][b8], 0);
		} else if (fused422_ok(b, d)) {
			s.path = 4;
			const size_t lds422 = (size_t)d.mcu_x * 256 + 16, cap422 = (size_t)b->ctx->max_dyn_lds;
			const int mk422 = 2 * lds422 > cap422 ? MK_422X : (3 * lds422 > cap422 ? MK_422W : (d.mcu_x <= MIJ_422T_MAX ? MK_422T : (d.mcu_x <= MIJ_422S_MAX ? MK_422S : MK_422)));
			size_t &l = lds_need[mk422][o4][wide][b8];
			l = lds422 > l ? lds422 : l;
			/* no halo: bands of about eight MCU rows keep the grid deep without making workgroups short */
			const int nb = (d.mcu_y + 7) / 8;
			for (int k = 0; k < nb; ++k)
				lists[mk422][o4][wide][b8].push_back(Work4{(uint32_t)i, (uint32_t)((long)d.mcu_y * k / nb), (uint32_t)((long)d.mcu_y * (k + 1) / nb), 0u});
		} else if (fused440_ok(b, d)) {
			s.path = 6;
			const int mk = fused440_kind(b, d);
			const int nseg = mk == MK_440C ? band_segments(b, d.mcu_x, fused440_lds(d) / (size_t)d.mcu_x) : 1;
			size_t &l = lds_need[mk][o4][wide][b8];
			const size_t need = mk == MK_440C ? band_segment_lds(d.mcu_x, nseg, fused440_lds(d) / (size_t)d.mcu_x) : fused440_lds(d);
			l = need > l ? need : l;
			/* band count by rounds of co-resident workgroups, as for 4:2:0 */
			int nb = b->band_rows > 0 ? (d.mcu_y + b->band_rows - 1) / b->band_rows : auto_nb[mk];
			nb = nb > d.mcu_y ? d.mcu_y : (nb < 1 ? 1 : nb);
			for (int k = 0; k < nb; ++k)
				for (int g = 0; g < nseg; ++g)
					lists[mk][o4][wide][b8].push_back(Work4{(uint32_t)i, (uint32_t)((long)d.mcu_y * k / nb), (uint32_t)((long)d.mcu_y * (k + 1) / nb),
																		 (uint32_t)((long)d.mcu_x * g / nseg) | (uint32_t)((long)d.mcu_x * (g + 1) / nseg) << 16});
		} else if (fused444_ok(b, d)) {
			s.path = 3;
			per_blocks(lists[MK_444][o4][wide][b8], 0);
		} else if (fused1x1c_ok(b, d)) {
			s.path = 7;
			per_blocks(lists[MK_1X1C][o4][wide][b8], 0);
		} else {
			s.path = 2;
			int ycc = 0;
			const int rk = resample_fast_kind(b, d, &ycc);
			std::vector<Work4> &R = rk < 0 ? lists[MK_RESAMPLE][0][0][0] : lists[MK_RS_FAST + rk][o4][ycc][0];
			/* the vs == 2 forms of k_resample_fast take item r as output rows r-1 .. r+2 (row pairs around a chroma row) */
			const uint32_t r_end = (uint32_t)d.height + ((rk == RS_V2 || rk == RS_HV2) ? 2u : 0u);
			for (uint32_t r = 0; r < r_end; r += MIJ_RESAMPLE_ROWS)
				R.push_back(Work4{(uint32_t)i, 0u, r, 0u});
			for (int c = 0; c < d.ncomp; ++c)
				per_blocks(lists[MK_PLANES][0][wide][b8], c);
			/* rebase this image's sample planes into the scratch arena */
			size_t rel0 = 0;
			for (int c = 0; c < d.ncomp; ++c) {
				s.dev.comp[c].plane_off = planes_off + rel0;
				rel0 += align_up((size_t)d.comp[c].bw * 8 * d.comp[c].bh * 8, 256);
			}
			planes_off += rel0;
			planes_need = planes_off;
		}
		s.dev.src16_off = s.stage_off == MIJ_NO_STAGE ? 0 : s.stage_off;
	}

	/* ---- scratch planes for the two-pass path */
	if (planes_need > b->planes_cap) {
		HIP_TRY(hipStreamSynchronize(b->stream));
		if (b->d_planes)
			(void)hipFree(b->d_planes);
		b->d_planes = nullptr;
		b->planes_cap = 0;
		hipError_t e = hipMalloc(reinterpret_cast<void **>(&b->d_planes), planes_need);
		if (e != hipSuccess)
			return set_err(e == hipErrorOutOfMemory ? MIJ_E_NOMEM : MIJ_E_HIP, "scratch planes: %s", hipGetErrorString(e));
		b->planes_cap = planes_need;
	}

	/* ---- work lists: the pack list first, then one range per launch */
	size_t total = pack.size();
	for (int k = 0; k < MK_KINDS; ++k)
		for (int v = 0; v < 8; ++v)
			total += lists[k][v >> 2][(v >> 1) & 1][v & 1].size();
	if (total > b->work_cap)
		HIP_TRY(hipStreamSynchronize(b->stream));
	int rc;
	if ((rc = grow_pair(b->h_work, b->d_work, b->work_cap, total)) != MIJ_OK)
		return rc;
	b->launches.clear();
	size_t pos = 0;
	if (!pack.empty())
		memcpy(b->h_work, pack.data(), pack.size() * sizeof(Work4));
	pos = pack.size();
	for (int k = 0; k < MK_KINDS; ++k)
		for (int v = 0; v < 8; ++v) {
			const std::vector<Work4> &L = lists[k][v >> 2][(v >> 1) & 1][v & 1];
			if (L.empty())
				continue;
			memcpy(b->h_work + pos, L.data(), L.size() * sizeof(Work4));
			mij_batch::Launch q;
			q.kind = k;
			q.nout = (v >> 2) ? 4 : 3;
			q.wide = (v >> 1) & 1;
			q.b8 = v & 1;
			q.first = pos;
			q.count = L.size();
			q.lds = lds_need[k][v >> 2][(v >> 1) & 1][v & 1];
			if (k == MK_420)
				if (const char *pad = getenv("MIJ_LDS_PAD")) { /* experiment knob: lower the occupancy on purpose */
					size_t want = q.lds + (size_t)atol(pad);
					q.lds = want > (size_t)b->ctx->max_dyn_lds ? (size_t)b->ctx->max_dyn_lds : want;
				}
			b->launches.push_back(q);
			pos += L.size();
		}
	for (size_t i = 0; i < n; ++i)
		b->h_imgs[i] = b->slots[i].dev;

	/* ---- copies, all on the batch stream */
	HIP_TRY(copy_table(b->d_imgs, b->h_imgs, sizeof(DevImage) * n, b->stream));
	if (total)
		HIP_TRY(copy_table(b->d_work, b->h_work, sizeof(Work4) * total, b->stream));
	/* staged coefficients: own-staging slots are contiguous in the staging arena, the upload scratch and the
	 * coefficient arena in add order, so runs of them with one destination go up in one copy each */
	size_t i = 0;
	while (i < n) {
		if (b->slots[i].clone_of >= 0 || b->slots[i].dev_coef || (b->slots[i].desc.flags & MIJ_FLAG_SKIP)) {
			++i;
			continue;
		}
		if (b->slots[i].desc.flags & MIJ_FLAG_STAGED_COMPACT) { /* compact planes written by the host stage: straight to their place, escape region only when used */
			const Slot &sc = b->slots[i];
			/* small pictures: a run of neighbours goes up in ONE copy, escape regions and all (unused ones carry whatever the staging arena held:
			 * nothing reads them) -- a batch of 8192 thumbnails paid 8192 copy calls, 40 ms, where the int16 staging of round 2 went up in one */
			const size_t small = (size_t)1 << 20;
			if (sc.coef_bytes <= small) {
				size_t j = i, bytes = 0;
				while (j < n && b->slots[j].clone_of < 0 && !b->slots[j].dev_coef && !(b->slots[j].desc.flags & MIJ_FLAG_SKIP) && (b->slots[j].desc.flags & MIJ_FLAG_STAGED_COMPACT) &&
						 b->slots[j].coef_bytes <= small && b->slots[j].stage_off == sc.stage_off + bytes && b->slots[j].coef_base == sc.coef_base + bytes) {
					bytes += b->slots[j].coef_bytes;
					++j;
				}
				HIP_TRY(hipMemcpyAsync(b->d_coef + sc.coef_base, b->stage + sc.stage_off, bytes, hipMemcpyHostToDevice, b->stream));
				i = j;
				continue;
			}
			const size_t main_bytes = mij_compact_main_bytes(&sc.desc);
			HIP_TRY(hipMemcpyAsync(b->d_coef + sc.coef_base, b->stage + sc.stage_off, (sc.desc.flags & MIJ_FLAG_HAS_ESCAPES) ? sc.coef_bytes : main_bytes, hipMemcpyHostToDevice, b->stream));
			++i;
			continue;
		}
		size_t j = i, bytes = 0;
		const int fmt = b->slots[i].coef_bytes_fmt;
		const size_t s0 = b->slots[i].stage_off, c0 = b->slots[i].coef_base;
		while (j < n && b->slots[j].clone_of < 0 && !b->slots[j].dev_coef && !(b->slots[j].desc.flags & (MIJ_FLAG_SKIP | MIJ_FLAG_STAGED_COMPACT)) && b->slots[j].coef_bytes_fmt == fmt &&
				 b->slots[j].stage_off == s0 + bytes && b->slots[j].coef_base == c0 + bytes) {
			bytes += b->slots[j].coef_bytes;
			++j;
		}
		HIP_TRY(hipMemcpyAsync(fmt ? b->d_up16 + s0 : b->d_coef + c0, b->stage + s0, bytes, hipMemcpyHostToDevice, b->stream));
		i = j;
	}
	b->pack_timed = false;
	if (!pack.empty()) {
		if (!b->ev_pack0) {
			HIP_TRY(hipEventCreate(&b->ev_pack0));
			HIP_TRY(hipEventCreate(&b->ev_pack1));
		}
		HIP_TRY(hipEventRecord(b->ev_pack0, b->stream));
		hipLaunchKernelGGL(k_pack_c8, dim3((unsigned)pack.size()), dim3(256), 0, b->stream, b->d_imgs, reinterpret_cast<const WorkIdct *>(b->d_work), b->d_up16, b->d_coef, b->d_l1max);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipEventRecord(b->ev_pack1, b->stream));
		b->pack_timed = true;
	}
	for (size_t k = 0; k < n; ++k) {
		const Slot &s = b->slots[k];
		if (s.clone_of < 0)
			continue;
		const Slot &src = b->slots[(size_t)s.clone_of];
		HIP_TRY(hipMemcpyAsync(b->d_coef + s.coef_base, b->d_coef + src.coef_base, s.coef_bytes, hipMemcpyDeviceToDevice, b->stream));
	}
	b->uploaded = true;
	b->launched = false;
	return MIJ_OK;
}

/* one kernel template over (n_out 3/4, wide IDCT, compact planes) */
#define MIJ_LAUNCH_NWB(K, WT, ARGS)                                                                                                \
	do {                                                                                                                            \
		const int v_ = (L.nout == 4 ? 4 : 0) | (L.wide ? 2 : 0) | (L.b8 ? 1 : 0);                                                    \
		switch (v_) {                                                                                                                \
		case 0: hipLaunchKernelGGL((K<3, false, false>), grid, block, L.lds, b->stream, b->d_imgs, reinterpret_cast<const WT *>(wk), ARGS); break; \
		case 1: hipLaunchKernelGGL((K<3, false, true>), grid, block, L.lds, b->stream, b->d_imgs, reinterpret_cast<const WT *>(wk), ARGS); break;  \
		case 2: hipLaunchKernelGGL((K<3, true, false>), grid, block, L.lds, b->stream, b->d_imgs, reinterpret_cast<const WT *>(wk), ARGS); break;  \
		case 3: hipLaunchKernelGGL((K<3, true, true>), grid, block, L.lds, b->stream, b->d_imgs, reinterpret_cast<const WT *>(wk), ARGS); break;   \
		case 4: hipLaunchKernelGGL((K<4, false, false>), grid, block, L.lds, b->stream, b->d_imgs, reinterpret_cast<const WT *>(wk), ARGS); break; \
		case 5: hipLaunchKernelGGL((K<4, false, true>), grid, block, L.lds, b->stream, b->d_imgs, reinterpret_cast<const WT *>(wk), ARGS); break;  \
		case 6: hipLaunchKernelGGL((K<4, true, false>), grid, block, L.lds, b->stream, b->d_imgs, reinterpret_cast<const WT *>(wk), ARGS); break;  \
		default: hipLaunchKernelGGL((K<4, true, true>), grid, block, L.lds, b->stream, b->d_imgs, reinterpret_cast<const WT *>(wk), ARGS); break;  \
		}                                                                                                                            \
	} while (0)
#define MIJ_LAUNCH_WB(K, ARGS)                                                                                                     \
	do {                                                                                                                            \
		const int v_ = (L.wide ? 2 : 0) | (L.b8 ? 1 : 0);                                                                            \
		switch (v_) {                                                                                                                \
		case 0: hipLaunchKernelGGL((K<false, false>), grid, block, 0, b->stream, b->d_imgs, reinterpret_cast<const WorkIdct *>(wk), ARGS); break; \
		case 1: hipLaunchKernelGGL((K<false, true>), grid, block, 0, b->stream, b->d_imgs, reinterpret_cast<const WorkIdct *>(wk), ARGS); break;  \
		case 2: hipLaunchKernelGGL((K<true, false>), grid, block, 0, b->stream, b->d_imgs, reinterpret_cast<const WorkIdct *>(wk), ARGS); break;  \
		default: hipLaunchKernelGGL((K<true, true>), grid, block, 0, b->stream, b->d_imgs, reinterpret_cast<const WorkIdct *>(wk), ARGS); break;  \
		}                                                                                                                            \
	} while (0)
#define MIJ_COEF_OUT b->d_coef, b->d_out
#define MIJ_COEF_OUT_PLANES b->d_coef, b->d_planes

extern "C" int mij_batch_launch(mij_batch *b)
{
	if (!b)
		return set_err(MIJ_E_ARG, "batch is NULL");
	if (!b->uploaded)
		return set_err(MIJ_E_STATE, "mij_batch_launch before mij_batch_upload");
	HIP_TRY(hipSetDevice(b->ctx->device));
	for (const auto &L : b->launches) { /* in family order: pass 2 of the two-pass family runs behind every pass-1 launch */
		const dim3 grid((unsigned)L.count), block((L.kind >= MK_420 && L.kind != MK_444 && L.kind != MK_GREY && L.kind != MK_1X1C) ? (unsigned)band_threads(L.kind) : 256u);
		const Work4 *wk = b->d_work + L.first;
		switch (L.kind) {
		case MK_420:
			MIJ_LAUNCH_NWB(k_fused420, WorkBand, MIJ_COEF_OUT);
			break;
		case MK_420W:
			MIJ_LAUNCH_NWB(k_fused420w, WorkBand, MIJ_COEF_OUT);
			break;
		case MK_420X:
			MIJ_LAUNCH_NWB(k_fused420x, WorkBand, MIJ_COEF_OUT);
			break;
		case MK_420S:
			MIJ_LAUNCH_NWB(k_fused420s, WorkBand, MIJ_COEF_OUT);
			break;
		case MK_420T:
			MIJ_LAUNCH_NWB(k_fused420t, WorkBand, MIJ_COEF_OUT);
			break;
		case MK_422:
			MIJ_LAUNCH_NWB(k_fused422, WorkBand, MIJ_COEF_OUT);
			break;
		case MK_422W:
			MIJ_LAUNCH_NWB(k_fused422w, WorkBand, MIJ_COEF_OUT);
			break;
		case MK_422X:
			MIJ_LAUNCH_NWB(k_fused422x, WorkBand, MIJ_COEF_OUT);
			break;
		case MK_422S:
			MIJ_LAUNCH_NWB(k_fused422s, WorkBand, MIJ_COEF_OUT);
			break;
		case MK_422T:
			MIJ_LAUNCH_NWB(k_fused422t, WorkBand, MIJ_COEF_OUT);
			break;
		case MK_440:
			MIJ_LAUNCH_NWB(k_fused440, WorkBand, MIJ_COEF_OUT);
			break;
		case MK_440W:
			MIJ_LAUNCH_NWB(k_fused440w, WorkBand, MIJ_COEF_OUT);
			break;
		case MK_420C:
			MIJ_LAUNCH_NWB(k_fused420c, WorkBand, MIJ_COEF_OUT);
			break;
		case MK_440C:
			MIJ_LAUNCH_NWB(k_fused440c, WorkBand, MIJ_COEF_OUT);
			break;
		case MK_444:
			MIJ_LAUNCH_NWB(k_fused444, WorkIdct, MIJ_COEF_OUT);
			break;
		case MK_1X1C:
			MIJ_LAUNCH_NWB(k_fused1x1c, WorkIdct, MIJ_COEF_OUT);
			break;
		case MK_GREY:
			MIJ_LAUNCH_WB(k_fused_grey, MIJ_COEF_OUT);
			break;
		case MK_PLANES:
			MIJ_LAUNCH_WB(k_idct_planes, MIJ_COEF_OUT_PLANES);
			break;
		case MK_RESAMPLE:
			hipLaunchKernelGGL(k_resample_color, grid, block, 0, b->stream, b->d_imgs, reinterpret_cast<const WorkIdct *>(wk), b->d_planes, b->d_out);
			break;
		default: { /* MK_RS_FAST + RS_*: L.nout, L.wide = YCbCr colour */
#define MIJ_RS(KIND)                                                                                                                             \
	case KIND:                                                                                                                                    \
		if (L.wide) {                                                                                                                              \
			if (L.nout == 4) hipLaunchKernelGGL((k_resample_fast<KIND, true, 4>), grid, block, 0, b->stream, b->d_imgs, reinterpret_cast<const WorkIdct *>(wk), b->d_planes, b->d_out); \
			else hipLaunchKernelGGL((k_resample_fast<KIND, true, 3>), grid, block, 0, b->stream, b->d_imgs, reinterpret_cast<const WorkIdct *>(wk), b->d_planes, b->d_out);            \
		} else {                                                                                                                                   \
			if (L.nout == 4) hipLaunchKernelGGL((k_resample_fast<KIND, false, 4>), grid, block, 0, b->stream, b->d_imgs, reinterpret_cast<const WorkIdct *>(wk), b->d_planes, b->d_out); \
			else hipLaunchKernelGGL((k_resample_fast<KIND, false, 3>), grid, block, 0, b->stream, b->d_imgs, reinterpret_cast<const WorkIdct *>(wk), b->d_planes, b->d_out);            \
		}                                                                                                                                          \
		break
			switch (L.kind - MK_RS_FAST) {
				MIJ_RS(RS_ROW1);
				MIJ_RS(RS_V2);
				MIJ_RS(RS_H2);
				MIJ_RS(RS_HV2);
				MIJ_RS(RS_GEN2);
				MIJ_RS(RS_GEN4);
			default:
				return set_err(MIJ_E_STATE, "unknown launch kind %d", L.kind);
			}
#undef MIJ_RS
			break;
		}
		}
		HIP_TRY(hipGetLastError());
	}
	b->launched = true;
	return MIJ_OK;
}

extern "C" int mij_batch_submit(mij_batch *b)
{
	int rc = mij_batch_upload(b);
	if (rc != MIJ_OK)
		return rc;
	return mij_batch_launch(b);
}

extern "C" int mij_batch_wait(mij_batch *b)
{
	if (!b)
		return set_err(MIJ_E_ARG, "batch is NULL");
	HIP_TRY(hipSetDevice(b->ctx->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	return MIJ_OK;
}

extern "C" int mij_batch_fetch(mij_batch *b, int slot, uint8_t *dst, size_t dst_bytes)
{
	if (!b || slot < 0 || slot >= (int)b->slots.size() || !dst)
		return set_err(MIJ_E_ARG, "bad slot or destination");
	if (!b->launched)
		return set_err(MIJ_E_STATE, "mij_batch_fetch before launch");
	const Slot &s = b->slots[(size_t)slot];
	if (s.desc.flags & MIJ_FLAG_SKIP)
		return set_err(MIJ_E_STATE, "slot %d was rejected by the host stage", slot);
	const size_t bytes = (size_t)s.desc.n_out * s.desc.width * s.desc.height;
	if (dst_bytes < bytes)
		return set_err(MIJ_E_ARG, "destination too small (%zu < %zu)", dst_bytes, bytes);
	HIP_TRY(hipSetDevice(b->ctx->device));
	return d2h_bounced(b->ctx, b->stream, dst, b->d_out + s.dev.out_off, bytes);
}

extern "C" int mij_batch_fetch_all_async(mij_batch *b, uint8_t *dst, size_t dst_bytes)
{
	if (!b || !dst)
		return set_err(MIJ_E_ARG, "bad batch or destination");
	if (!b->launched)
		return set_err(MIJ_E_STATE, "mij_batch_fetch_all_async before launch");
	if (dst_bytes < b->out_used)
		return set_err(MIJ_E_ARG, "destination too small (%zu < %zu)", dst_bytes, b->out_used);
	HIP_TRY(hipSetDevice(b->ctx->device));
	if (b->out_used)
		HIP_TRY(hipMemcpyAsync(dst, b->d_out, b->out_used, hipMemcpyDeviceToHost, b->stream));
	return MIJ_OK;
}

extern "C" size_t mij_batch_out_offset(const mij_batch *b, int slot)
{
	if (!b || slot < 0 || slot >= (int)b->slots.size())
		return (size_t)-1;
	return (size_t)b->slots[(size_t)slot].dev.out_off;
}

extern "C" size_t mij_batch_out_bytes(const mij_batch *b) { return b ? b->out_used : 0; }

extern "C" void *mij_host_alloc(size_t bytes)
{
	void *p = nullptr;
	if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) {
		(void)hipGetLastError();
		set_err(MIJ_E_NOMEM, "pinned host allocation of %zu bytes failed", bytes);
		return nullptr;
	}
	return p;
}

extern "C" void mij_host_free(void *p)
{
	if (p)
		(void)hipHostFree(p);
}

extern "C" void *mij_batch_device_out(mij_batch *b, int slot)
{
	if (!b || slot < 0 || slot >= (int)b->slots.size())
		return nullptr;
	return b->d_out + b->slots[(size_t)slot].dev.out_off;
}

extern "C" int mij_batch_timer_begin(mij_batch *b)
{
	if (!b)
		return set_err(MIJ_E_ARG, "batch is NULL");
	HIP_TRY(hipSetDevice(b->ctx->device));
	HIP_TRY(hipEventRecord(b->ev_begin, b->stream));
	return MIJ_OK;
}
extern "C" int mij_batch_timer_end(mij_batch *b)
{
	if (!b)
		return set_err(MIJ_E_ARG, "batch is NULL");
	HIP_TRY(hipSetDevice(b->ctx->device));
	HIP_TRY(hipEventRecord(b->ev_end, b->stream));
	return MIJ_OK;
}
extern "C" int mij_batch_timer_elapsed_ms(mij_batch *b, float *ms)
{
	if (!b || !ms)
		return set_err(MIJ_E_ARG, "bad argument");
	HIP_TRY(hipSetDevice(b->ctx->device));
	HIP_TRY(hipEventSynchronize(b->ev_end));
	HIP_TRY(hipEventElapsedTime(ms, b->ev_begin, b->ev_end));
	return MIJ_OK;
}

/* duration of k_pack_c8 (int16 staging -> compact planes) in the last mij_batch_upload, HIP events on the batch's stream; *ms = -1
 * when that upload packed nothing (every slot staged compact by the host walk, written by the GPU entropy stage, or int16 planes) */
extern "C" int mij_batch_pack_ms(mij_batch *b, float *ms)
{
	if (!b || !ms)
		return set_err(MIJ_E_ARG, "bad argument");
	*ms = -1.0f;
	if (!b->pack_timed)
		return MIJ_OK;
	HIP_TRY(hipSetDevice(b->ctx->device));
	HIP_TRY(hipEventSynchronize(b->ev_pack1));
	HIP_TRY(hipEventElapsedTime(ms, b->ev_pack0, b->ev_pack1));
	return MIJ_OK;
}

extern "C" int mij_batch_hash_out(mij_batch *b, int slot, uint64_t *hash)
{
	if (!b || slot < 0 || slot >= (int)b->slots.size() || !hash)
		return set_err(MIJ_E_ARG, "bad argument");
	const Slot &s = b->slots[(size_t)slot];
	const size_t bytes = (size_t)s.desc.n_out * s.desc.width * s.desc.height;
	std::vector<uint8_t> tmp(bytes);
	int rc = mij_batch_fetch(b, slot, tmp.data(), bytes);
	if (rc != MIJ_OK)
		return rc;
	uint64_t h = 1469598103934665603ull;
	for (size_t i = 0; i < bytes; ++i) {
		h ^= tmp[i];
		h *= 1099511628211ull;
	}
	*hash = h;
	return MIJ_OK;
}

/* words of 16 bytes in which two device images differ (parity checks of big batches: clones against their source) */
__global__ __launch_bounds__(256) void k_count_diff(const uint4 *__restrict__ a, const uint4 *__restrict__ b, uint32_t n, unsigned long long *__restrict__ out)
{
	uint32_t bad = 0;
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
		const uint4 x = a[i], y = b[i];
		bad += (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
	}
	if (bad)
		atomicAdd(out, (unsigned long long)bad);
}

extern "C" int mij_batch_diff_slots(mij_batch *b, const int *sa, const int *sb, int n, uint64_t *ndiff)
{
	if (!b || !sa || !sb || !ndiff || n < 0)
		return set_err(MIJ_E_ARG, "bad argument");
	if (!b->launched)
		return set_err(MIJ_E_STATE, "mij_batch_diff_slots before launch");
	HIP_TRY(hipSetDevice(b->ctx->device));
	unsigned long long *d_cnt = nullptr, h_cnt = 0;
	HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_cnt), sizeof(unsigned long long)));
	hipError_t e = hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), b->stream);
	for (int i = 0; i < n && e == hipSuccess; ++i) {
		if (sa[i] < 0 || sb[i] < 0 || sa[i] >= (int)b->slots.size() || sb[i] >= (int)b->slots.size()) {
			(void)hipFree(d_cnt);
			return set_err(MIJ_E_ARG, "bad slot pair %d", i);
		}
		const Slot &x = b->slots[(size_t)sa[i]], &y = b->slots[(size_t)sb[i]];
		const size_t bytes = align_up((size_t)x.desc.n_out * x.desc.width * x.desc.height, 16);
		if (bytes != align_up((size_t)y.desc.n_out * y.desc.width * y.desc.height, 16)) {
			(void)hipFree(d_cnt);
			return set_err(MIJ_E_ARG, "slot pair %d: different sizes", i);
		}
		/* outputs are 256-byte aligned and padded in the arena, so whole 16-byte words may be compared */
		hipLaunchKernelGGL(k_count_diff, dim3(1024), dim3(256), 0, b->stream, reinterpret_cast<const uint4 *>(b->d_out + x.dev.out_off),
								 reinterpret_cast<const uint4 *>(b->d_out + y.dev.out_off), (uint32_t)(bytes / 16), d_cnt);
		e = hipGetLastError();
	}
	int rc_copy = MIJ_OK;
	if (e == hipSuccess)
		rc_copy = d2h_bounced(b->ctx, b->stream, &h_cnt, d_cnt, sizeof(h_cnt));
	else
		(void)hipStreamSynchronize(b->stream);
	(void)hipFree(d_cnt);
	if (e != hipSuccess)
		return set_err(MIJ_E_HIP, "mij_batch_diff_slots: %s", hipGetErrorString(e));
	if (rc_copy != MIJ_OK)
		return rc_copy;
	*ndiff = (uint64_t)h_cnt;
	return MIJ_OK;
}

/* ------------------------------------------------------------------ GPU entropy stage (mij_batch_entropy_*)
 *
 * See mij_entropy_kernels.h for the algorithm.  The arena holds everything the five kernels touch besides the
 * coefficient planes: the unstuffed streams, per-scan descriptors and Huffman tables, three state words and a
 * counter per subsequence, a DC difference and an L1 accumulator per block, and four verdict words per scan.
 */
#include "mij_entropy_kernels.h"

struct EsArena {
	uint8_t *stage, *d_stream; /* pinned / device, stream_cap bytes */
	size_t stream_cap;
	DevScan *h_scans, *d_scans;
	DevHuff *h_huff, *d_huff;
	EsWork *h_work, *d_work;
	size_t work_cap;
	WorkIdct *h_pack, *d_pack; /* k_es_pack: (slot, component, first block) per 256 blocks of every compact-plane slot */
	size_t pack_cap, pack_used;
	uint64_t *d_start, *d_end[2]; /* d_end[0]: the cold pass's end states, d_end[1]: the current ones (first round on) */
	uint4 *d_uni;          /* k_es_tables: one EsUni per table set */
	uint4 *d_wtab;         /* ... and one EsW (the record-form write pass's tables) */
	uint32_t *h_tabscan, *d_tabscan; /* table set -> a scan that uses it */
	uint32_t *d_qidx[2];  /* the queues of k_es_syncq, alternating by round: subsequence ... */
	uint64_t *d_qstate[2]; /* ... and the start state it has to run from; a scan's entries start at its sub_off */
	uint32_t *d_cnt, *d_base;
	size_t sub_cap;
	uint32_t sub_bits; /* bits per subsequence of this arena's scans */
	int rounds0;       /* synchronisation rounds queued before the first look at the verdicts */
	uint64_t *d_meta; /* per block: L1 of its AC coefficients | DC difference << 32 */
	uint8_t *d_zz; /* compact planes: the write pass's intermediate image, 64 bytes per block in zigzag order */
	uint64_t *d_rec;  /* compact planes, record form (k_es_writer / k_es_pack2): rec_region64 eight-byte words per subsequence */
	uint32_t rec_region64;
	size_t rec_words;
	bool use_records;
	size_t blk_cap;
	uint32_t *d_verdict, *h_verdict; /* [5][scan_cap]: anomaly, changed, total, l1max, final bit position */
	uint32_t *d_rounds_changed, *h_rounds_changed; /* [MAX_ROUNDS] sum over scans, for tuning */
	uint32_t *d_changed; /* [ES_MAX_ROUNDS][scan_cap]: subsequences moved per scan, one slice per synchronisation round -- cleared once per
	                      * launch instead of once per round (a one-picture batch queues 24 rounds: 24 fewer commands per stbi_load call) */
	std::vector<int> scan_slot; /* scan index -> batch slot */
	size_t sub_used, blk_used, work_used, scan_cap, n_tabs;
	int last_rounds, cur;
	bool in_flight;
};

static const int ES_MAX_ROUNDS = 96;

static void es_free(EsArena *e)
{
	if (!e)
		return;
	if (e->stage) (void)hipHostFree(e->stage);
	if (e->d_stream) (void)hipFree(e->d_stream);
	if (e->h_scans) (void)hipHostFree(e->h_scans);
	if (e->d_scans) (void)hipFree(e->d_scans);
	if (e->h_huff) (void)hipHostFree(e->h_huff);
	if (e->d_huff) (void)hipFree(e->d_huff);
	if (e->h_work) (void)hipHostFree(e->h_work);
	if (e->d_work) (void)hipFree(e->d_work);
	if (e->h_pack) (void)hipHostFree(e->h_pack);
	if (e->d_pack) (void)hipFree(e->d_pack);
	if (e->d_start) (void)hipFree(e->d_start);
	if (e->d_end[0]) (void)hipFree(e->d_end[0]);
	if (e->d_end[1]) (void)hipFree(e->d_end[1]);
	if (e->d_uni) (void)hipFree(e->d_uni);
	if (e->d_wtab) (void)hipFree(e->d_wtab);
	if (e->h_tabscan) (void)hipHostFree(e->h_tabscan);
	if (e->d_tabscan) (void)hipFree(e->d_tabscan);
	for (int q = 0; q < 2; ++q) {
		if (e->d_qidx[q]) (void)hipFree(e->d_qidx[q]);
		if (e->d_qstate[q]) (void)hipFree(e->d_qstate[q]);
	}
	if (e->d_cnt) (void)hipFree(e->d_cnt);
	if (e->d_base) (void)hipFree(e->d_base);
	if (e->d_meta) (void)hipFree(e->d_meta);
	if (e->d_zz) (void)hipFree(e->d_zz);
	if (e->d_rec) (void)hipFree(e->d_rec);
	if (e->d_verdict) (void)hipFree(e->d_verdict);
	if (e->h_verdict) (void)hipHostFree(e->h_verdict);
	if (e->d_rounds_changed) (void)hipFree(e->d_rounds_changed);
	if (e->d_changed) (void)hipFree(e->d_changed);
	if (e->h_rounds_changed) (void)hipHostFree(e->h_rounds_changed);
	delete e;
}

static void es_free_fwd(EsArena *e) { es_free(e); }
static void es_reset_fwd(EsArena *e)
{
	if (!e)
		return;
	e->scan_slot.clear();
	e->sub_used = e->blk_used = e->work_used = e->n_tabs = 0;
	e->pack_used = 0;
	e->in_flight = false;
}

extern "C" int mij_batch_entropy_reserve(mij_batch *b, size_t stream_bytes)
{
	if (!b || !stream_bytes)
		return set_err(MIJ_E_ARG, "mij_batch_entropy_reserve: bad argument");
	if (b->es)
		return set_err(MIJ_E_STATE, "the entropy arena exists already");
	HIP_TRY(hipSetDevice(b->ctx->device));
	EsArena *e = new (std::nothrow) EsArena();
	if (!e)
		return set_err(MIJ_E_NOMEM, "out of host memory");
	memset(static_cast<void *>(e), 0, offsetof(EsArena, scan_slot));
	const size_t n = (size_t)b->max_images;
	e->stream_cap = align_up(stream_bytes + 64 * n, 256);
	e->scan_cap = 16 * n + 1024; /* restart intervals are walked one DevScan each */
	/* one picture per batch: short subsequences (latency), see MIJ_ES_BITS_SINGLE; MIJ_ES_BITS_OVERRIDE for experiments */
	e->sub_bits = n == 1 ? MIJ_ES_BITS_SINGLE : MIJ_ES_BITS;
	if (const char *env = getenv("MIJ_ES_BITS_OVERRIDE")) {
		const long v = atol(env);
		if (v >= 256 && v <= (long)MIJ_ES_BITS && (v & 31) == 0)
			e->sub_bits = (uint32_t)v;
	}
	/* shorter subsequences settle in more rounds; a round in which nothing moves costs a launch (~8 us), a look at the verdicts
	 * a wait: measured on one-picture batches, 1024 bits with 24 rounds queued up front is the quickest (profiles/r02v_single_call.json) */
	e->rounds0 = e->sub_bits >= 4096u ? 4 : (e->sub_bits >= 2048u ? 12 : (e->sub_bits >= 1024u ? 24 : 32));
	e->sub_cap = e->stream_cap * 8 / e->sub_bits + 2 * e->scan_cap;
	e->blk_cap = b->coef_cap / 128 + n;
	e->work_cap = e->sub_cap / MIJ_ES_WG + 2 * e->scan_cap;
	e->pack_cap = e->blk_cap / 256 + 8 * n;
	hipError_t r = hipHostMalloc(reinterpret_cast<void **>(&e->stage), e->stream_cap, hipHostMallocDefault);
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_stream), e->stream_cap);
	if (r == hipSuccess) r = hipHostMalloc(reinterpret_cast<void **>(&e->h_scans), sizeof(DevScan) * e->scan_cap, hipHostMallocDefault);
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_scans), sizeof(DevScan) * e->scan_cap);
	if (r == hipSuccess) r = hipHostMalloc(reinterpret_cast<void **>(&e->h_huff), sizeof(DevHuff) * 8 * n, hipHostMallocDefault);
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_huff), sizeof(DevHuff) * 8 * n);
	if (r == hipSuccess) r = hipHostMalloc(reinterpret_cast<void **>(&e->h_work), sizeof(EsWork) * e->work_cap, hipHostMallocDefault);
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_work), sizeof(EsWork) * e->work_cap);
	if (r == hipSuccess) r = hipHostMalloc(reinterpret_cast<void **>(&e->h_pack), sizeof(WorkIdct) * e->pack_cap, hipHostMallocDefault);
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_pack), sizeof(WorkIdct) * e->pack_cap);
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_start), sizeof(uint64_t) * e->sub_cap);
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_end[0]), sizeof(uint64_t) * e->sub_cap);
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_end[1]), sizeof(uint64_t) * e->sub_cap);
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_uni), sizeof(EsUni) * n);
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_wtab), sizeof(EsW) * n);
	if (r == hipSuccess) r = hipHostMalloc(reinterpret_cast<void **>(&e->h_tabscan), sizeof(uint32_t) * n, hipHostMallocDefault);
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_tabscan), sizeof(uint32_t) * n);
	for (int q = 0; q < 2; ++q) {
		if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_qidx[q]), sizeof(uint32_t) * e->sub_cap);
		if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_qstate[q]), sizeof(uint64_t) * e->sub_cap);
	}
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_cnt), sizeof(uint32_t) * e->sub_cap);
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_base), sizeof(uint32_t) * e->sub_cap);
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_meta), sizeof(uint64_t) * e->blk_cap);
	/* the write pass's intermediate form for compact planes: the record stream (mij_entropy_kernels.h) when its arena's record indices fit
	 * 32 bits (8 GiB: streams of about 1 GiB per batch) and MIJ_ES_RECORDS / the environment allow it, else the zigzag image */
	e->rec_region64 = (e->sub_bits + MIJ_ES_REC_SLACK) / 8u;
	e->rec_words = (e->sub_cap + 1) * (size_t)e->rec_region64;
	e->use_records = MIJ_ES_RECORDS && e->rec_words < ((size_t)1 << 30) && !(getenv("MIJ_ES_RECORDS") && getenv("MIJ_ES_RECORDS")[0] == '0');
	if (e->use_records) {
		if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_rec), sizeof(uint64_t) * e->rec_words);
	} else if (r == hipSuccess)
		r = hipMalloc(reinterpret_cast<void **>(&e->d_zz), 64 * e->blk_cap);
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_verdict), sizeof(uint32_t) * 5 * e->scan_cap);
	if (r == hipSuccess) r = hipHostMalloc(reinterpret_cast<void **>(&e->h_verdict), sizeof(uint32_t) * 5 * e->scan_cap, hipHostMallocDefault);
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_rounds_changed), sizeof(uint32_t) * ES_MAX_ROUNDS);
	if (r == hipSuccess) r = hipMalloc(reinterpret_cast<void **>(&e->d_changed), sizeof(uint32_t) * ES_MAX_ROUNDS * e->scan_cap);
	if (r == hipSuccess) r = hipHostMalloc(reinterpret_cast<void **>(&e->h_rounds_changed), sizeof(uint32_t) * ES_MAX_ROUNDS, hipHostMallocDefault);
	if (r != hipSuccess) {
		es_free(e);
		return set_err(r == hipErrorOutOfMemory ? MIJ_E_NOMEM : MIJ_E_HIP, "mij_batch_entropy_reserve: %s", hipGetErrorString(r));
	}
	b->es = e;
	return MIJ_OK;
}

extern "C" uint8_t *mij_batch_entropy_stage(mij_batch *b, size_t *capacity)
{
	if (!b || !b->es)
		return nullptr;
	if (capacity)
		*capacity = b->es->stream_cap;
	return b->es->stage;
}

extern "C" int mij_batch_add_stream(mij_batch *b, const mjg_scan *scan, uint8_t *stream, size_t stream_len)
{
	if (!b || !b->es || !scan || !stream)
		return set_err(MIJ_E_ARG, "mij_batch_add_stream: bad argument (entropy arena reserved?)");
	EsArena *e = b->es;
	if (stream < e->stage || stream + stream_len + 32 > e->stage + e->stream_cap || ((size_t)(stream - e->stage) & 3u))
		return set_err(MIJ_E_ARG, "the stream must lie 4-byte aligned inside the pinned entropy region with 32 spare bytes behind it");
	if (scan->blocks_per_mcu < 1 || scan->blocks_per_mcu > 10 || scan->nblocks == 0 || stream_len >= (1u << 28))
		return set_err(MIJ_E_ARG, "bad scan description");
	/* the walk kernels compute plane addresses straight from these fields: they must describe the image's own MCU */
	{
		int rc = check_desc(&scan->desc);
		if (rc != MIJ_OK)
			return rc;
		const mij_image_desc &sd = scan->desc;
		uint32_t want = 0;
		for (int c = 0; c < sd.ncomp; ++c)
			want += (uint32_t)(sd.comp[c].h * sd.comp[c].v);
		if (want != scan->blocks_per_mcu || (uint64_t)scan->nblocks != (uint64_t)want * (uint32_t)sd.mcu_x * (uint32_t)sd.mcu_y)
			return set_err(MIJ_E_ARG, "scan block counts do not match the descriptor");
		for (uint32_t k = 0; k < scan->blocks_per_mcu; ++k) {
			const uint32_t ci = scan->blk_comp[k];
			if (ci >= (uint32_t)sd.ncomp || scan->blk_dx[k] >= (uint32_t)sd.comp[ci].h || scan->blk_dy[k] >= (uint32_t)sd.comp[ci].v)
				return set_err(MIJ_E_ARG, "scan block %u lies outside its component's MCU", k);
		}
		for (int c = 0; c < sd.ncomp; ++c)
			if (scan->dc_tab[c] > 3 || scan->ac_tab[c] < 4 || scan->ac_tab[c] > 7)
				return set_err(MIJ_E_ARG, "bad Huffman table index for component %d", c);
		/* blocks of an MCU in the order of the interleaved scan (codec/jpeg.c:1204-1219): component, then row, then column */
		uint32_t k = 0;
		for (int c = 0; c < sd.ncomp; ++c)
			for (int y = 0; y < sd.comp[c].v; ++y)
				for (int x = 0; x < sd.comp[c].h; ++x, ++k)
					if (scan->blk_comp[k] != c || scan->blk_dx[k] != x || scan->blk_dy[k] != y)
						return set_err(MIJ_E_ARG, "scan block %u is not in interleaved-scan order", k);
	}
	size_t need_pack = 0;
	for (int c = 0; c < scan->desc.ncomp; ++c)
		need_pack += (comp_tiles(scan->desc.comp[c]) * 64 + 255) / 256;
	if (b->coef_fmt && b->es->pack_used + need_pack > b->es->pack_cap)
		return set_err(MIJ_E_NOMEM, "entropy arena exhausted");
	/* one DevScan per restart interval (one for the whole stream without restart markers) */
	const uint32_t nseg = scan->n_seg ? scan->n_seg : 1u;
	if ((size_t)scan->seg_table_off + 8u * (size_t)nseg > stream_len || (scan->seg_table_off & 3u))
		return set_err(MIJ_E_ARG, "bad segment table");
	const uint32_t *table = reinterpret_cast<const uint32_t *>(stream + scan->seg_table_off);
	const uint32_t nmcu = scan->nblocks / scan->blocks_per_mcu;
	size_t need_sub = 0, need_work = 0;
	for (uint32_t g = 0; g < nseg; ++g) {
		const size_t off = table[2 * g], len = table[2 * g + 1];
		if ((off & 3u) || off + len + 32 > stream_len + 32 || off + len > scan->seg_table_off)
			return set_err(MIJ_E_ARG, "bad segment %u", g);
		const size_t ns = (len * 8 + e->sub_bits - 1) / e->sub_bits;
		need_sub += ns ? ns : 1;
		need_work += (ns ? ns : 1) / MIJ_ES_WG + 1;
	}
	if (scan->n_seg && ((uint64_t)scan->restart_mcus * (nseg - 1) >= nmcu || (uint64_t)scan->restart_mcus * nseg < nmcu))
		return set_err(MIJ_E_ARG, "segment count does not match the restart interval");
	if (e->sub_used + need_sub > e->sub_cap || e->blk_used + scan->nblocks > e->blk_cap || e->work_used + need_work > e->work_cap ||
		 e->scan_slot.size() + nseg > e->scan_cap || e->n_tabs + 1 > (size_t)b->max_images)
		return set_err(MIJ_E_NOMEM, "entropy arena exhausted");
	const int slot = add_common(b, &scan->desc, -1, true);
	if (slot < 0)
		return slot;
	Slot &s = b->slots[(size_t)slot];
	s.dev_coef = 1;
	s.es_index = (int)e->scan_slot.size();
	/* The walk writes the batch's plane format straight into HBM: compact planes by default (every decode kernel
	 * reads them; a coefficient outside -128..127 becomes an escape byte, nothing is handed back for its size),
	 * int16 tile layout on request (mij_batch_set_coef_format). */
	s.coef_bytes_fmt = b->coef_fmt ? 1 : 0;
	layout_coef(s);
	if (s.coef_bytes_fmt)
		for (int c = 0; c < scan->desc.ncomp; ++c)
			for (uint32_t f = 0, nb = (uint32_t)comp_tiles(scan->desc.comp[c]) * 64u; f < nb; f += 256) {
				WorkIdct w = {(uint32_t)slot, (uint32_t)c, f, 0u};
				e->h_pack[e->pack_used++] = w;
			}
	s.dev.es_blk_off = (uint32_t)e->blk_used;
	s.dev.es_bpm = (uint8_t)scan->blocks_per_mcu;
	for (int c = 0, j = 0; c < scan->desc.ncomp; ++c) {
		s.dev.es_j0[c] = (uint8_t)j;
		j += scan->desc.comp[c].h * scan->desc.comp[c].v;
	}
	static_assert(sizeof(DevHuff) == sizeof(mjg_huff), "mjg_huff and DevHuff must match");
	const size_t tab = e->n_tabs++;
	memcpy(&e->h_huff[8 * tab], scan->huff, sizeof(mjg_huff) * 8);
	e->h_tabscan[tab] = (uint32_t)e->scan_slot.size(); /* the first of this picture's scans */
	uint32_t first_mcu = 0;
	for (uint32_t g = 0; g < nseg; ++g) {
		const size_t k = e->scan_slot.size();
		const size_t off = table[2 * g], len = table[2 * g + 1];
		const uint32_t seg_mcus = scan->n_seg ? (g + 1 < nseg ? scan->restart_mcus : nmcu - first_mcu) : nmcu;
		const size_t ns = (len * 8 + e->sub_bits - 1) / e->sub_bits;
		DevScan &d = e->h_scans[k];
		memset(&d, 0, sizeof(d));
		d.stream_off = (uint64_t)(stream - e->stage) + off;
		d.nbits = (uint32_t)(len * 8);
		d.nsub = (uint32_t)(ns ? ns : 1);
		d.sub_off = (uint32_t)e->sub_used;
		d.img = (uint32_t)slot;
		d.nblocks = seg_mcus * scan->blocks_per_mcu;
		d.blk_off = (uint32_t)e->blk_used;
		d.bpm = scan->blocks_per_mcu;
		d.mcu_x = (uint32_t)scan->desc.mcu_x;
		d.first_mcu = first_mcu;
		d.last_seg = g + 1 == nseg;
		d.fmt = (uint32_t)s.coef_bytes_fmt;
		memcpy(d.blk_comp, scan->blk_comp, 12);
		memcpy(d.blk_dx, scan->blk_dx, 12);
		memcpy(d.blk_dy, scan->blk_dy, 12);
		memcpy(d.dc_tab, scan->dc_tab, 4);
		memcpy(d.ac_tab, scan->ac_tab, 4);
		d.tab_off = (uint32_t)(8 * tab);
		d.sub_bits = e->sub_bits;
		memcpy(d.qz, scan->qz, sizeof(d.qz));
		for (uint32_t f = 0; f < d.nsub; f += MIJ_ES_WG) {
			EsWork w = {(uint32_t)k, f};
			e->h_work[e->work_used++] = w;
		}
		e->sub_used += d.nsub;
		e->blk_used += d.nblocks;
		e->scan_slot.push_back(slot);
		first_mcu += seg_mcus;
	}
	return slot;
}

/* everything after the synchronisation rounds: offsets, write, tails, DC, verdict D2H (asynchronous) */
static int es_enqueue_tail(mij_batch *b)
{
	EsArena *e = b->es;
	hipStream_t st = b->stream;
	const size_t ns = e->scan_slot.size();
	/* the counters of the last round queued so far (slice 0 is clear when there was none) */
	uint32_t *v_anom = e->d_verdict, *v_changed = e->d_changed + (size_t)(e->last_rounds > 0 ? e->last_rounds - 1 : 0) * e->scan_cap, *v_total = e->d_verdict + 2 * e->scan_cap,
				*v_l1 = e->d_verdict + 3 * e->scan_cap, *v_pfinal = e->d_verdict + 4 * e->scan_cap;
	const dim3 gw((unsigned)e->work_used), gs((unsigned)ns), blk(256), wblk(MIJ_ES_WG);
	hipLaunchKernelGGL(k_es_offsets, gs, blk, 0, st, e->d_scans, e->d_cnt, e->d_base, v_total);
	HIP_TRY(hipGetLastError());
	{
		/* compact planes: the write pass fills a cleared intermediate image (k_es_pack then writes every byte of the
		 * tiles, k_es_dc the DC array, the first escape of a block clears its escape bytes: the planes themselves need
		 * no clearing).  int16 planes: the write pass only stores non-zero coefficients, so the planes of those images
		 * are cleared (neighbours in the arena as one range). */
		if (e->pack_used && !e->use_records)
			HIP_TRY(hipMemsetAsync(e->d_zz, 0, 64 * e->blk_used, st));
		if (e->pack_used && e->use_records) /* record form: "no subsequence began this block" until the write pass says where its records start */
			HIP_TRY(hipMemsetAsync(e->d_meta, 0xff, sizeof(uint64_t) * e->blk_used, st));
		size_t lo = 0, hi = 0;
		for (const Slot &sl : b->slots) {
			if (!sl.dev_coef || sl.clone_of >= 0 || sl.coef_bytes_fmt)
				continue;
			const size_t a = sl.coef_base, z = a + sl.coef_bytes;
			if (a == hi && hi > lo) {
				hi = z;
				continue;
			}
			if (hi > lo)
				HIP_TRY(hipMemsetAsync(b->d_coef + lo, 0, hi - lo, st));
			lo = a;
			hi = z;
		}
		if (hi > lo)
			HIP_TRY(hipMemsetAsync(b->d_coef + lo, 0, hi - lo, st));
		/* one launch per plane format in use (the other kind's workgroups leave at once) */
		bool any_fmt[2] = {false, false};
		for (size_t k = 0; k < ns; ++k)
			any_fmt[e->h_scans[k].fmt ? 1 : 0] = true;
		if (any_fmt[1] && e->use_records)
			hipLaunchKernelGGL(k_es_writer, gw, wblk, 0, st, e->d_scans, e->d_work, e->d_huff, e->d_stream, e->d_start, e->d_base, e->d_meta, v_anom, v_pfinal, e->d_rec,
									 e->rec_region64, e->d_wtab);
		else if (any_fmt[1])
			hipLaunchKernelGGL(k_es_write<true>, gw, wblk, 0, st, e->d_scans, e->d_work, e->d_huff, e->d_stream, b->d_imgs, e->d_start, e->d_base,
									 reinterpret_cast<int16_t *>(b->d_coef), e->d_meta, v_anom, v_pfinal, e->d_zz);
		if (any_fmt[0])
			hipLaunchKernelGGL(k_es_write<false>, gw, wblk, 0, st, e->d_scans, e->d_work, e->d_huff, e->d_stream, b->d_imgs, e->d_start, e->d_base,
									 reinterpret_cast<int16_t *>(b->d_coef), e->d_meta, v_anom, v_pfinal, e->d_zz);
		HIP_TRY(hipGetLastError());
		if (any_fmt[1] && !e->use_records)
			hipLaunchKernelGGL(k_es_tails<true>, gw, wblk, 0, st, e->d_scans, e->d_work, e->d_huff, e->d_stream, b->d_imgs, e->d_start, e->d_base,
									 reinterpret_cast<int16_t *>(b->d_coef), e->d_meta, e->d_rounds_changed, e->d_zz);
		if (any_fmt[0])
			hipLaunchKernelGGL(k_es_tails<false>, gw, wblk, 0, st, e->d_scans, e->d_work, e->d_huff, e->d_stream, b->d_imgs, e->d_start, e->d_base,
									 reinterpret_cast<int16_t *>(b->d_coef), e->d_meta, e->d_rounds_changed, e->d_zz);
	}
	HIP_TRY(hipGetLastError());
	if (e->pack_used) {
		if (e->use_records)
			hipLaunchKernelGGL(k_es_pack2, dim3((unsigned)e->pack_used), blk, 0, st, b->d_imgs, e->d_pack, e->d_rec, (uint32_t)e->rec_words, b->d_coef, e->d_meta);
		else
			hipLaunchKernelGGL(k_es_pack, dim3((unsigned)e->pack_used), blk, 0, st, b->d_imgs, e->d_pack, e->d_zz, b->d_coef, e->d_meta);
		HIP_TRY(hipGetLastError());
	}
	hipLaunchKernelGGL(k_es_dc, gs, blk, 0, st, e->d_scans, b->d_imgs, v_total, v_changed, reinterpret_cast<int16_t *>(b->d_coef), e->d_meta,
							 v_anom, v_l1, v_pfinal, e->d_stream);
	HIP_TRY(hipGetLastError());
	/* only the words of this batch's scans: five short runs */
	for (int v = 0; v < 5; ++v)
		HIP_TRY(copy_table_to_host(e->h_verdict + (size_t)v * e->scan_cap, v == 1 ? v_changed : e->d_verdict + (size_t)v * e->scan_cap, sizeof(uint32_t) * ns, st));
	return MIJ_OK;
}

static int es_enqueue_round(mij_batch *b)
{
	EsArena *e = b->es;
	hipStream_t st = b->stream;
	const size_t ns = e->scan_slot.size();
	if (e->last_rounds >= ES_MAX_ROUNDS)
		return set_err(MIJ_E_STATE, "too many synchronisation rounds");
	uint32_t *v_changed = e->d_changed + (size_t)e->last_rounds * e->scan_cap; /* this round's slice, cleared by the launch */
	const dim3 gw((unsigned)e->work_used), blk(MIJ_ES_WG);
	(void)ns;
	/* v_changed: what this round leaves for the next one (0 everywhere: the chains have settled) */
	const int r = e->last_rounds;
	if (r == 0)
		hipLaunchKernelGGL(k_es_sync, gw, blk, 0, st, e->d_scans, e->d_work, e->d_huff, e->d_stream, e->d_start, e->d_end[0], e->d_end[1], e->d_cnt, v_changed, e->d_qidx[0],
								 e->d_qstate[0], e->d_uni);
	else
		hipLaunchKernelGGL(k_es_syncq, gw, blk, 0, st, e->d_scans, e->d_work, e->d_huff, e->d_stream, e->d_start, e->d_end[1], e->d_cnt, v_changed - e->scan_cap,
								 e->d_qidx[(r - 1) & 1], e->d_qstate[(r - 1) & 1], v_changed, e->d_qidx[r & 1], e->d_qstate[r & 1], e->d_uni);
	HIP_TRY(hipGetLastError());
	++e->last_rounds;
	return MIJ_OK;
}

/* Asynchronous: uploads, cold pass, a fixed number of synchronisation rounds (enough for ordinary pictures),
 * write / DC passes and the verdict copy are queued on the batch's stream; nothing waits. */
extern "C" int mij_batch_entropy_launch(mij_batch *b)
{
	if (!b || !b->es)
		return set_err(MIJ_E_ARG, "mij_batch_entropy_launch: bad argument");
	EsArena *e = b->es;
	const size_t ns = e->scan_slot.size();
	e->in_flight = false;
	if (!ns)
		return MIJ_OK;
	HIP_TRY(hipSetDevice(b->ctx->device));
	hipStream_t st = b->stream;
	const size_t n = b->slots.size();
	for (size_t i = 0; i < n; ++i)
		b->h_imgs[i] = b->slots[i].dev;
	HIP_TRY(copy_table(b->d_imgs, b->h_imgs, sizeof(DevImage) * n, st));
	HIP_TRY(copy_table(e->d_scans, e->h_scans, sizeof(DevScan) * ns, st));
	HIP_TRY(copy_table(e->d_huff, e->h_huff, sizeof(DevHuff) * 8 * e->n_tabs, st));
	HIP_TRY(copy_table(e->d_tabscan, e->h_tabscan, sizeof(uint32_t) * e->n_tabs, st));
	HIP_TRY(copy_table(e->d_work, e->h_work, sizeof(EsWork) * e->work_used, st));
	HIP_TRY(copy_table(e->d_pack, e->h_pack, sizeof(WorkIdct) * e->pack_used, st));
	/* streams: one copy from the first to the last byte in use */
	size_t lo = (size_t)-1, hi = 0;
	for (size_t k = 0; k < ns; ++k) {
		const DevScan &d = e->h_scans[k];
		lo = d.stream_off < lo ? (size_t)d.stream_off : lo;
		const size_t end = (size_t)d.stream_off + d.nbits / 8 + 32;
		hi = end > hi ? end : hi;
	}
	/* The copy engine, not the copy kernel: measured both ways with four walks in flight (profiles/r02n): through
	 * k_copy_words16 the call never blocks but the 58 MB cross PCIe under a kernel that holds workgroup slots (78 Gpix/s end
	 * to end); hipMemcpyAsync blocks the host for ~7 ms one call in three or four and still comes out ahead (100 Gpix/s). */
	HIP_TRY(hipMemcpyAsync(e->d_stream + lo, e->stage + lo, hi - lo, hipMemcpyHostToDevice, st));
	/* the write pass stores every block of the MCU grid whole and its L1 word with it, so neither the
	 * coefficient planes nor the accumulators need clearing; the verdicts do */
	HIP_TRY(hipMemsetAsync(e->d_verdict, 0, sizeof(uint32_t) * 5 * e->scan_cap, st));
	HIP_TRY(hipMemsetAsync(e->d_changed, 0, sizeof(uint32_t) * ES_MAX_ROUNDS * e->scan_cap, st));
	const dim3 gw((unsigned)e->work_used), blk(MIJ_ES_WG);
	hipLaunchKernelGGL(k_es_tables, dim3((unsigned)e->n_tabs), dim3(256), 0, st, e->d_scans, e->d_tabscan, e->d_huff, e->d_uni, e->d_wtab);
	HIP_TRY(hipGetLastError());
	hipLaunchKernelGGL(k_es_cold, gw, blk, 0, st, e->d_scans, e->d_work, e->d_huff, e->d_stream, e->d_start, e->d_end[0], e->d_cnt, e->d_uni);
	HIP_TRY(hipGetLastError());
	e->cur = 0;
	e->last_rounds = 0;
	int rounds = e->rounds0; /* 4096-bit subsequences: ordinary pictures settle in two or three of four; finish() adds rounds for those that have not */
	if (const char *env = getenv("MIJ_ES_ROUNDS"))
		rounds = atoi(env) > 0 && atoi(env) <= ES_MAX_ROUNDS ? atoi(env) : rounds;
	for (int r = 0; r < rounds; ++r) {
		int rc = es_enqueue_round(b);
		if (rc != MIJ_OK)
			return rc;
	}
	int rc = es_enqueue_tail(b);
	if (rc != MIJ_OK)
		return rc;
	e->in_flight = true;
	return MIJ_OK;
}

/* Waits for mij_batch_entropy_launch.  Images whose chains had not settled get more rounds (stopping as soon as
 * a round moves nothing, ES_MAX_ROUNDS at most) and the write / DC passes are repeated; then the verdicts. */
extern "C" int mij_batch_entropy_finish(mij_batch *b, int *fallback, int cap, int *n_fallback)
{
	if (!b || !b->es || !n_fallback)
		return set_err(MIJ_E_ARG, "mij_batch_entropy_finish: bad argument");
	EsArena *e = b->es;
	*n_fallback = 0;
	const size_t ns = e->scan_slot.size();
	if (!ns || !e->in_flight)
		return MIJ_OK;
	e->in_flight = false;
	HIP_TRY(hipSetDevice(b->ctx->device));
	hipStream_t st = b->stream;
	HIP_TRY(hipStreamSynchronize(st));
	bool unsettled = false;
	for (size_t k = 0; k < ns; ++k)
		unsettled |= (e->h_verdict[k] & 8u) != 0;
	if (unsettled) {
		while (e->last_rounds < ES_MAX_ROUNDS) {
			int rc = MIJ_OK;
			for (int r = 0; r < 4 && e->last_rounds < ES_MAX_ROUNDS && rc == MIJ_OK; ++r)
				rc = es_enqueue_round(b);
			if (rc != MIJ_OK)
				return rc;
			HIP_TRY(hipMemcpyAsync(e->h_verdict + e->scan_cap, e->d_changed + (size_t)(e->last_rounds - 1) * e->scan_cap, sizeof(uint32_t) * ns, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipStreamSynchronize(st));
			uint32_t any = 0;
			for (size_t k = 0; k < ns; ++k)
				any |= e->h_verdict[e->scan_cap + k];
			if (!any)
				break;
		}
		/* the passes behind the rounds again, on clean verdicts (the last round's counters stay) */
		HIP_TRY(hipMemsetAsync(e->d_verdict, 0, sizeof(uint32_t) * e->scan_cap, st));
		HIP_TRY(hipMemsetAsync(e->d_verdict + 2 * e->scan_cap, 0, sizeof(uint32_t) * 3 * e->scan_cap, st));
		int rc = es_enqueue_tail(b);
		if (rc != MIJ_OK)
			return rc;
		HIP_TRY(hipStreamSynchronize(st));
	}
	if (getenv("MIJ_ES_DEBUG"))
		for (size_t k = 0; k < ns; ++k)
			fprintf(stderr, "es scan %zu slot %d: anomaly %u changed %u blocks %u/%u l1max %u nsub %u rounds %d\n", k, e->scan_slot[k], e->h_verdict[k],
					  e->h_verdict[e->scan_cap + k], e->h_verdict[2 * e->scan_cap + k], e->h_scans[k].nblocks,
					  e->h_verdict[3 * e->scan_cap + k], e->h_scans[k].nsub, e->last_rounds);
	/* a slot's restart intervals are consecutive scans: any verdict bit in one of them hands the image back */
	for (size_t k = 0; k < ns;) {
		const int slot = e->scan_slot[k];
		uint32_t bad = 0, l1max = 0;
		for (; k < ns && e->scan_slot[k] == slot; ++k) {
			bad |= e->h_verdict[k];
			const uint32_t v = e->h_verdict[3 * e->scan_cap + k];
			l1max = v > l1max ? v : l1max;
		}
		Slot &s = b->slots[(size_t)slot];
		if (bad) {
			if (*n_fallback < cap && fallback)
				fallback[*n_fallback] = slot;
			++*n_fallback;
			continue;
		}
		if (l1max > MIJ_BLOCK_L1_LIMIT)
			s.desc.flags |= MIJ_FLAG_WIDE_IDCT;
	}
	b->uploaded = b->launched = false;
	if (*n_fallback > cap)
		return set_err(MIJ_E_ARG, "fallback list too small (%d > %d)", *n_fallback, cap);
	return MIJ_OK;
}

extern "C" int mij_batch_entropy_run(mij_batch *b, int *fallback, int cap, int *n_fallback)
{
	int rc = mij_batch_entropy_launch(b);
	if (rc != MIJ_OK)
		return rc;
	return mij_batch_entropy_finish(b, fallback, cap, n_fallback);
}

extern "C" int mij_batch_fallback_prepare(mij_batch *b, int slot)
{
	if (!b || slot < 0 || slot >= (int)b->slots.size())
		return set_err(MIJ_E_ARG, "bad slot");
	Slot &s = b->slots[(size_t)slot];
	if (!s.dev_coef)
		return MIJ_OK;
	if (s.stage_off == MIJ_NO_STAGE)
		return set_err(MIJ_E_NOMEM, "slot %d has no staging planes (the staging arena was too small when it was added)", slot);
	s.dev_coef = 0;
	s.coef_bytes_fmt = 0; /* staged as int16 by the host walk; upload decides the format in HBM */
	layout_coef(s);
	s.desc.flags &= ~(uint32_t)MIJ_FLAG_WIDE_IDCT;
	memset(b->stage + s.stage_off, 0, s.coef_bytes);
	b->uploaded = b->launched = false;
	return MIJ_OK;
}

extern "C" int mij_batch_fetch_coef(mij_batch *b, int slot, int16_t *dst, size_t dst_elems)
{
	if (!b || slot < 0 || slot >= (int)b->slots.size() || !dst)
		return set_err(MIJ_E_ARG, "bad slot or destination");
	const Slot &s = b->slots[(size_t)slot];
	size_t elems = 0;
	for (int c = 0; c < s.desc.ncomp; ++c)
		elems += comp_tiles(s.desc.comp[c]) << 12;
	if (dst_elems < elems)
		return set_err(MIJ_E_ARG, "destination too small");
	HIP_TRY(hipSetDevice(b->ctx->device));
	if (!s.coef_bytes_fmt) {
		return d2h_bounced(b->ctx, b->stream, dst, b->d_coef + s.coef_base, elems * sizeof(int16_t));
	}
	/* compact planes: bring the region over and expand it on the host into the int16 tile layout */
	std::vector<uint8_t> raw(s.coef_bytes);
	{
		const int rc = d2h_bounced(b->ctx, b->stream, raw.data(), b->d_coef + s.coef_base, s.coef_bytes);
		if (rc != MIJ_OK)
			return rc;
	}
	size_t eoff = 0;
	for (int c = 0; c < s.desc.ncomp; ++c) {
		const size_t nt = comp_tiles(s.desc.comp[c]);
		size_t lo_o, dc_o, hi_o;
		mij_compact_offsets(&s.desc, c, &lo_o, &dc_o, &hi_o);
		const uint8_t *lo = raw.data() + lo_o, *dcp = raw.data() + dc_o, *hi = raw.data() + hi_o;
		int16_t *out = dst + eoff;
		for (size_t L = 0; L < nt * 64; ++L) {
			const uint8_t *blo = lo + ((L >> 6) << 12) + ((L & 63) << 3);
			int16_t *bo = out + ((L >> 6) << 12) + ((L & 63) << 3);
			const bool esc = (blo[0] & 1u) != 0;
			for (int P = 0; P < 64; ++P) {
				int v = (int8_t)blo[((size_t)(P >> 3) << 9) + (P & 7)];
				if (esc)
					v += 256 * (int)hi[(L << 6) + P];
				bo[((size_t)(P >> 3) << 9) + (P & 7)] = (int16_t)v;
			}
			uint16_t dcv;
			memcpy(&dcv, dcp + 2 * L, 2);
			bo[0] = (int16_t)dcv;
		}
		eoff += nt << 12;
	}
	return MIJ_OK;
}

extern "C" int mij_batch_set_coef_format(mij_batch *b, int fmt)
{
	if (!b || (fmt != MIJ_COEF_INT16 && fmt != MIJ_COEF_COMPACT))
		return set_err(MIJ_E_ARG, "bad coefficient format");
	b->coef_fmt = fmt;
	b->uploaded = b->launched = false;
	return MIJ_OK;
}

extern "C" int mij_batch_slot_escapes(mij_batch *b, int slot)
{
	if (!b || slot < 0 || slot >= (int)b->slots.size())
		return set_err(MIJ_E_ARG, "bad slot");
	const Slot &s = b->slots[(size_t)slot];
	if (!s.coef_bytes_fmt)
		return 0;
	HIP_TRY(hipSetDevice(b->ctx->device));
	int total = 0;
	for (int c = 0; c < s.desc.ncomp; ++c) {
		const size_t nt = comp_tiles(s.desc.comp[c]);
		std::vector<uint8_t> lo(nt << 12);
		{
			const int rc = d2h_bounced(b->ctx, b->stream, lo.data(), b->d_coef + s.dev.comp[c].coef_off, nt << 12);
			if (rc != MIJ_OK)
				return rc;
		}
		for (size_t L = 0; L < (size_t)(s.desc.comp[c].bw * s.desc.comp[c].bh); ++L)
			total += lo[((L >> 6) << 12) + ((L & 63) << 3)] & 1;
	}
	return total;
}

/* Measurement: how many wavefronts of the decode kernels took which sparse-block transform (mij_kernels.h, "sparse blocks").  With
 * on != 0 every image descriptor of the uploaded batch gets MIJ_DEV_COUNT_CLASSES and the device counters are cleared; the launches
 * that follow add to them; mij_batch_idct_class_counts waits for the stream and reads them.  Off by default: the timed launches of
 * bench.py run without it. */
extern "C" int mij_batch_count_idct_classes(mij_batch *b, int on)
{
	if (!b)
		return set_err(MIJ_E_ARG, "batch is NULL");
	if (!b->uploaded)
		return set_err(MIJ_E_STATE, "mij_batch_count_idct_classes before mij_batch_upload");
	HIP_TRY(hipSetDevice(b->ctx->device));
	HIP_TRY(hipStreamSynchronize(b->stream));
	const size_t n = b->slots.size();
	for (size_t i = 0; i < n; ++i) {
		if (on)
			b->h_imgs[i].flags |= MIJ_DEV_COUNT_CLASSES;
		else
			b->h_imgs[i].flags &= ~(int32_t)MIJ_DEV_COUNT_CLASSES;
	}
	HIP_TRY(copy_table(b->d_imgs, b->h_imgs, sizeof(DevImage) * n, b->stream));
	if (on) {
		void *sym = nullptr;
		HIP_TRY(hipGetSymbolAddress(&sym, HIP_SYMBOL(g_idct_class)));
		HIP_TRY(hipMemsetAsync(sym, 0, sizeof(unsigned long long) * 4, b->stream));
	}
	HIP_TRY(hipStreamSynchronize(b->stream));
	return MIJ_OK;
}

extern "C" int mij_batch_idct_class_counts(mij_batch *b, uint64_t out[4])
{
	if (!b || !out)
		return set_err(MIJ_E_ARG, "bad argument");
	HIP_TRY(hipSetDevice(b->ctx->device));
	void *sym = nullptr;
	HIP_TRY(hipGetSymbolAddress(&sym, HIP_SYMBOL(g_idct_class)));
	unsigned long long v[4];
	const int rc = d2h_bounced(b->ctx, b->stream, v, sym, sizeof(v));
	if (rc != MIJ_OK)
		return rc;
	for (int i = 0; i < 4; ++i)
		out[i] = v[i];
	return MIJ_OK;
}

extern "C" int mij_batch_slot_coef_bytes(const mij_batch *b, int slot)
{
	if (!b || slot < 0 || slot >= (int)b->slots.size())
		return 0;
	return b->slots[(size_t)slot].coef_bytes_fmt;
}

extern "C" int mij_batch_entropy_rounds(const mij_batch *b) { return b && b->es ? b->es->last_rounds : 0; }

/* ------------------------------------------------------------------ encoder (mij_enc_*)
 *
 * GPU half of the JPEG writer: colour transform + 2x2 chroma mean + float AAN fDCT + quantiser
 * (codec/jpeg_write.c:24-118, :283-352) for a batch of images; the host then Huffman-codes the data
 * units (mjw_emit, mij_host.h).  Same arena / stream / work-list design as the decode batch.
 */
#include "mij_host.h"

struct EncSlot {
	mjw_plan plan;
	EncImage dev;
	size_t stage_off, pix_bytes, du_bytes;
	int pad_w; /* pixels per staged row: the width rounded up to whole MCU columns (enc_padded_width); staged rows are packed RGB */
	int clone_of, flip;
};

struct mij_encoder {
	mij_ctx *ctx;
	hipStream_t stream;
	hipEvent_t ev_begin, ev_end;
	int max_images;
	uint8_t *stage;   /* pinned pixels */
	size_t stage_cap, stage_used;
	uint8_t *d_pix;
	size_t pix_cap, pix_used;
	int16_t *d_du;
	size_t du_cap, du_used; /* bytes */
	int16_t *h_du;   /* pinned mirror for fetch */
	EncImage *h_imgs, *d_imgs;
	WorkIdct *h_work, *d_work;
	size_t work_cap;
	size_t n_work[6], first_work[6]; /* [sub*2 + kind]: kind 0 luma units, 1 chroma units; [4]: fused 4:2:0 strips; [5]: fused 4:4:4 strips */
	std::vector<EncSlot> slots;
	bool uploaded, launched, force_generic;
};

extern "C" int mij_enc_create(mij_ctx *ctx, int max_images, size_t pixel_bytes, size_t du_bytes, mij_encoder **out)
{
	if (!ctx || !out || max_images <= 0)
		return set_err(MIJ_E_ARG, "mij_enc_create: bad argument");
	*out = nullptr;
	HIP_TRY(hipSetDevice(ctx->device));
	mij_encoder *e = new (std::nothrow) mij_encoder();
	if (!e)
		return set_err(MIJ_E_NOMEM, "out of host memory");
	e->ctx = ctx;
	e->max_images = max_images;
	e->stage = nullptr;
	e->d_pix = nullptr;
	e->d_du = nullptr;
	e->h_du = nullptr;
	e->h_imgs = e->d_imgs = nullptr;
	e->h_work = e->d_work = nullptr;
	e->work_cap = 0;
	e->stage_cap = pixel_bytes;
	e->pix_cap = pixel_bytes;
	e->du_cap = du_bytes;
	e->stage_used = e->pix_used = e->du_used = 0;
	e->uploaded = e->launched = false;
	e->force_generic = getenv("MIJ_ENC_GENERIC") != nullptr;
	e->stream = nullptr;
	e->ev_begin = e->ev_end = nullptr;
	hipError_t r = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
	if (r == hipSuccess)
		r = hipEventCreate(&e->ev_begin);
	if (r == hipSuccess)
		r = hipEventCreate(&e->ev_end);
	if (r == hipSuccess)
		r = hipHostMalloc(reinterpret_cast<void **>(&e->stage), pixel_bytes ? pixel_bytes : 16, hipHostMallocDefault);
	if (r == hipSuccess)
		r = hipMalloc(reinterpret_cast<void **>(&e->d_pix), pixel_bytes ? pixel_bytes : 16);
	if (r == hipSuccess)
		r = hipMalloc(reinterpret_cast<void **>(&e->d_du), du_bytes ? du_bytes : 16);
	if (r == hipSuccess)
		r = hipHostMalloc(reinterpret_cast<void **>(&e->h_imgs), sizeof(EncImage) * (size_t)max_images, hipHostMallocDefault);
	if (r == hipSuccess)
		r = hipMalloc(reinterpret_cast<void **>(&e->d_imgs), sizeof(EncImage) * (size_t)max_images);
	if (r != hipSuccess) {
		int code = (r == hipErrorOutOfMemory) ? MIJ_E_NOMEM : MIJ_E_HIP;
		set_err(code, "mij_enc_create: %s", hipGetErrorString(r));
		mij_enc_destroy(e);
		return code;
	}
	*out = e;
	return MIJ_OK;
}

extern "C" void mij_enc_destroy(mij_encoder *e)
{
	if (!e)
		return;
	(void)hipSetDevice(e->ctx->device);
	if (e->stream)
		(void)hipStreamSynchronize(e->stream);
	if (e->stage)
		(void)hipHostFree(e->stage);
	if (e->d_pix)
		(void)hipFree(e->d_pix);
	if (e->d_du)
		(void)hipFree(e->d_du);
	if (e->h_du)
		(void)hipHostFree(e->h_du);
	if (e->h_imgs)
		(void)hipHostFree(e->h_imgs);
	if (e->d_imgs)
		(void)hipFree(e->d_imgs);
	if (e->h_work)
		(void)hipHostFree(e->h_work);
	if (e->d_work)
		(void)hipFree(e->d_work);
	if (e->ev_begin)
		(void)hipEventDestroy(e->ev_begin);
	if (e->ev_end)
		(void)hipEventDestroy(e->ev_end);
	if (e->stream)
		(void)hipStreamDestroy(e->stream);
	delete e;
}

extern "C" int mij_enc_reset(mij_encoder *e)
{
	if (!e)
		return set_err(MIJ_E_ARG, "encoder is NULL");
	HIP_TRY(hipSetDevice(e->ctx->device));
	HIP_TRY(hipStreamSynchronize(e->stream));
	e->slots.clear();
	e->stage_used = e->pix_used = e->du_used = 0;
	e->uploaded = e->launched = false;
	return MIJ_OK;
}

/* Every picture is staged as packed RGB with rows of whole MCU columns: the last pixel of a row repeated into the padding -- the reference's
 * edge rule (codec/jpeg_write.c:294-296) applied once on the way in -- and, round 3, the reference's channel rule applied there too
 * (codec/jpeg_write.c:276-279: "ofsG = comp > 2 ? 1 : 0, ofsB = comp > 2 ? 2 : 0", r = data[p]): a grey or grey + alpha picture becomes
 * r = g = b = grey, an RGBA picture loses its alpha.  The device therefore only ever sees three channels, the same float expressions run on the
 * same values, and the strip kernels (k_encode420 / k_encode444: 16-byte or 8-byte row chunks, no column clamp) take every width and every
 * `comp`, not only RGB at multiples of 16 / 8 (the per-unit kernels remain as their test twins, mij_enc_force_generic). */
static int enc_padded_width(int width, int subsample)
{
	const int unit = subsample ? 16 : 8;
	return (width + unit - 1) / unit * unit;
}
static void enc_stage_rows(uint8_t *dst, const uint8_t *src, int width, int height, int comp, int pad_w)
{
	if (comp == 3 && pad_w == width) {
		memcpy(dst, src, (size_t)width * height * 3);
		return;
	}
	const size_t in_pitch = (size_t)width * comp, out_pitch = (size_t)pad_w * 3;
	const int og = comp > 2 ? 1 : 0, ob = comp > 2 ? 2 : 0;
	for (int y = 0; y < height; ++y) {
		uint8_t *d = dst + (size_t)y * out_pitch;
		const uint8_t *p = src + (size_t)y * in_pitch;
		if (comp == 3)
			memcpy(d, p, in_pitch);
		else
			for (int x = 0; x < width; ++x) {
				d[3 * x] = p[(size_t)x * comp];
				d[3 * x + 1] = p[(size_t)x * comp + og];
				d[3 * x + 2] = p[(size_t)x * comp + ob];
			}
		for (int x = width; x < pad_w; ++x)
			memcpy(d + (size_t)x * 3, d + (size_t)(width - 1) * 3, 3);
	}
}
extern "C" size_t mij_enc_pixel_bytes(int width, int height, int comp, int quality)
{
	mjw_plan plan;
	if (!mjw_plan_init(&plan, width, height, comp, quality))
		return 0;
	return align_up((size_t)enc_padded_width(width, plan.subsample) * height * 3, 256);
}

static int enc_add_common(mij_encoder *e, const mjw_plan &plan, const void *pixels, int flip, int clone_of)
{
	if ((int)e->slots.size() >= e->max_images)
		return set_err(MIJ_E_NOMEM, "encoder batch is full (%d images)", e->max_images);
	EncSlot s;
	s.plan = plan;
	s.flip = flip;
	s.clone_of = clone_of;
	s.pad_w = enc_padded_width(plan.width, plan.subsample);
	s.pix_bytes = align_up((size_t)s.pad_w * plan.height * 3, 256); /* staged as packed RGB whatever plan.comp is (enc_stage_rows) */
	s.du_bytes = align_up(mjw_plan_du_count(&plan) * 128, 256);
	if (e->pix_used + s.pix_bytes > e->pix_cap)
		return set_err(MIJ_E_NOMEM, "pixel arena exhausted");
	if (e->du_used + s.du_bytes > e->du_cap)
		return set_err(MIJ_E_NOMEM, "data-unit arena exhausted");
	if (clone_of < 0) {
		if (e->stage_used + s.pix_bytes > e->stage_cap)
			return set_err(MIJ_E_NOMEM, "pixel staging exhausted");
		s.stage_off = e->stage_used;
		if (pixels) /* mij_enc_add_uncopied: the caller stages the pixels itself (mij_enc_stage_pixels, several threads at once) */
			enc_stage_rows(e->stage + s.stage_off, static_cast<const uint8_t *>(pixels), plan.width, plan.height, plan.comp, s.pad_w);
		e->stage_used += s.pix_bytes;
	} else {
		s.stage_off = e->slots[(size_t)clone_of].stage_off;
	}
	memset(&s.dev, 0, sizeof(s.dev));
	s.dev.width = s.pad_w; /* the device sees whole MCU columns */
	s.dev.height = plan.height;
	s.dev.comp = 3; /* what the staging holds */
	s.dev.subsample = plan.subsample;
	s.dev.mcu_x = plan.mcu_x;
	s.dev.mcu_y = plan.mcu_y;
	s.dev.flip = flip;
	s.dev.pix_off = e->pix_used;
	s.dev.du_off = e->du_used;
	memcpy(s.dev.fy, plan.fdtbl_y, sizeof(s.dev.fy));
	memcpy(s.dev.fc, plan.fdtbl_c, sizeof(s.dev.fc));
	e->pix_used += s.pix_bytes;
	e->du_used += s.du_bytes;
	e->slots.push_back(s);
	e->uploaded = e->launched = false;
	return (int)e->slots.size() - 1;
}

extern "C" int mij_enc_add(mij_encoder *e, const void *pixels, int width, int height, int comp, int quality, int flip_vertically)
{
	if (!e || !pixels)
		return set_err(MIJ_E_ARG, "bad argument");
	mjw_plan plan;
	if (!mjw_plan_init(&plan, width, height, comp, quality))
		return set_err(MIJ_E_ARG, "bad image arguments (%dx%dx%d)", width, height, comp);
	return enc_add_common(e, plan, pixels, flip_vertically ? 1 : 0, -1);
}

extern "C" int mij_enc_add_uncopied(mij_encoder *e, int width, int height, int comp, int quality, int flip_vertically)
{
	if (!e)
		return set_err(MIJ_E_ARG, "bad argument");
	mjw_plan plan;
	if (!mjw_plan_init(&plan, width, height, comp, quality))
		return set_err(MIJ_E_ARG, "bad image arguments (%dx%dx%d)", width, height, comp);
	return enc_add_common(e, plan, nullptr, flip_vertically ? 1 : 0, -1);
}

extern "C" int mij_enc_stage_pixels(mij_encoder *e, int slot, const void *pixels)
{
	if (!e || !pixels || slot < 0 || slot >= (int)e->slots.size() || e->slots[(size_t)slot].clone_of >= 0)
		return set_err(MIJ_E_ARG, "bad slot or pixels");
	const EncSlot &s = e->slots[(size_t)slot];
	enc_stage_rows(e->stage + s.stage_off, static_cast<const uint8_t *>(pixels), s.plan.width, s.plan.height, s.plan.comp, s.pad_w);
	return MIJ_OK;
}

extern "C" void *mij_enc_staging(mij_encoder *e, int slot)
{
	if (!e || slot < 0 || slot >= (int)e->slots.size() || e->slots[(size_t)slot].clone_of >= 0)
		return nullptr;
	return e->stage + e->slots[(size_t)slot].stage_off;
}

/* every slot's data units into the encoder's pinned mirror in one copy; mij_enc_units(slot) then points at a slot's units */
extern "C" int mij_enc_fetch_all(mij_encoder *e)
{
	if (!e)
		return set_err(MIJ_E_ARG, "encoder is NULL");
	if (!e->launched)
		return set_err(MIJ_E_STATE, "mij_enc_fetch_all before launch");
	HIP_TRY(hipSetDevice(e->ctx->device));
	if (!e->h_du)
		HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->h_du), e->du_cap, hipHostMallocDefault));
	if (e->du_used)
		HIP_TRY(hipMemcpyAsync(e->h_du, e->d_du, e->du_used, hipMemcpyDeviceToHost, e->stream));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return MIJ_OK;
}

/* the same without the wait: mij_enc_wait before mij_enc_units is read */
extern "C" int mij_enc_fetch_all_async(mij_encoder *e)
{
	if (!e)
		return set_err(MIJ_E_ARG, "encoder is NULL");
	if (!e->launched)
		return set_err(MIJ_E_STATE, "mij_enc_fetch_all_async before launch");
	HIP_TRY(hipSetDevice(e->ctx->device));
	if (!e->h_du)
		HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&e->h_du), e->du_cap, hipHostMallocDefault));
	if (e->du_used)
		HIP_TRY(hipMemcpyAsync(e->h_du, e->d_du, e->du_used, hipMemcpyDeviceToHost, e->stream));
	return MIJ_OK;
}

extern "C" const int16_t *mij_enc_units(const mij_encoder *e, int slot)
{
	if (!e || !e->h_du || slot < 0 || slot >= (int)e->slots.size())
		return nullptr;
	return reinterpret_cast<const int16_t *>(reinterpret_cast<const uint8_t *>(e->h_du) + e->slots[(size_t)slot].dev.du_off);
}

extern "C" int mij_enc_add_clone(mij_encoder *e, int src_slot)
{
	if (!e || src_slot < 0 || src_slot >= (int)e->slots.size())
		return set_err(MIJ_E_ARG, "bad source slot");
	const int root = e->slots[(size_t)src_slot].clone_of >= 0 ? e->slots[(size_t)src_slot].clone_of : src_slot;
	const EncSlot src = e->slots[(size_t)root];
	return enc_add_common(e, src.plan, nullptr, src.flip, root);
}

extern "C" int mij_enc_upload(mij_encoder *e)
{
	if (!e)
		return set_err(MIJ_E_ARG, "encoder is NULL");
	HIP_TRY(hipSetDevice(e->ctx->device));
	const size_t n = e->slots.size();
	if (!n)
		return set_err(MIJ_E_STATE, "encoder batch is empty");
	std::vector<WorkIdct> work[6];
	for (size_t i = 0; i < n; ++i) {
		const EncSlot &s = e->slots[i];
		const uint32_t nm = (uint32_t)(s.plan.mcu_x * s.plan.mcu_y);
		const int sub = s.plan.subsample ? 1 : 0;
		const uint32_t ny = nm * (sub ? 4u : 1u), nc = nm * 2u;
		e->h_imgs[i] = s.dev;
		/* strips of 32 MCUs through the fused kernel: whole 16-pixel columns, packed RGB, 16-byte aligned rows (every width: enc_padded_width) */
		if (sub && !e->force_generic) { /* every comp: the staging is packed RGB */
			for (uint32_t f = 0; f < nm; f += MIJ_ENC_STRIP) {
				WorkIdct w = {(uint32_t)i, 0u, f, 0u};
				work[4].push_back(w);
			}
			continue;
		}
		/* 4:4:4 (quality above 90): strips of 64 MCUs through k_encode444: whole 8-pixel columns, packed RGB, 8-byte aligned rows */
		if (!sub && !e->force_generic) {
			for (uint32_t f = 0; f < nm; f += MIJ_ENC444_STRIP) {
				WorkIdct w = {(uint32_t)i, 0u, f, 0u};
				work[5].push_back(w);
			}
			continue;
		}
		for (uint32_t f = 0; f < ny; f += 256) {
			WorkIdct w = {(uint32_t)i, 0u, f, 0u};
			work[sub * 2 + 0].push_back(w);
		}
		for (uint32_t f = 0; f < nc; f += 256) {
			WorkIdct w = {(uint32_t)i, 1u, f, 0u};
			work[sub * 2 + 1].push_back(w);
		}
	}
	const size_t total = work[0].size() + work[1].size() + work[2].size() + work[3].size() + work[4].size() + work[5].size();
	if (total > e->work_cap)
		HIP_TRY(hipStreamSynchronize(e->stream));
	int rc = grow_pair(e->h_work, e->d_work, e->work_cap, total);
	if (rc != MIJ_OK)
		return rc;
	size_t pos = 0;
	for (int g = 0; g < 6; ++g) {
		e->first_work[g] = pos;
		e->n_work[g] = work[g].size();
		if (!work[g].empty())
			memcpy(e->h_work + pos, work[g].data(), work[g].size() * sizeof(WorkIdct));
		pos += work[g].size();
	}
	HIP_TRY(hipMemcpyAsync(e->d_imgs, e->h_imgs, sizeof(EncImage) * n, hipMemcpyHostToDevice, e->stream));
	HIP_TRY(hipMemcpyAsync(e->d_work, e->h_work, sizeof(WorkIdct) * total, hipMemcpyHostToDevice, e->stream));
	for (size_t i = 0; i < n; ++i) {
		const EncSlot &s = e->slots[i];
		if (s.clone_of < 0)
			HIP_TRY(hipMemcpyAsync(e->d_pix + s.dev.pix_off, e->stage + s.stage_off, s.pix_bytes, hipMemcpyHostToDevice, e->stream));
	}
	for (size_t i = 0; i < n; ++i) {
		const EncSlot &s = e->slots[i];
		if (s.clone_of >= 0)
			HIP_TRY(hipMemcpyAsync(e->d_pix + s.dev.pix_off, e->d_pix + e->slots[(size_t)s.clone_of].dev.pix_off, s.pix_bytes, hipMemcpyDeviceToDevice, e->stream));
	}
	e->uploaded = true;
	e->launched = false;
	return MIJ_OK;
}

extern "C" int mij_enc_launch(mij_encoder *e)
{
	if (!e)
		return set_err(MIJ_E_ARG, "encoder is NULL");
	if (!e->uploaded)
		return set_err(MIJ_E_STATE, "mij_enc_launch before mij_enc_upload");
	HIP_TRY(hipSetDevice(e->ctx->device));
	for (int g = 0; g < 6; ++g) {
		if (!e->n_work[g])
			continue;
		const dim3 grid((unsigned)e->n_work[g]), block(g >= 4 ? 192 : 256);
		const WorkIdct *wk = e->d_work + e->first_work[g];
		if (g == 4)
			hipLaunchKernelGGL(k_encode420, grid, block, MIJ_ENC_LDS, e->stream, e->d_imgs, wk, e->d_pix, e->d_du);
		else if (g == 5)
			hipLaunchKernelGGL(k_encode444, grid, block, MIJ_ENC444_LDS, e->stream, e->d_imgs, wk, e->d_pix, e->d_du);
		else if (g == 0)
			hipLaunchKernelGGL((k_encode_y<0>), grid, block, 0, e->stream, e->d_imgs, wk, e->d_pix, e->d_du);
		else if (g == 1)
			hipLaunchKernelGGL((k_encode_c<0>), grid, block, 0, e->stream, e->d_imgs, wk, e->d_pix, e->d_du);
		else if (g == 2)
			hipLaunchKernelGGL((k_encode_y<1>), grid, block, 0, e->stream, e->d_imgs, wk, e->d_pix, e->d_du);
		else
			hipLaunchKernelGGL((k_encode_c<1>), grid, block, 0, e->stream, e->d_imgs, wk, e->d_pix, e->d_du);
		HIP_TRY(hipGetLastError());
	}
	e->launched = true;
	return MIJ_OK;
}

extern "C" int mij_enc_force_generic(mij_encoder *e, int on)
{
	if (!e)
		return set_err(MIJ_E_ARG, "encoder is NULL");
	e->force_generic = on != 0;
	e->uploaded = e->launched = false;
	return MIJ_OK;
}

extern "C" int mij_enc_wait(mij_encoder *e)
{
	if (!e)
		return set_err(MIJ_E_ARG, "encoder is NULL");
	HIP_TRY(hipSetDevice(e->ctx->device));
	HIP_TRY(hipStreamSynchronize(e->stream));
	return MIJ_OK;
}

extern "C" int mij_enc_fetch(mij_encoder *e, int slot, int16_t *dst, size_t dst_elems)
{
	if (!e || slot < 0 || slot >= (int)e->slots.size() || !dst)
		return set_err(MIJ_E_ARG, "bad slot or destination");
	if (!e->launched)
		return set_err(MIJ_E_STATE, "mij_enc_fetch before launch");
	const EncSlot &s = e->slots[(size_t)slot];
	const size_t elems = mjw_plan_du_count(&s.plan) * 64;
	if (dst_elems < elems)
		return set_err(MIJ_E_ARG, "destination too small");
	HIP_TRY(hipSetDevice(e->ctx->device));
	return d2h_bounced(e->ctx, e->stream, dst, reinterpret_cast<const uint8_t *>(e->d_du) + s.dev.du_off, elems * 2);
}

extern "C" int mij_enc_plan(const mij_encoder *e, int slot, mjw_plan *out)
{
	if (!e || slot < 0 || slot >= (int)e->slots.size() || !out)
		return set_err(MIJ_E_ARG, "bad slot");
	*out = e->slots[(size_t)slot].plan;
	return MIJ_OK;
}

extern "C" int mij_enc_timer_begin(mij_encoder *e)
{
	if (!e)
		return set_err(MIJ_E_ARG, "encoder is NULL");
	HIP_TRY(hipEventRecord(e->ev_begin, e->stream));
	return MIJ_OK;
}
extern "C" int mij_enc_timer_end(mij_encoder *e)
{
	if (!e)
		return set_err(MIJ_E_ARG, "encoder is NULL");
	HIP_TRY(hipEventRecord(e->ev_end, e->stream));
	return MIJ_OK;
}
extern "C" int mij_enc_timer_elapsed_ms(mij_encoder *e, float *ms)
{
	if (!e || !ms)
		return set_err(MIJ_E_ARG, "bad argument");
	HIP_TRY(hipEventSynchronize(e->ev_end));
	HIP_TRY(hipEventElapsedTime(ms, e->ev_begin, e->ev_end));
	return MIJ_OK;
}
