/*
 * oracle/ref/ref_prelude.h  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The reference tree (/root/reference) does not compile as shipped: image_api.h is
 * empty, no .c file has an #include, and ~45 foundational symbols are used but never
 * defined.  This header supplies exactly those
 * symbols, with semantics inferred from the reference's own call sites, so that the
 * reference's JPEG sources can be compiled *where they lie* into oracle/_ref/.
 * Nothing here is copied from the reference; nothing from the reference is copied
 * into this repository.
 *
 * Call sites each definition serves are cited as <file>:<line> under /root/reference.
 */
#ifndef REF_PRELUDE_H
#define REF_PRELUDE_H

#include <assert.h>
#include <limits.h>
#include <math.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* keep only the JPEG codec (image_api.c:10-53, common.c:42-178 guards) */
#define STBI_NO_PNG
#define STBI_NO_BMP
#define STBI_NO_GIF
#define STBI_NO_PSD
#define STBI_NO_PIC
#define STBI_NO_PNM
#define STBI_NO_HDR
#define STBI_NO_TGA
#define STBI_NO_ZLIB
#define STBI_NO_LINEAR

typedef unsigned char stbi_uc;
typedef unsigned short stbi_us;
typedef uint16_t stbi__uint16;
typedef int16_t stbi__int16;
typedef uint32_t stbi__uint32;
typedef int32_t stbi__int32;

#define STBIDEF extern
#define STBI_EXTERN extern
#define stbi_inline inline
#define STBI_ASSERT(x) assert(x)
#define STBI_MALLOC(sz) malloc(sz)
#define STBI_FREE(p) free(p)
#define STBI_REALLOC_SIZED(p, o, n) realloc(p, n)
#define STBI_NOTUSED(v) (void)sizeof(v)
#define STBI_MAX_DIMENSIONS (1 << 24)                         /* codec/jpeg.c:1565 */
#define STBI_SIMD_ALIGN(type, name) type name __attribute__((aligned(16))) /* codec/jpeg.c:1163 */
#define stbi_lrot(x, y) (((x) << (y)) | ((x) >> (-(y) & 31))) /* codec/jpeg.c:258 */

enum { STBI_ORDER_RGB, STBI_ORDER_BGR };                      /* image_api.c:7 */

typedef struct {                                              /* common.c:12,49,79 */
	int (*read)(void *user, char *data, int size);
	void (*skip)(void *user, int n);
	int (*eof)(void *user);
} stbi_io_callbacks;

typedef struct {                                              /* common.c:12-26, codec/jpeg.c:1559-1572 */
	stbi__uint32 img_x, img_y;
	int img_n, img_out_n;
	stbi_io_callbacks io;
	void *io_user_data;
	int read_from_callbacks;
	int buflen;
	stbi_uc buffer_start[128];
	int callback_already_read;
	stbi_uc *img_buffer, *img_buffer_end;
	stbi_uc *img_buffer_original, *img_buffer_original_end;
} stbi__context;

typedef struct {                                              /* image_api.c:5-8 */
	int bits_per_channel;
	int num_channels;
	int channel_order;
} stbi__result_info;

/* ---- failure reason (codec/jpeg.c:110 and every other stbi__err site) ---- */
static const char *stbi__g_failure_reason;
static int stbi__err(const char *str, const char *usr) { (void)usr; stbi__g_failure_reason = str; return 0; }
#define stbi__errpuc(x, y) ((unsigned char *)(size_t)(stbi__err(x, y) ? NULL : NULL))
#define stbi__errpf(x, y) ((float *)(size_t)(stbi__err(x, y) ? NULL : NULL))
STBIDEF const char *stbi_failure_reason(void);
const char *stbi_failure_reason(void) { return stbi__g_failure_reason; }

/* ---- overflow-checked allocation (codec/jpeg.c:1604,1641,1651,2293) ---- */
static void *stbi__malloc(size_t size) { return STBI_MALLOC(size); }
static int stbi__addsizes_valid(int a, int b) { if (b < 0) return 0; return a <= INT_MAX - b; }
static int stbi__mul2sizes_valid(int a, int b)
{
	if (a < 0 || b < 0) return 0;
	if (b == 0) return 1;
	return a <= INT_MAX / b;
}
static int stbi__mad2sizes_valid(int a, int b, int add) { return stbi__mul2sizes_valid(a, b) && stbi__addsizes_valid(a * b, add); }
static int stbi__mad3sizes_valid(int a, int b, int c, int add)
{
	return stbi__mul2sizes_valid(a, b) && stbi__mul2sizes_valid(a * b, c) && stbi__addsizes_valid(a * b * c, add);
}
static void *stbi__malloc_mad2(int a, int b, int add) { if (!stbi__mad2sizes_valid(a, b, add)) return NULL; return stbi__malloc(a * b + add); }
static void *stbi__malloc_mad3(int a, int b, int c, int add) { if (!stbi__mad3sizes_valid(a, b, c, add)) return NULL; return stbi__malloc(a * b * c + add); }

/* ---- I/O context start/rewind (convert.c:203,257,264; codec/jpeg.c:2461) ---- */
static void stbi__refill_buffer(stbi__context *s); /* defined at common.c:10 */
static FILE *stbi__fopen(char const *filename, char const *mode); /* defined at convert.c:160 */

static void stbi__start_mem(stbi__context *s, stbi_uc const *buffer, int len)
{
	s->io.read = NULL;
	s->read_from_callbacks = 0;
	s->callback_already_read = 0;
	s->img_buffer = s->img_buffer_original = (stbi_uc *)buffer;
	s->img_buffer_end = s->img_buffer_original_end = (stbi_uc *)buffer + len;
}
static void stbi__start_callbacks(stbi__context *s, stbi_io_callbacks *c, void *user)
{
	s->io = *c;
	s->io_user_data = user;
	s->buflen = sizeof(s->buffer_start);
	s->read_from_callbacks = 1;
	s->callback_already_read = 0;
	s->img_buffer = s->img_buffer_original = s->buffer_start;
	stbi__refill_buffer(s);
	s->img_buffer_original_end = s->img_buffer_end;
}
static int stbi__stdio_read(void *user, char *data, int size) { return (int)fread(data, 1, size, (FILE *)user); }
static void stbi__stdio_skip(void *user, int n)
{
	int ch;
	fseek((FILE *)user, n, SEEK_CUR);
	ch = fgetc((FILE *)user);
	if (ch != EOF) ungetc(ch, (FILE *)user);
}
static int stbi__stdio_eof(void *user) { return feof((FILE *)user) || ferror((FILE *)user); }
static stbi_io_callbacks stbi__stdio_callbacks = {stbi__stdio_read, stbi__stdio_skip, stbi__stdio_eof};
static void stbi__start_file(stbi__context *s, FILE *f) { stbi__start_callbacks(s, &stbi__stdio_callbacks, (void *)f); }
static void stbi__rewind(stbi__context *s)
{
	s->img_buffer = s->img_buffer_original;
	s->img_buffer_end = s->img_buffer_original_end;
}

/* ---- flip flag + free (convert.c:97; callers of stbi_load) ---- */
static int stbi__vertically_flip_on_load;
STBIDEF void stbi_set_flip_vertically_on_load(int flag);
void stbi_set_flip_vertically_on_load(int flag) { stbi__vertically_flip_on_load = flag; }
STBIDEF void stbi_image_free(void *p);
void stbi_image_free(void *p) { STBI_FREE(p); }

/* ---- public prototypes used before their definition ---- */
STBIDEF stbi_uc *stbi_load_from_file(FILE *f, int *x, int *y, int *comp, int req_comp);          /* convert.c:194 */
STBIDEF stbi__uint16 *stbi_load_from_file_16(FILE *f, int *x, int *y, int *comp, int req_comp);  /* convert.c:233 */
STBIDEF int stbi_info_from_file(FILE *f, int *x, int *y, int *comp);                              /* image_api.c:80 */
STBIDEF int stbi_is_16_bit_from_file(FILE *f);                                                    /* image_api.c:102 */
STBIDEF int stbi_is_hdr_from_file(FILE *f);                                                       /* convert.c:365 */
STBIDEF stbi_uc *stbi_load_from_memory(stbi_uc const *buffer, int len, int *x, int *y, int *comp, int req_comp);
STBIDEF int stbi_info_from_memory(stbi_uc const *buffer, int len, int *x, int *y, int *comp);

/* ---- writer side (codec/write_bmp.c:15,53-84; codec/jpeg_write.c:292,370-382) ---- */
#define STBIWDEF extern
#define STBIW_ASSERT(x) assert(x)
#define STBIW_MALLOC(sz) malloc(sz)
#define STBIW_REALLOC(p, n) realloc(p, n)
#define STBIW_REALLOC_SIZED(p, o, n) realloc(p, n)
#define STBIW_FREE(p) free(p)
#define STBIW_MEMMOVE(a, b, sz) memmove(a, b, sz)
#define STBIW_UCHAR(x) (unsigned char)((x) & 0xff)
typedef void stbi_write_func(void *context, void *data, int size);
typedef struct {
	stbi_write_func *func;
	void *context;
	unsigned char buffer[64];
	int buf_used;
} stbi__write_context;
static int stbi__flip_vertically_on_write;
STBIWDEF void stbi_flip_vertically_on_write(int flag);
void stbi_flip_vertically_on_write(int flag) { stbi__flip_vertically_on_write = flag; }
static void stbi__start_write_callbacks(stbi__write_context *s, stbi_write_func *c, void *context) { s->func = c; s->context = context; }
static void stbi__stdio_write(void *context, void *data, int size) { fwrite(data, 1, size, (FILE *)context); }
static int stbi__start_write_file(stbi__write_context *s, const char *filename)
{
	FILE *f = fopen(filename, "wb");
	stbi__start_write_callbacks(s, stbi__stdio_write, (void *)f);
	return f != NULL;
}
static void stbi__end_write_file(stbi__write_context *s) { fclose((FILE *)s->context); }
STBIWDEF int stbi_write_jpg_to_func(stbi_write_func *func, void *context, int x, int y, int comp, const void *data, int quality);
STBIWDEF int stbi_write_jpg(char const *filename, int x, int y, int comp, const void *data, int quality);
STBIWDEF int stbi_write_bmp_to_func(stbi_write_func *func, void *context, int x, int y, int comp, const void *data);
STBIWDEF int stbi_write_bmp(char const *filename, int x, int y, int comp, const void *data);

#endif /* REF_PRELUDE_H */
