/*
 * image_api.h -- public load / info / write surface (stb-compatible).
 *
 * The reference ships this header EMPTY (/root/reference/image_api.h is 0 bytes) while its
 * sources define and call the functions below; this file authors the declarations a user of
 * the reference needs, with the same names, argument meaning, ownership and error behaviour.
 * Each prototype cites the reference definition it is a drop-in for.
 *
 * Behind this surface the JPEG path runs on an AMD MI355X (gfx950): the host parses markers
 * and walks the Huffman bitstream (codec/jpeg.c:1155-1317 stays on the CPU), the quantised
 * coefficient blocks go through the C-ABI in mij.h, and de-quantisation, the 8x8 integer
 * IDCT, chroma up-sampling and YCbCr->RGB run in hand-written HIP kernels.  There is no CPU
 * fallback for those stages: without a GPU the loaders fail with reason "no gpu device".
 *
 * Only JPEG is handled (every other codec of the reference is out of scope).
 */
#ifndef IMAGE_API_H
#define IMAGE_API_H

#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned char stbi_uc;
typedef unsigned short stbi_us;

/* usage in the reference: common.c:12 (read), common.c:79 (skip), common.c:49 (eof) */
typedef struct
{
	int (*read)(void *user, char *data, int size); /* fill 'data' with up to 'size' bytes; return bytes read */
	void (*skip)(void *user, int n);                /* skip the next 'n' bytes */
	int (*eof)(void *user);                         /* non-zero at end of data */
} stbi_io_callbacks;

/* usage in the reference: codec/write_bmp.c:15, codec/jpeg_write.c:368 */
typedef void stbi_write_func(void *context, void *data, int size);

/*
 * Loaders.  *x,*y = dimensions; *comp = components in the file (3 or 1 for JPEG,
 * codec/jpeg.c:2437; comp may be NULL, :2436); req_comp in 0..4 (0 = as in file), anything
 * else fails with "bad req_comp" (:2230).  Returns a malloc block of n*x*y bytes (+1, :2293),
 * rows top-down, n interleaved bytes per pixel, no padding; free with stbi_image_free.
 * On failure returns NULL and stbi_failure_reason() gives the reference's short reason.
 */
stbi_uc *stbi_load(char const *filename, int *x, int *y, int *comp, int req_comp);                                   /* convert.c:188 */
stbi_uc *stbi_load_from_file(FILE *f, int *x, int *y, int *comp, int req_comp);                                     /* convert.c:199 */
stbi_uc *stbi_load_from_memory(stbi_uc const *buffer, int len, int *x, int *y, int *comp, int req_comp);            /* convert.c:254 */
stbi_uc *stbi_load_from_callbacks(stbi_io_callbacks const *clbk, void *user, int *x, int *y, int *comp, int req_comp); /* convert.c:261 */

/* 16-bit variants: JPEG is 8-bit, so these widen v -> v*257 (convert.c:18-33, :106-133) */
stbi_us *stbi_load_16(char const *filename, int *x, int *y, int *comp, int req_comp);                               /* convert.c:227 */
stbi_us *stbi_load_from_file_16(FILE *f, int *x, int *y, int *comp, int req_comp);                                  /* convert.c:213 */
stbi_us *stbi_load_16_from_memory(stbi_uc const *buffer, int len, int *x, int *y, int *comp, int req_comp);         /* convert.c:240 */
stbi_us *stbi_load_16_from_callbacks(stbi_io_callbacks const *clbk, void *user, int *x, int *y, int *comp, int req_comp); /* convert.c:247 */

/* header-only queries (host only; no GPU needed) */
int stbi_info(char const *filename, int *x, int *y, int *comp);                                                     /* image_api.c:74 */
int stbi_info_from_file(FILE *f, int *x, int *y, int *comp);                                                        /* image_api.c:85 */
int stbi_info_from_memory(stbi_uc const *buffer, int len, int *x, int *y, int *comp);                               /* image_api.c:119 */
int stbi_info_from_callbacks(stbi_io_callbacks const *clbk, void *user, int *x, int *y, int *comp);                 /* image_api.c:126 */
int stbi_is_16_bit(char const *filename);                                                                           /* image_api.c:96: always 0 for JPEG */
int stbi_is_16_bit_from_file(FILE *f);                                                                              /* image_api.c:107 */
int stbi_is_16_bit_from_memory(stbi_uc const *buffer, int len);                                                     /* image_api.c:133 */
int stbi_is_16_bit_from_callbacks(stbi_io_callbacks const *clbk, void *user);                                       /* image_api.c:140 */
int stbi_is_hdr(char const *filename);                                                                              /* convert.c:359: always 0 here */
int stbi_is_hdr_from_file(FILE *f);                                                                                 /* convert.c:371 */
int stbi_is_hdr_from_memory(stbi_uc const *buffer, int len);                                                        /* convert.c:345 */
int stbi_is_hdr_from_callbacks(stbi_io_callbacks const *clbk, void *user);                                          /* convert.c:388 */

/* called by the reference's users but never defined in the reference */
void stbi_image_free(void *retval_from_stbi_load);
const char *stbi_failure_reason(void);           /* thread-local here; process-global in stb */
void stbi_set_flip_vertically_on_load(int flag); /* consumed at convert.c:97 */

/*
 * JPEG writer.  comp 1..4 (2 = grey+alpha, alpha ignored; 4 = alpha ignored), quality 1..100
 * (0 -> 90); quality <= 90 writes 4:2:0, above writes 4:4:4 (codec/jpeg_write.c:220-221).
 * Returns 1 on success, 0 on bad arguments or open failure.
 */
int stbi_write_jpg_to_func(stbi_write_func *func, void *context, int x, int y, int comp, const void *data, int quality); /* codec/jpeg_write.c:368 */
int stbi_write_jpg(char const *filename, int x, int y, int comp, const void *data, int quality);                    /* codec/jpeg_write.c:376 */
void stbi_flip_vertically_on_write(int flag);    /* consumed at codec/jpeg_write.c:292 */

#ifdef __cplusplus
}
#endif

#endif /* IMAGE_API_H */
