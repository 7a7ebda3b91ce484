/*
 * mij_host.h -- the host half of the JPEG decode path, as plain C entry points.
 *
 * This is the part of the reference that stays on the CPU -- marker parsing, Huffman tables and
 * the sequential entropy-coded-segment walk (codec/jpeg.c:88-558, :1119-1756) -- restated so
 * that every decoded block lands, still quantised, in the tile-layout staging planes of mij.h
 * instead of being de-quantised and inverse-transformed on the spot
 * (codec/jpeg.c:1178,:1217,:1342).  stbi_load* (image_api.h) is built from exactly these calls.
 */
#ifndef MIJ_HOST_H
#define MIJ_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "mij.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Type test + header + describe (stbi__jpeg_test, then codec/jpeg.c:1670-1699, :2241-2249):
 * fills *desc so the caller can size the coefficient planes.  Returns 1, or 0 with *reason = the
 * reference's short failure string ("unknown image type", "bad req_comp", "only 8-bit", ...).
 */
int mjh_probe_memory(const uint8_t *buf, int len, int req_comp, mij_image_desc *desc, const char **reason);

/*
 * The same plus the scans (codec/jpeg.c:1713-1755) into `arena`: the component planes in tile
 * layout, back to back in component order -- exactly the layout of the staging a mij batch hands
 * out (mij_batch_coef(b, slot, 0)).  The callee zero-fills what it uses.  desc->flags gets
 * MIJ_FLAG_WIDE_IDCT when the fast IDCT's range guarantee does not hold.  Returns 1 / 0 + reason.
 */
int mjh_decode_memory(const uint8_t *buf, int len, int req_comp, mij_image_desc *desc, int16_t *arena, size_t arena_elems, const char **reason);

/*
 * Batch front end: the host stage of n JPEGs on `threads` host threads (one image per task),
 * straight into a mij batch's pinned staging.  Images are added to the batch in input order
 * (slots[i] = the slot of image i, or -1 with reasons[i] set when its header is rejected);
 * an image whose entropy data is rejected keeps its slot but is flagged MIJ_FLAG_SKIP (reasons[i]
 * set, slot reported as -1 - slot).  Follow with mij_batch_submit().  Returns the number of
 * images decoded successfully, or a negative MIJ_E_* code when the batch arenas are too small.
 * The end-to-end rate of this path is bounded by the Huffman walk (about 0.25-0.5 Gpix/s per host
 * core) and by PCIe (3 B/px up), not by the GPU.
 */
int mjh_decode_batch(mij_batch *b, const uint8_t *const *bufs, const int *lens, int n, int req_comp, int threads, int *slots, const char **reasons);

#ifdef __cplusplus
}
#endif

#endif /* MIJ_HOST_H */
