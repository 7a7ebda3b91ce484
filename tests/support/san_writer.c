/* The writer's host stages (mjw_plan_init, mjw_transform_host, mjw_emit: image-codecs_amd/csrc/jpeg_write_host.c) under
 * AddressSanitizer + UBSan, or under MemorySanitizer where the toolchain has one (CPU builds only; tests/test_writer_golden_r3.py
 * builds and runs this).  For a spread of sizes, channel counts and qualities -- extremes included -- it
 *   - transforms into a heap block of exactly the units' size (any read or write past it is reported),
 *   - emits every picture twice, and a second picture in between, and requires the two streams of a picture to be byte-equal
 *     (state that survives a call -- a table built on first use, a sink field not set before its first use -- would show),
 *   - runs the whole set a second time in the opposite order and requires the same streams again.
 * Prints an FNV-1a hash over all streams; exit status 0 only when every comparison held. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "image_api.h"
#include "mij_host.h"

typedef struct { unsigned char *p; size_t n, cap; } sink;
static void sink_write(void *ctx, void *data, int size)
{
	sink *s = (sink *)ctx;
	if (s->n + (size_t)size > s->cap) { s->cap = (s->n + (size_t)size) * 2; s->p = realloc(s->p, s->cap); }
	memcpy(s->p + s->n, data, (size_t)size);
	s->n += (size_t)size;
}
static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

typedef struct { int w, h, comp, q, kind; } pic;
static unsigned char *make_pixels(const pic *c, uint32_t seed)
{
	size_t n = (size_t)c->w * c->h * c->comp, i;
	unsigned char *px = malloc(n);
	for (i = 0; i < n; ++i) {
		uint32_t r = lcg(&seed);
		px[i] = c->kind == 0 ? (unsigned char)r : c->kind == 1 ? (unsigned char)((i * 7 / (size_t)c->comp) & 255) : (unsigned char)((r & 1) ? 255 : 0);
	}
	return px;
}
static int encode(const pic *c, const unsigned char *px, sink *out)
{
	mjw_plan plan;
	int16_t *du;
	size_t elems;
	if (!mjw_plan_init(&plan, c->w, c->h, c->comp, c->q))
		return 0;
	elems = mjw_plan_du_count(&plan) * 64;
	du = malloc(elems * sizeof(int16_t)); /* exact size: the emitter's 16-byte loads must stay inside */
	mjw_transform_host(&plan, px, 0, du);
	out->n = 0;
	if (!mjw_emit(&plan, du, sink_write, out)) { free(du); return 0; }
	free(du);
	return 1;
}

int main(void)
{
	static const pic cases[] = {{1, 1, 3, 90, 0},   {2, 3, 3, 90, 0},    {17, 33, 3, 75, 0},  {33, 17, 1, 50, 1},  {64, 64, 3, 90, 2},  {97, 51, 4, 95, 0},
										 {250, 3, 2, 100, 2}, {3, 250, 3, 1, 2},   {128, 128, 3, 100, 0}, {129, 65, 3, 91, 1}, {640, 480, 3, 90, 0}, {16, 16, 3, 0, 2}};
	const int n = (int)(sizeof cases / sizeof cases[0]);
	sink a = {0}, b = {0}, other = {0};
	unsigned char **first = calloc((size_t)n, sizeof *first);
	size_t *first_n = calloc((size_t)n, sizeof *first_n);
	uint64_t h = 1469598103934665603ull;
	int pass, k, bad = 0;
	for (pass = 0; pass < 2; ++pass)
		for (k = 0; k < n; ++k) {
			const int i = pass ? n - 1 - k : k, o = (i + 5) % n;
			unsigned char *px = make_pixels(&cases[i], 1000u + (uint32_t)i), *po = make_pixels(&cases[o], 2000u + (uint32_t)o);
			size_t t;
			if (!encode(&cases[i], px, &a) || !encode(&cases[o], po, &other) || !encode(&cases[i], px, &b)) { fprintf(stderr, "case %d refused\n", i); bad++; }
			else if (a.n != b.n || memcmp(a.p, b.p, a.n)) { fprintf(stderr, "case %d: second emission differs from the first\n", i); bad++; }
			else if (pass == 0) { first[i] = malloc(a.n); memcpy(first[i], a.p, a.n); first_n[i] = a.n; for (t = 0; t < a.n; ++t) { h ^= a.p[t]; h *= 1099511628211ull; } }
			else if (a.n != first_n[i] || memcmp(a.p, first[i], a.n)) { fprintf(stderr, "case %d: stream of the reversed pass differs\n", i); bad++; }
			free(px); free(po);
		}
	for (k = 0; k < n; ++k) free(first[k]);
	free(first); free(first_n); free(a.p); free(b.p); free(other.p);
	printf("writer harness: %d cases x 2 orders, fnv %016llx, %d failures\n", n, (unsigned long long)h, bad);
	return bad ? 1 : 0;
}
