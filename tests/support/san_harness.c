/* Host stage under AddressSanitizer + UBSan (CPU build only): probe, Huffman walk and scan extraction over a
 * corpus of damaged files given on the command line.  Built and run by tests/test_host_cpu.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "jpeg_entropy.h"
int main(int argc, char **argv)
{
	int i, ok = 0, okc = 0, ex = 0;
	for (i = 1; i < argc; ++i) {
		FILE *f = fopen(argv[i], "rb");
		long n;
		uint8_t *buf;
		if (!f) continue;
		fseek(f, 0, SEEK_END); n = ftell(f); fseek(f, 0, SEEK_SET);
		buf = malloc((size_t)n ? (size_t)n : 1);
		if (fread(buf, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(buf); continue; }
		fclose(f);
		for (int req = 0; req <= 4; req += 3) {
			mij_image_desc d; const char *why = NULL;
			if (mjh_probe_memory(buf, (int)n, req, &d, &why)) {
				size_t elems = mij_image_coef_bytes(&d) / 2;
				if (elems < (64u << 20)) {
					int16_t *arena = malloc(elems * 2 + 16);
					ok += mjh_decode_memory(buf, (int)n, req, &d, arena, elems, &why);
					free(arena);
					{ /* the same walk staging compact planes itself: a region of exactly the size the batch hands out */
						mij_image_desc d2; const char *why2 = NULL;
						size_t bytes = mij_image_region_bytes(&d);
						uint8_t *region = malloc(bytes ? bytes : 1);
						okc += mjh_decode_memory_fmt(buf, (int)n, req, &d2, region, bytes, 1, &why2);
						free(region);
					}
				}
			}
			{
				mjg_scan *sc = malloc(sizeof *sc);
				size_t cap = (size_t)n + (size_t)n / 8 + 4096, len = 0;
				uint8_t *st = malloc(cap);
				ex += mjh_extract_scan(buf, (int)n, req, sc, st, cap, &len, &why) == 1;
				free(st); free(sc);
			}
		}
		free(buf);
	}
	printf("decoded ok %d (compact staging %d), extracted %d\n", ok, okc, ex);
	return ok == okc ? 0 : 1;
}
/* the one runtime function the harness needs (the real one lives in the HIP runtime) */
size_t mij_image_coef_bytes(const mij_image_desc *d) { size_t t = 0; for (int c = 0; c < d->ncomp; ++c) t += mij_plane_elems((uint32_t)(d->comp[c].bw * d->comp[c].bh)) * 2; return t; }
