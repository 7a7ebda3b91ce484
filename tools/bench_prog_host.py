"""Host stage of a progressive 4:4:4 stream (ten scans, the test-side writer's script 1) on ONE thread: mjh_decode_memory_fmt into a
staging region, best of N, for the library named by MIJ_LIB (interleave two builds from the shell for an A/B on one box)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import image_codecs_amd as ica  # noqa: E402
import helpers  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
plan, du = ica.host_transform(ica.synth_rgb(size, size, 1), 95)
data = helpers.progressive_from_du(plan, du, 1)
best = 1e9
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    t = time.perf_counter()
    ica.host_decode_staged(data, 3, True)
    best = min(best, time.perf_counter() - t)
print("%s: %d x %d, %d bytes: %.1f ms = %.1f Mpix/s on one thread" % (os.path.basename(os.path.dirname(os.environ.get("MIJ_LIB", "lib/x"))), size, size, len(data), best * 1e3, size * size / best / 1e6))
