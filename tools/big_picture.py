"""One very large picture through the one-call boundary (stbi_write_jpg_to_memory -> stbi_load_from_memory), checked against the CPU
checker by hash of the pixel rows: arena sizing, 32-bit offsets and work-list limits at sizes the test suites do not reach."""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import image_codecs_amd as ica  # noqa: E402
import helpers  # noqa: E402

oracle = helpers.Oracle()
for (w, h, q) in [(int(a) for a in s.split("x")) for s in (sys.argv[1:] or ["8192x8192x90", "16384x16384x90", "20000x3000x95", "3000x20000x75"])]:
    rgb = ica.synth_rgb(w, h, 5)
    t0 = time.perf_counter()
    data = ica.stbi_write_jpg_to_memory(rgb, q)
    t1 = time.perf_counter()
    for walk in ("gpu", "host"):
        os.environ["MIJ_GPU_WALK_MIN_PIXELS"] = "0" if walk == "gpu" else str(1 << 40)
        t2 = time.perf_counter()
        r = ica.stbi_load_from_memory(data, 3)
        t3 = time.perf_counter()
        if r is None:
            print("%d x %d q%d: load (%s walk) FAILED: %s" % (w, h, q, walk, ica.stbi_failure_reason()), flush=True)
            continue
        px = r[0]
        got = hashlib.sha1(np.ascontiguousarray(px)).hexdigest()
        if walk == "gpu":
            want = hashlib.sha1(np.ascontiguousarray(oracle.load(data, 3)[1])).hexdigest()
        print("%d x %d q%d: %d bytes, write %.2f s, load (%s walk) %.3f s, %s" % (w, h, q, len(data), t1 - t0, walk, t3 - t2, "== checker" if got == want else "DIFFERS"), flush=True)
        del px
