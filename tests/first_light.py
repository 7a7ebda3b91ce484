"""GPU bring-up script (not a test; lives under tests/ because it links the checker): parity of the GPU path against the reference-built
oracle/_ref library on a spread of sizes, then a rough timing of the fused kernel."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_codecs_amd as ica  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libstbref.so"))
R.stbi_load_from_memory.restype = C.POINTER(C.c_ubyte)
R.stbi_load_from_memory.argtypes = [C.c_char_p, C.c_int] + [C.POINTER(C.c_int)] * 3 + [C.c_int]
R.stbi_image_free.argtypes = [C.c_void_p]


def ref_load(data, req):
    x, y, c = C.c_int(), C.c_int(), C.c_int()
    p = R.stbi_load_from_memory(data, len(data), x, y, c, req)
    if not p:
        return None
    n = req if req else c.value
    a = np.ctypeslib.as_array(p, shape=(y.value * x.value * n,)).reshape(y.value, x.value, n).copy()
    R.stbi_image_free(p)
    return a


def main():
    ctx = ica.Context()
    print("device:", ctx.info())
    bad = 0
    rng = np.random.default_rng(0)
    cases = [(64, 64), (1, 1), (2, 3), (17, 33), (33, 17), (16, 16), (15, 15), (31, 47), (100, 60), (128, 128), (250, 130), (640, 360), (1920, 1080)]
    for (w, h) in cases:
        for q in (90, 95, 50):
            for req in (3, 4, 1, 0):
                img = ica.synth_rgb(w, h, seed=w + h) if w > 64 else rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
                data = ica.stbi_write_jpg_to_memory(img, q)
                ref = ref_load(data, req)
                got = ica.stbi_load_from_memory(data, req)
                if got is None:
                    print("FAIL", w, h, q, req, ica.stbi_failure_reason())
                    bad += 1
                    continue
                ok = np.array_equal(got[0], ref)
                if not ok:
                    diff = np.argwhere(got[0] != ref)
                    print("MISMATCH", w, h, q, req, "ndiff", len(diff), "first", diff[:5].tolist())
                    bad += 1
    # forced generic path on 4:2:0 must agree with the fused path
    print("mismatches:", bad)

    # timing: N clones of one 1080p image, device resident
    data = ica.synth_jpeg(1920, 1080, 0, 90)
    d = ica.HostDecoder.probe(data, 3)
    N = int(os.environ.get("FL_N", "256"))
    cb, ob = ica.Batch.coef_bytes(d), ica.Batch.out_bytes(d)
    b = ica.Batch(ctx, N, cb, cb * N, ob * N)
    s0 = b.add_jpeg(data, 3)
    for _ in range(N - 1):
        b.add_clone(s0)
    b.upload()
    b.launch()
    b.wait()
    ref = ref_load(data, 3)
    for s in (0, 1, N - 1):
        print("slot", s, "path", b.slot_path(s), "equal", np.array_equal(b.fetch(s), ref))
    for rows in (None,):
        for it in range(3):
            b.timer_begin()
            for _ in range(5):
                b.launch()
            b.timer_end()
            ms = b.timer_ms() / 5
            px = N * 1920 * 1080
            print("fused launch: %.3f ms  %.1f Gpix/s  %.2f TB/s algorithmic" % (ms, px / ms / 1e6, (N * 12487680) / ms / 1e9))
    b.close()
    ctx.close()
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
