"""Latency of ONE stbi_write_jpg_to_func call by picture size: the transform stage on the host (what stbi_write_jpg* does, like the
reference) against mij_write_jpg_to_func (transform on the GPU), next to the CPU checker's writer on the same thread."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import image_codecs_amd as ica  # noqa: E402
import bench  # noqa: E402


def main():
    L = ica.lib()
    Lc, kind = bench.cpu_checker()
    fenc = Lc.ref_encode if kind == "reference" else Lc.orc_encode
    fenc.restype = C.c_long
    fenc.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    WCB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int)
    L.stbi_write_jpg_to_func.argtypes = [WCB, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.mij_write_jpg_to_func.argtypes = [WCB, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    out = {}
    for (w, h) in ((256, 256), (512, 512), (1024, 768), (1920, 1080), (4096, 4096)):
        img = np.ascontiguousarray(ica.synth_rgb(w, h, 1))
        row = {}
        for q in (90, 95):
            for name, fn in (("host_transform", L.stbi_write_jpg_to_func), ("gpu_transform", L.mij_write_jpg_to_func)):
                sizes = []
                cb = WCB(lambda _c, data, size: sizes.append(size))
                ts = []
                for i in range(6 if w < 4096 else 3):
                    sizes.clear()
                    t0 = time.perf_counter()
                    ok = fn(cb, None, w, h, 3, img.ctypes.data, q)
                    ts.append(time.perf_counter() - t0)
                    assert ok
                row["q%d_%s_ms" % (q, name)] = round(float(np.median(ts[1:])) * 1e3, 3)
            buf = np.zeros(w * h * 3 + 4096, np.uint8)
            t0 = time.perf_counter()
            nb = fenc(buf.ctypes.data, buf.size, w, h, 3, img.ctypes.data, q)
            row["q%d_cpu_%s_ms" % (q, kind)] = round((time.perf_counter() - t0) * 1e3, 3)
        out["%dx%d" % (w, h)] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
