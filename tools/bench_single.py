"""Latency of ONE stbi_load_from_memory call (host memory in, host memory out) by picture size, with the Huffman walk on the host
and on the GPU (MIJ_GPU_WALK_MIN_PIXELS toggled per call), next to the CPU checker on the same thread: where the default threshold of
image_api.c comes from."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import image_codecs_amd as ica  # noqa: E402
import bench  # noqa: E402


def main():
    L = ica.lib()
    Lc, kind = bench.cpu_checker()
    L.stbi_image_free.argtypes = [C.c_void_p]
    x, y, c = C.c_int(), C.c_int(), C.c_int()
    out = {}
    sizes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(256, 256), (512, 512), (1024, 768), (1024, 1024), (1920, 1080), (2048, 2048), (4096, 4096)]
    for (w, h) in sizes:
        data = ica.stbi_write_jpg_to_memory(ica.synth_rgb(w, h, 1), 90)
        row = {"bytes": len(data)}
        for mode, thr in (("host_walk", str(1 << 40)), ("gpu_walk", "0")):
            os.environ["MIJ_GPU_WALK_MIN_PIXELS"] = thr
            ts = []
            for i in range(30):
                t0 = time.perf_counter()
                p = L.stbi_load_from_memory(data, len(data), C.byref(x), C.byref(y), C.byref(c), 3)
                ts.append(time.perf_counter() - t0)
                assert p
                L.stbi_image_free(p)
            row[mode + "_ms"] = round(float(np.median(ts[10:])) * 1e3, 3)
        t0 = time.perf_counter()
        for _ in range(3):
            bench.cpu_decode(Lc, kind, data)
        row["cpu_%s_ms" % kind] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
        out["%dx%d" % (w, h)] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
