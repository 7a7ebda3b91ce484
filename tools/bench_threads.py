"""The drop-in call under a host thread pool: T host threads, each calling stbi_load_from_memory on 1080p 4:2:0 JPEGs one after the
other (host memory in, malloc'ed pixels out) -- what a data loader built on the reference's API does.  Aggregate Mpix/s by T."""
import ctypes as C
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_codecs_amd as ica  # noqa: E402


def main():
    if os.environ.get("BT_SCHED"):  # experiment: HIP's wait policy (1 spin, 2 yield, 4 blocking)
        hip = C.CDLL("libamdhip64.so")
        print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(C.c_uint(int(os.environ["BT_SCHED"]))))
    L = ica.lib()
    L.stbi_image_free.argtypes = [C.c_void_p]
    w, h = 1920, 1080
    datas = [ica.stbi_write_jpg_to_memory(ica.synth_rgb(w, h, s), 90) for s in range(8)]
    out = {}
    for T in (1, 2, 4, 8, 16):
        per = 40

        def worker(t):
            x, y, c = C.c_int(), C.c_int(), C.c_int()
            for k in range(per):
                d = datas[(t + k) % len(datas)]
                p = L.stbi_load_from_memory(d, len(d), C.byref(x), C.byref(y), C.byref(c), 3)
                assert p
                L.stbi_image_free(p)

        ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        for th in ths:  # warm the per-thread batches
            pass
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dt = time.perf_counter() - t0
        out["threads_%d" % T] = {"calls": T * per, "ms_per_call_per_thread": round(dt / per * 1e3, 3), "mpix_s": round(T * per * w * h / dt / 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
