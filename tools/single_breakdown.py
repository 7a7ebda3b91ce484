"""Where the time of one stbi_load-style call goes on a one-picture batch: extract + add, the GPU walk, the fused kernel, the copy back."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_codecs_amd as ica
ctx = ica.Context()
SIZES = [tuple(map(int, a.split("x"))) for a in sys.argv[1:]] or [(1920, 1080), (4096, 4096)]
for (w, h) in SIZES:
    data = ica.stbi_write_jpg_to_memory(ica.synth_rgb(w, h, 1), 90)
    b = ica.Batch(ctx, 1, 128 << 20, 128 << 20, 64 << 20)
    b.entropy_reserve(8 << 20)
    for it in range(4):
        b.reset()
        t0 = time.perf_counter()
        st, slot = b.add_jpeg_stream(data, 3)
        t1 = time.perf_counter()
        fb = b.entropy_run()
        t2 = time.perf_counter()
        b.submit(); b.wait()
        t3 = time.perf_counter()
        px = b.fetch(slot)
        t4 = time.perf_counter()
    print(w, h, "extract+add %.3f  entropy_run %.3f  submit+wait %.3f  fetch(D2H pageable) %.3f ms  rounds %d  host fallbacks %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, b.entropy_rounds(), len(fb)))
    b.close()
