"""End to end through the batch front ends (JPEG bytes in host RAM -> pixels in HBM, one call per batch) by picture size:
mjh_decode_batch (the default: Huffman walk on the GPU where it applies) against mjh_decode_batch_host (every walk on the host threads).
Where do small pictures belong?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import image_codecs_amd as ica  # noqa: E402

threads = int(os.environ.get("BBS_THREADS", "16"))
ctx = ica.Context()
for (w, h) in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(64, 64), (128, 128), (256, 256), (512, 512), (1024, 768), (1920, 1080)]:
    n = min(8192, max(16, int(128 * 1920 * 1080 / (w * h))))
    datas = [ica.synth_jpeg(w, h, s, 90) for s in range(8)]
    jl = [datas[i % 8] for i in range(n)]
    d = ica.HostDecoder.probe(datas[0], 3)
    cb, ob = ica.Batch.coef_bytes(d), ica.Batch.out_bytes(d)
    row = {}
    for name, mode in (("gpu_walk", None), ("host_walk", False)):
        b = ica.Batch(ctx, n, cb * n, cb * n, ob * n)
        if mode is None:
            b.entropy_reserve(sum(len(x) * 9 // 8 + 4352 for x in jl))
        ts = []
        for it in range(4):
            b.reset()
            t0 = time.perf_counter()
            ok, slots, reasons = b.decode_jpegs(jl, 3, threads, gpu_entropy=mode)
            b.submit()
            b.wait()
            ts.append(time.perf_counter() - t0)
            assert ok == n, reasons[:3]
        row[name] = round(n * w * h / min(ts[1:]) / 1e6, 1)
        b.close()
    print("%dx%d  %d pictures per call: default front end (mjh_decode_batch) %.1f Mpix/s, mjh_decode_batch_host (%d threads) %.1f Mpix/s" % (w, h, n, row["gpu_walk"], threads, row["host_walk"]), flush=True)
ctx.close()
