"""The GPU entropy stage alone: N x 1080p 4:2:0 q=90 bitstreams already extracted (header parsed, unstuffed) in
pinned memory -> quantised coefficient planes in HBM.  Wall time of mij_batch_entropy_run (H2D of the streams,
cold pass, synchronisation rounds, offsets, write pass, DC pass, verdict D2H)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_codecs_amd as ica  # noqa: E402


def main():
    n = int(os.environ.get("BENT_N", "512"))
    w, h = 1920, 1080
    ctx = ica.Context()
    datas = [ica.synth_jpeg(w, h, s, 90) for s in range(8)]
    d = ica.HostDecoder.probe(datas[0], 3)
    cb, ob = ica.Batch.coef_bytes(d), ica.Batch.out_bytes(d)
    b = ica.Batch(ctx, n, cb * 2, cb * n, ob * n)
    b.entropy_reserve(sum(len(x) for x in datas) // len(datas) * n * 2)
    res = []
    for rep in range(4):
        b.reset()
        t0 = time.perf_counter()
        for i in range(n):
            st, _ = b.add_jpeg_stream(datas[i % len(datas)], 3)
            assert st == 1
        t1 = time.perf_counter()
        fb = b.entropy_run()
        t2 = time.perf_counter()
        assert fb == []
        res.append((t1 - t0, t2 - t1))
    ext, run = min(r[0] for r in res), min(r[1] for r in res)
    print(json.dumps({"images": n, "extract_and_add_ms_single_thread": round(ext * 1e3, 2), "entropy_run_ms": round(run * 1e3, 3),
                      "entropy_run_mpix_s": round(n * w * h / run / 1e6, 1), "sync_rounds": b.entropy_rounds()}))
    b.close()
    ctx.close()


if __name__ == "__main__":
    main()
