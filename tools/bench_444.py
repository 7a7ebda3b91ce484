"""BASELINE config 4 (4096x4096 progressive 4:4:4): the host progressive stage (ten scans re-staging
the coefficient planes) timed per thread, and the register-resident 4:4:4 kernel timed with the planes
resident.  The stream comes from tests/support/prog_writer.c (no libjpeg on the GPU box).
Not a bench.py line: config 4 is a parity-test configuration."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import image_codecs_amd as ica  # noqa: E402
import helpers  # noqa: E402


def main():
    w = h = int(os.environ.get("B444_SIZE", "4096"))
    n = int(os.environ.get("B444_N", "32"))
    steps = 10
    ctx = ica.Context()
    plan, du = ica.host_transform(ica.synth_rgb(w, h, 1), 95)  # quality > 90 -> 4:4:4
    data = helpers.progressive_from_du(plan, du, 1)
    d = ica.HostDecoder.probe(data, 3)
    import numpy as np
    arena = np.empty(d.coef_elems(), np.int16)
    t0 = time.time()
    ica.HostDecoder.decode(data, 3, out=arena)
    host_s = time.time() - t0
    cb, ob = ica.Batch.coef_bytes(d), ica.Batch.out_bytes(d)
    b = ica.Batch(ctx, n, cb, cb * n, ob * n)
    s0 = b.add_jpeg(data, 3)
    for _ in range(n - 1):
        b.add_clone(s0)
    b.upload()
    for _ in range(3):
        b.launch()
    b.wait()
    assert b.slot_path(0) == 3, b.slot_path(0)
    h0 = b.hash_out(0)
    assert b.hash_out(n - 1) == h0
    for _ in range(10):  # warm-up directly in front of the timed region (launch times settle after ~10 launches)
        b.launch()
    b.wait()
    b.timer_begin()
    for _ in range(steps):
        b.launch()
    b.timer_end()
    ms = b.timer_ms() / steps
    blocks = 3 * d.comp[0].bw * d.comp[0].bh
    algo = n * (128 * blocks + 3 * w * h)
    print(json.dumps({"stream": "progressive, 10 scans, %d bytes" % len(data), "host_stage_mpix_s_per_thread": round(w * h / host_s / 1e6, 1),
                      "kernel": "mij::k_fused444<3,false>", "images": n, "size": [w, h], "ms_per_launch": round(ms, 4),
                      "mpix_s": round(n * w * h / ms / 1e3, 1), "algorithmic_GB_s": round(algo / ms / 1e6, 1),
                      "frac_of_8TBs": round(algo / ms / 1e6 / 8000, 4)}))
    b.close()
    ctx.close()


if __name__ == "__main__":
    main()
