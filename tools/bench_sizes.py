"""Kernel-resident timing of the fused 4:2:0 path by picture size (batches of equal pixel count resident in HBM): how the band kernel's
LDS footprint (448 bytes per MCU column) and with it the number of co-resident workgroups moves the rate.  frac as in bench.py."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import image_codecs_amd as ica  # noqa: E402
import helpers  # noqa: E402


def run(ctx, data, n, want):
    d = ica.HostDecoder.probe(data, 3)
    cb, ob = ica.Batch.coef_bytes(d), ica.Batch.out_bytes(d)
    b = ica.Batch(ctx, n, cb, cb * n, ob * n)
    s0 = b.add_jpeg(data, 3)
    for _ in range(n - 1):
        b.add_clone(s0)
    b.upload()
    for _ in range(10):
        b.launch()
    b.wait()
    ok = np.array_equal(b.fetch(0), want) and b.hash_out(n - 1) == b.hash_out(0)
    b.timer_begin()
    for _ in range(10):
        b.launch()
    b.timer_end()
    ms = b.timer_ms() / 10
    nblk = sum(int(d.comp[c].bw) * int(d.comp[c].bh) for c in range(d.ncomp))
    algo = n * (nblk * 128 + 3 * d.width * d.height)
    r = {"path": b.slot_path(0), "images": n, "ms_per_launch": round(ms, 4), "gpix_s": round(n * d.width * d.height / ms / 1e6, 1), "frac": round(algo / ms / 1e6 / 8000, 4), "parity": bool(ok)}
    b.close()
    return r


def main():
    ctx = ica.Context()
    oracle = helpers.Oracle()
    out = {}
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(640, 480), (1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (5120, 2880), (7680, 4320)]
    for (w, h) in sizes:
        n = min(16384, max(2, int(256 * 1920 * 1080 / (w * h))))
        data = ica.synth_jpeg(w, h, 0, 90)
        want = oracle.load(data, 3)[1]
        out["%dx%d" % (w, h)] = run(ctx, data, n, want)
        print("%dx%d" % (w, h), out["%dx%d" % (w, h)], flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
