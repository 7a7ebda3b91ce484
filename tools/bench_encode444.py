"""Kernel-resident timing of the GPU encode transform for 4:4:4 (qualities above 90): 512 x 1080p RGB images resident in HBM ->
quantised data units in HBM, fused strip kernel (k_encode444) against the per-unit kernels.  Algorithmic bytes per image:
3 x W x H read + 3 x 128 B x MCUs written (9 B/px).  Not a bench.py line."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_codecs_amd as ica  # noqa: E402


def main():
    n = int(os.environ.get("BENC_N", "512"))
    w, h = 1920, 1080
    ctx = ica.Context()
    img = ica.synth_rgb(w, h, 0)
    ref = ica.host_transform(img, 95)[1]
    pix = (w * h * 3 + 255) // 256 * 256
    dub = (240 * 135 * 3 * 128 + 255) // 256 * 256
    out = {}
    for generic in (False, True):
        enc = ica.Encoder(ctx, n, pix * n, dub * n)
        enc.force_generic(generic)
        s0 = enc.add(img, 95)
        for _ in range(n - 1):
            enc.add_clone(s0)
        enc.upload()
        enc.launch()
        enc.wait()
        assert np.array_equal(enc.fetch(0), ref) and np.array_equal(enc.fetch(n - 1), ref)
        for _ in range(10):
            enc.launch()
        enc.wait()
        enc.timer_begin()
        for _ in range(5):
            enc.launch()
        enc.timer_end()
        ms = enc.timer_ms() / 5
        algo = n * (w * h * 3 + 240 * 135 * 3 * 128)
        out["per_unit_kernels" if generic else "k_encode444"] = {"images": n, "ms_per_launch": round(ms, 4), "mpix_s": round(n * w * h / ms / 1e3, 1),
                                                               "frac_of_8TBs": round(algo / ms / 1e6 / 8000, 4), "parity_with_host_transform": True}
        enc.close()
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
