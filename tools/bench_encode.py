"""Kernel-resident timing of the GPU encode transform at BASELINE config 5's shape: a batch of 1080p
RGB images resident in HBM -> quantised data units in HBM (colour + 2x2 mean + fDCT + quantiser).
Algorithmic bytes per image: 6 220 800 read + 6 266 880 written.  Not a bench.py line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_codecs_amd as ica  # noqa: E402


def main():
    n = int(os.environ.get("BENC_N", "512"))
    steps = 5
    w, h = 1920, 1080
    ctx = ica.Context()
    img = ica.synth_rgb(w, h, 0)
    pix = (w * h * 3 + 255) // 256 * 256
    dub = (120 * 68 * 6 * 128 + 255) // 256 * 256
    enc = ica.Encoder(ctx, n, pix * n, dub * n)
    s0 = enc.add(img, 90)
    for _ in range(n - 1):
        enc.add_clone(s0)
    enc.upload()
    enc.launch()
    enc.wait()
    ref = ica.host_transform(img, 90)[1]
    import numpy as np
    assert np.array_equal(enc.fetch(0), ref) and np.array_equal(enc.fetch(n - 1), ref)
    for _ in range(10):  # warm-up directly in front of the timed region (launch times settle after ~10 launches)
        enc.launch()
    enc.wait()
    enc.timer_begin()
    for _ in range(steps):
        enc.launch()
    enc.timer_end()
    ms = enc.timer_ms() / steps
    algo = n * (w * h * 3 + 120 * 68 * 6 * 128)
    print(json.dumps({"kernels": "mij::k_encode_y<1> + mij::k_encode_c<1>" if os.environ.get("MIJ_ENC_GENERIC") else "mij::k_encode420", "images": n, "ms_per_launch": round(ms, 4),
                      "mpix_s": round(n * w * h / ms / 1e3, 1), "algorithmic_GB_s": round(algo / ms / 1e6, 1), "frac_of_8TBs": round(algo / ms / 1e6 / 8000, 4)}))
    enc.close()
    ctx.close()


if __name__ == "__main__":
    main()
