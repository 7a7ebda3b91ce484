"""Kernel-resident timing of the fused 4:2:2 (h2v1) kernel: 1080p images, coefficients resident in HBM.
Algorithmic bytes per image: 128 B x 64 800 blocks read + 6 220 800 written = 14 515 200 (7 B/px).
The stream comes from the test-side writer (no libjpeg on the GPU box).  Not a bench.py line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import image_codecs_amd as ica  # noqa: E402
import helpers  # noqa: E402


def main():
    n = int(os.environ.get("B422_N", "512"))
    w, h = 1920, 1080
    ctx = ica.Context()
    plan, du = ica.host_transform(ica.synth_rgb(w, h, 1), 95)
    data = helpers.progressive_422_from_444(plan, du, 0)
    d = ica.HostDecoder.probe(data, 3)
    cb, ob = ica.Batch.coef_bytes(d), ica.Batch.out_bytes(d)
    out = {}
    for generic in (False, True):
        b = ica.Batch(ctx, n, cb, cb * n, ob * n)
        b.force_generic(generic)
        s0 = b.add_jpeg(data, 3)
        for _ in range(n - 1):
            b.add_clone(s0)
        b.upload()
        for _ in range(3):
            b.launch()
        b.wait()
        hsh = b.hash_out(n - 1)
        for _ in range(10):  # warm-up directly in front of the timed region (launch times settle after ~10 launches)
            b.launch()
        b.wait()
        b.timer_begin()
        for _ in range(10):
            b.launch()
        b.timer_end()
        ms = b.timer_ms() / 10
        blocks = sum(d.comp[c].bw * d.comp[c].bh for c in range(3))
        algo = n * (128 * blocks + 3 * w * h)
        out["two_pass" if generic else "fused422"] = {"path": b.slot_path(0), "ms_per_launch": round(ms, 4), "mpix_s": round(n * w * h / ms / 1e3, 1),
                                                      "algorithmic_GB_s": round(algo / ms / 1e6, 1), "frac_of_8TBs": round(algo / ms / 1e6 / 8000, 4), "hash": hsh}
        b.close()
    assert out["fused422"]["hash"] == out["two_pass"]["hash"]
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
