"""Kernel-resident timing of the two-pass family (k_idct_planes + k_resample_color) on the headline batch
shape, forced with mij_batch_force_generic: what every layout without a fused kernel (4:2:2, grey, CMYK ...) costs."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_codecs_amd as ica  # noqa: E402


def main():
    n = int(os.environ.get("BGEN_N", "256"))
    w, h = 1920, 1080
    ctx = ica.Context()
    data = ica.synth_jpeg(w, h, 0, 90)
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
        b.timer_begin()
        for _ in range(10):
            b.launch()
        b.timer_end()
        ms = b.timer_ms() / 10
        out["two_pass" if generic else "fused"] = {"path": b.slot_path(0), "ms_per_launch": round(ms, 4), "mpix_s": round(n * w * h / ms / 1e3, 1), "hash": hsh}
        b.close()
    assert out["fused"]["hash"] == out["two_pass"]["hash"]
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
