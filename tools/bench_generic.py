"""Kernel-resident timing of the two-pass family on 1080p batches resident in HBM: the headline 4:2:0 images forced off the fused
kernel (mij_batch_force_generic 1: pass 2 compiled per resampler, k_resample_fast; 2: the run-time-general k_resample_color),
and the layouts that only have the two-pass path (4:4:0, 4:1:1, CMYK, YCCK), each checked against the CPU checker in the run.
frac = algorithmic bytes (2 B x coefficients + n_out x W x H) / time / 8 TB/s, as for the fused kernels."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import image_codecs_amd as ica  # noqa: E402
import helpers  # noqa: E402


def run(ctx, data, n, level, want):
    d = ica.HostDecoder.probe(data, 3)
    cb, ob = ica.Batch.coef_bytes(d), ica.Batch.out_bytes(d)
    b = ica.Batch(ctx, n, cb, cb * n, ob * n)
    b.force_generic(level)
    s0 = b.add_jpeg(data, 3)
    for _ in range(n - 1):
        b.add_clone(s0)
    b.upload()
    for _ in range(10):
        b.launch()
    b.wait()
    ok = want is None or (np.array_equal(b.fetch(0), want) and b.hash_out(n - 1) == b.hash_out(0))
    b.timer_begin()
    for _ in range(10):
        b.launch()
    b.timer_end()
    ms = b.timer_ms() / 10
    nblk = sum(int(d.comp[c].bw) * int(d.comp[c].bh) for c in range(d.ncomp))
    algo = n * (nblk * 128 + 3 * d.width * d.height)
    r = {"path": b.slot_path(0), "ms_per_launch": round(ms, 4), "gpix_s": round(n * d.width * d.height / ms / 1e6, 1), "frac": round(algo / ms / 1e6 / 8000, 4), "parity": bool(ok)}
    b.close()
    return r


def main():
    n = int(os.environ.get("BGEN_N", "256"))
    w, h = 1920, 1080
    ctx = ica.Context()
    oracle = helpers.Oracle()
    out = {}
    data = ica.synth_jpeg(w, h, 0, 90)
    want = oracle.load(data, 3)[1]
    out["420_fused"] = run(ctx, data, n, 0, want)
    out["420_two_pass_specialised"] = run(ctx, data, n, 1, want)
    out["420_two_pass_general"] = run(ctx, data, n, 2, want)
    plan, du = ica.host_transform(ica.synth_rgb(w, h, 0), 90 + 1)  # quality > 90: the 4:4:4 writer plan the layout helper starts from
    for name, hv, app14 in (("440", [(1, 2), (1, 1), (1, 1)], -1), ("411", [(4, 1), (1, 1), (1, 1)], -1), ("cmyk", [(1, 1)] * 4, 0), ("ycck_420", [(2, 2), (1, 1), (1, 1), (2, 2)], 2)):
        data = helpers.baseline_layout_from_444(plan, du, hv, app14)
        want = oracle.load(data, 3)[1]
        if name == "440":
            out["440_fused"] = run(ctx, data, n, 0, want)
        out[name + "_specialised"] = run(ctx, data, n, 1, want)
        out[name + "_general"] = run(ctx, data, n, 2, want)
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
