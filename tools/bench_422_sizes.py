"""Kernel-resident timing of the fused 4:2:2 path by picture width (k_fused422 / w / x), batches of ~0.53 Gpix resident in HBM."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import image_codecs_amd as ica  # noqa: E402
import helpers  # noqa: E402
import bench_sizes  # noqa: E402

ctx = ica.Context()
oracle = helpers.Oracle()
for (w, h) in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(1920, 1080), (3840, 2160), (6000, 4000)]:
    plan, du = ica.host_transform(ica.synth_rgb(w, h, 0), 91)
    data = helpers.baseline_layout_from_444(plan, du, [(2, 1), (1, 1), (1, 1)], -1)
    n = max(2, int(256 * 1920 * 1080 / (w * h)))
    print("%dx%d 4:2:2" % (w, h), bench_sizes.run(ctx, data, n, oracle.load(data, 3)[1]), flush=True)
ctx.close()
