#!/usr/bin/env python3
"""Generates tests/golden/jpeg_golden_r2b.npz from the REAL reference: streams with the sampling layouts and colour tags the
fused kernels do not take (4:4:0, 4:1:1, 4:1:0, h2v4, h1v4, RGB-tagged, CMYK, YCCK, four-component YCbCr, sub-sampled luma).

Build container only: needs oracle/_ref/libstbref.so (the reference compiled in place by oracle/Makefile).  The streams come from
the test-side writer (tests/support/prog_writer.c) fed with the product's host transform of seeded pictures; what is stored is DATA:
  lay/<k>/jpg      the stream        lay/<k>/hv, lay/<k>/app14   its factors and Adobe transform byte (-1: no APP14)
  lay/<k>/out3     the reference's pixels for req_comp 3
  lay/<k>/fnv      FNV-1a 64 of the reference's pixels for req_comp 0..4
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import image_codecs_amd as ica  # noqa: E402
from test_oracle_golden import LAYOUTS_R2  # noqa: E402


def main():
    ref = helpers.Reference()
    out = {}
    k = 0
    for li, (hv, app14) in enumerate(LAYOUTS_R2):
        for si, (w, h) in enumerate(((64, 48), (36, 20), (30, 17))):
            plan, du = ica.host_transform(ica.synth_rgb(w, h, 900 + 5 * li + si), 92)
            data = helpers.baseline_layout_from_444(plan, du, hv, app14, restart_mcus=(2 if si == 1 else 0))
            key = "lay/%d" % k
            out[key + "/jpg"] = np.frombuffer(data, np.uint8)
            out[key + "/hv"] = np.array(hv, np.int32)
            out[key + "/app14"] = np.array([app14], np.int32)
            fnv = []
            for req in range(5):
                kind, px, _ = ref.load(data, req)
                assert kind == "ok", (hv, app14, w, h, req)
                fnv.append(helpers.fnv1a64(px))
                if req == 3:
                    out[key + "/out3"] = px
            out[key + "/fnv"] = np.array(fnv, np.uint64)
            k += 1
    out["lay/count"] = np.array([k], np.int32)
    np.savez_compressed(os.path.join(HERE, "jpeg_golden_r2b.npz"), **out)
    print("wrote", k, "cases")


if __name__ == "__main__":
    main()
