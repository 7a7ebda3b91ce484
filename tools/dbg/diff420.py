"""debug helper: decode one 4:2:0 image through the batch API and describe where it differs from the oracle"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import image_codecs_amd as ica
import helpers
o = helpers.Oracle()
ctx = ica.Context()
for (w, h, seed) in ((200, 150, 3), (640, 360, 6), (1920, 1080, 1)):
    d = ica.synth_jpeg(w, h, seed)
    want = o.load(d, 3)[1]
    for rows in (os.environ.get("ROWS", "1,1000").split(",")):
        os.environ["MIJ_BAND_ROWS"] = rows
        b = ica.Batch(ctx, 1, 64 << 20, 64 << 20, 64 << 20)
        ok, slots, reasons = b.decode_jpegs([d], 3, threads=1, gpu_entropy=False)
        b.submit(); b.wait()
        got = b.fetch(slots[0])
        diff = (got != want)
        print(w, h, "rows", rows, "path", b.slot_path(slots[0]), "ndiff px", int(diff.any(axis=2).sum()), "per channel", diff.sum(axis=(0, 1)).tolist())
        if diff.any():
            ys, xs = np.nonzero(diff.any(axis=2))
            print("  rows with diffs:", np.unique(ys)[:40].tolist(), "... n", len(np.unique(ys)))
            print("  cols with diffs: min", xs.min(), "max", xs.max(), "unique cols//8:", np.unique(xs // 16)[:50].tolist())
        b.close()
