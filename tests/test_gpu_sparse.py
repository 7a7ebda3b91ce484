"""Sparse-block transforms of the decode kernels (mij_kernels.h, "sparse blocks"; round 3): a wavefront whose 64 blocks are all
DC-only / inside the top-left 2x2 / inside the 4x4 takes a reduced IDCT -- the reference's own zero-column shortcut
(codec/jpeg.c:625-633) taken per block class instead of per column.  These streams put uniform wavefronts of every class, wavefronts
that mix classes (they must take the widest one), escaped blocks (always the full transform) and partial wavefronts through the band
kernels in BOTH plane formats and from both producers; pixels against the oracle, bit for bit, and the class counters must show that
every path actually ran."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

# natural (row-major) position of every zigzag index: units are stored in zigzag order
NAT_OF_ZZ = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57,
                      50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])
ROW, COL = NAT_OF_ZZ // 8, NAT_OF_ZZ % 8
KEEP = {0: (ROW == 0) & (COL == 0), 1: (ROW < 2) & (COL < 2), 2: (ROW < 4) & (COL < 4), 3: np.ones(64, bool)}


def _classed_stream(ica, w, h, seed, q, layout, row_classes, escapes=False):
    """4:2:0 (or `layout`) baseline stream of a noise picture whose blocks were cut down to a class per MCU row:
    row_classes[comp][mcu_row % len] in {0, 1, 2, 3, 'mix'}; 'mix' draws a class per block.  Cut blocks keep a non-zero at the far corner of
    their class (so that a 4x4 block really needs the 4x4 transform)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    plan, du = ica.host_transform(img, q)
    du = du.copy()
    per = plan.du_per_mcu
    nlum = per - 2
    for m in range(plan.mcu_x * plan.mcu_y):
        my = m // plan.mcu_x
        for j in range(per):
            comp = 0 if j < nlum else (1 if j == nlum else 2)
            rc = row_classes[comp][my % len(row_classes[comp])]
            cls = int(rng.integers(0, 4)) if rc == "mix" else rc
            b = m * per + j
            du[b, ~KEEP[cls]] = 0
            if cls in (1, 2):
                far = int(np.nonzero((ROW == (1 if cls == 1 else 3)) & (COL == (1 if cls == 1 else 3)))[0][0])
                du[b, far] = 3 if (b & 1) else -2
            if escapes and cls != 0 and rng.random() < 0.05:
                du[b, 1] = 300  # a coefficient beyond a byte: escaped block, class 3 whatever its extent
    du[:, 0] = np.clip(du[:, 0], -900, 900)
    return helpers.baseline_from_du(plan, du, layout=layout)


CASES = [
    # 1040 px: 65 MCU columns -> chroma wavefronts of 64 + 1 lanes, luma 130 blocks per block row
    dict(w=1040, h=160, q=90, layout="native", rows=[[3, 0, 1, 2, "mix", 3, 0, 2, 1, "mix"], [0, 1, 2, 3, "mix", 0, 0, 1, 2, 3], [1, 1, 0, 2, "mix", 3, 2, 0, 0, 1]]),
    dict(w=1024, h=96, q=75, layout="native", rows=[[0, 1, 2, "mix", 3, 0], [0, 0, 1, 2, "mix", 3], [2, 1, 0, 0, 3, "mix"]]),
    dict(w=1920, h=64, q=90, layout="native", rows=[[2, 0, 1, 3], [0, 1, 2, 3], [1, 2, 0, 0]]),
    dict(w=333, h=80, q=50, layout="native", rows=[["mix", 0, 1, 2, 3], [0, "mix", 2, 1, 0], [1, 0, "mix", 2, 2]]),
    # the other kernel families: 4:4:4 (k_fused444 keeps the full transform -- it is HBM-bound -- but must of course decode these), 4:2:2 (k_fused422), grey (k_fused_grey)
    dict(w=1040, h=80, q=95, layout="native", path=3, rows=[[3, 0, 1, 2, "mix", 3, 0, 2, 1, "mix"], [0, 1, 2, 3, "mix", 0, 0, 1, 2, 3], [1, 1, 0, 2, "mix", 3, 2, 0, 0, 1]]),
    dict(w=2064, h=72, q=95, layout="422", path=4, rows=[[0, 1, 2, 3, "mix", 3, 0, 2, 1], [0, 1, 2, 3, "mix", 0, 0, 1, 2], [1, 1, 0, 2, "mix", 3, 2, 0, 0]]),
    dict(w=1040, h=80, q=92, layout="grey", path=5, rows=[[3, 0, 1, 2, "mix", 3, 0, 2, 1, "mix"], [0], [0]]),
]


@pytest.mark.parametrize("escapes", [False, True])
def test_every_sparse_class_in_both_plane_formats_and_from_both_producers(ica, oracle, gpu_ctx, escapes):
    datas = [_classed_stream(ica, c["w"], c["h"], 40 + i, c["q"], c["layout"], c["rows"], escapes) for i, c in enumerate(CASES)]
    for req in (3, 4):
        want = []
        for d in datas:
            kind, px, _ = oracle.load(d, req)
            assert kind == "ok", px
            want.append(px)
        for gpu_walk, fmt in ((False, "compact"), (False, "int16"), (True, "compact")):
            b = ica.Batch(gpu_ctx, len(datas), 64 << 20, 64 << 20, 64 << 20)
            b.set_coef_format(fmt)  # before the host walk: it stages the format the batch asks for
            if gpu_walk:
                b.entropy_reserve(16 << 20)
            ok, slots, reasons = b.decode_jpegs(datas, req, threads=2, gpu_entropy=gpu_walk)
            assert ok == len(datas), reasons
            if True:
                b.upload()
                assert {b.slot_coef_bytes(s) for s in slots} == {1 if fmt == "compact" else 0}
                b.count_idct_classes(True)
                b.launch()
                b.wait()
                counts = b.idct_class_counts()
                b.count_idct_classes(False)
                assert [b.slot_path(s) for s in slots] == [c.get("path", 1) for c in CASES], "a picture did not take the kernel family it was built for"
                assert all(v > 0 for v in counts), ("a sparse class never ran", fmt, gpu_walk, counts)
                for i, s in enumerate(slots):
                    assert np.array_equal(b.fetch(s), want[i]), (i, req, fmt, "gpu walk" if gpu_walk else "host walk")
                # one more launch without counting: the same pixels
                b.launch()
                b.wait()
                assert np.array_equal(b.fetch(slots[0]), want[0])
                # and through the two-pass family (k_idct_planes takes the same sparse transforms)
                b.force_generic(True)
                b.upload()
                b.count_idct_classes(True)
                b.launch()
                b.wait()
                counts = b.idct_class_counts()
                b.count_idct_classes(False)
                assert all(v > 0 for v in counts), ("two-pass", counts)
                for i, s in enumerate(slots):
                    assert b.slot_path(s) == 2 and np.array_equal(b.fetch(s), want[i]), ("two-pass", i, req, fmt)
                b.force_generic(False)
            b.close()


def test_class_counters_follow_the_content(ica, oracle, gpu_ctx):
    """A picture whose chroma is flat and whose luma is noise: every chroma wavefront DC-only, every luma wavefront full; and the bench
    picture (synth_rgb): Cb mostly DC-only, Cr inside the 2x2, luma full."""
    rng = np.random.default_rng(3)
    grey = np.repeat(rng.integers(0, 256, (128, 1024, 1)).astype(np.uint8), 3, axis=2)
    for img, expect in ((grey, "flat chroma"), (ica.synth_rgb(1920, 1080, 0), "bench")):
        data = ica.stbi_write_jpg_to_memory(img, 90)
        kind, want, _ = oracle.load(data, 3)
        b = ica.Batch(gpu_ctx, 1, 64 << 20, 64 << 20, 64 << 20)
        ok, slots, reasons = b.decode_jpegs([data], 3, threads=1, gpu_entropy=False)
        assert ok == 1, reasons
        b.upload()
        b.count_idct_classes(True)
        b.launch()
        b.wait()
        counts = b.idct_class_counts()
        assert np.array_equal(b.fetch(slots[0]), want)
        b.close()
        total = sum(counts)
        if expect == "flat chroma":
            # 8 MCU rows, one band or more: per MCU row 4 luma wavefronts (full) and 2 chroma wavefronts (DC only), halo rows aside
            assert counts[3] >= 4 * 8 and counts[0] >= 2 * 8 and counts[1] == 0 and counts[2] == 0, counts
        else:
            assert counts[3] > 0.6 * total and counts[0] > 0.1 * total and counts[1] + counts[2] > 0.1 * total, counts
