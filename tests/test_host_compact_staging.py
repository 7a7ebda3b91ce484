"""The host Huffman walk writing COMPACT planes itself (mjh_decode_memory_fmt, jpeg_entropy.c decode_block_c8; round 3): for every
baseline stream the compact region, expanded, equals the int16 tile-layout staging of the same walk (mjh_decode_memory, the planes
rounds 1 and 2 packed on the device with k_pack_c8) -- golden streams of the reference, seeded pictures, streams with coefficients on
both sides of the byte range and at the int16 extremes, restart intervals, multi-scan baseline files, damaged streams (same verdict,
same reason, same planes up to the failure) -- and progressive files stay int16.  CPU only: no GPU call."""
import numpy as np
import pytest

import helpers

EDGE = [127, -128, 128, -129, 255, -256, 256, 1000, -1000, 32767, -32767, 32511, -32640, 383, -384]


def _check_equal(ica, data, req=3, expect_compact=True):
    try:
        d16, arena = ica.HostDecoder.decode(data, req)
        ok16, why16 = True, None
    except ica.MijError as exc:
        ok16, why16 = False, str(exc)
    try:
        dc, region = ica.host_decode_staged(data, req, True)
        okc, whyc = True, None
    except ica.MijError as exc:
        okc, whyc = False, str(exc)
    assert ok16 == okc and why16 == whyc, (why16, whyc)
    if not ok16:
        return None
    if not (dc.flags & 4):
        assert not expect_compact
        n = d16.coef_elems()
        assert np.array_equal(region[:2 * n].view(np.int16), arena[:n])
        return dc
    assert expect_compact
    assert (dc.flags & 1) == (d16.flags & 1), "WIDE_IDCT verdict differs"
    assert dc.color == d16.color
    got = ica.expand_compact_region(dc, region)
    assert np.array_equal(got, arena[:got.size])
    # escape flag of the descriptor == some block carries the flag bit
    offs, main = ica.compact_offsets(dc)
    any_esc = False
    for c, (lo_o, dc_o, hi_o) in enumerate(offs):
        nt = ((dc.comp[c].bw * dc.comp[c].bh) + 63) // 64
        flags = region[lo_o:lo_o + nt * 4096].reshape(nt, 8, 64, 8)[:, 0, :, 0]
        assert np.all(flags <= 1)
        any_esc |= bool(flags.any())
        vals = np.abs(ica.detile_coefficients(d16, arena)[c].astype(np.int32))
    assert any_esc == bool(dc.flags & 8)
    return dc


def test_golden_streams(ica, golden):
    n_compact = n_int16 = 0
    for name in golden.names:
        data = golden.jpg(name)
        for req in (0, 3):
            prog = helpers_is_progressive(data)
            dc = _check_equal(ica, data, req, expect_compact=not prog)
            if dc is not None:
                n_compact += bool(dc.flags & 4)
                n_int16 += not (dc.flags & 4)
    assert n_compact > 40 and n_int16 > 4


def helpers_is_progressive(data):
    i = 2
    while i + 4 <= len(data):
        if data[i] != 0xFF:
            i += 1
            continue
        m = data[i + 1]
        if m in (0xC0, 0xC1, 0xC2):
            return m == 0xC2
        if m == 0xFF or m == 0x01 or 0xD0 <= m <= 0xD9:
            i += 2 if m != 0xFF else 1
            continue
        i += 2 + (data[i + 2] << 8) + data[i + 3]
    return False


def test_escaped_and_extreme_coefficients(ica):
    rng = np.random.default_rng(11)
    seen_escape = 0
    for (w, h, q, layout, rst) in ((96, 64, 90, "native", 5), (72, 40, 95, "native", 0), (80, 48, 95, "422", 0), (64, 56, 95, "grey", 3), (200, 120, 50, "native", 0)):
        for extreme in (False, True):
            img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
            plan, du = ica.host_transform(img, q)
            du = du.copy()
            hit = rng.random(du.shape[0]) < 0.3
            for b in np.nonzero(hit)[0]:
                for _ in range(int(rng.integers(1, 6))):
                    k = int(rng.integers(1, 64))
                    du[b, k] = EDGE[int(rng.integers(0, len(EDGE)))] if extreme else int(rng.integers(128, 400)) * (1 if rng.random() < 0.5 else -1)
            du[:, 0] = np.clip(du[:, 0], -900, 900)
            data = helpers.baseline_from_du(plan, du, restart_mcus=rst, layout=layout)
            dc = _check_equal(ica, data, 3)
            seen_escape += bool(dc.flags & 8)
    assert seen_escape >= 8


def test_seeded_pictures_and_damaged_streams(ica):
    rng = np.random.default_rng(5)
    bases = []
    for i in range(12):
        w, h = int(rng.integers(1, 300)), int(rng.integers(1, 200))
        q = int(rng.choice([30, 75, 90, 95, 100]))
        bases.append(ica.synth_jpeg(w, h, seed=i, quality=q))
        _check_equal(ica, bases[-1], int(rng.choice([0, 1, 3, 4])))
    # non-interleaved baseline (a scan per component) and layouts with other sampling factors
    plan, du = ica.host_transform(ica.synth_rgb(97, 51, 1), 95)
    for hv in ([(1, 2), (1, 1), (1, 1)], [(4, 1), (1, 1), (1, 1)], [(2, 2), (1, 1), (1, 1)], [(1, 1)] * 4):
        _check_equal(ica, helpers.baseline_layout_from_444(plan, du, hv, 0 if len(hv) == 4 else -1, restart_mcus=2), 3)
    for k in range(300):
        d = helpers.mutate(bases[k % len(bases)], 1000 + k, n_mut=1 + k % 4, allow_markers=(k % 3 == 0))
        _check_equal(ica, d, 3)
    for k in range(40):  # truncations
        b = bases[k % len(bases)]
        _check_equal(ica, b[:int(rng.integers(len(b) // 3, len(b)))], 3)


def test_progressive_stays_int16(ica):
    plan, du = ica.host_transform(ica.synth_rgb(80, 48, 2), 92)
    data = helpers.progressive_from_du(plan, du, 1)
    dc, region = ica.host_decode_staged(data, 3, True)
    assert not (dc.flags & 4)
    _check_equal(ica, data, 3, expect_compact=False)
    # and a caller that asks for int16 staging gets it for baseline files too
    d2, region = ica.host_decode_staged(ica.synth_jpeg(64, 48, 1), 3, False)
    assert not (d2.flags & 4)
