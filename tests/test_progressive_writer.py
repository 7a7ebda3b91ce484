"""The progressive test-stream writer (tests/support/prog_writer.c) and the host progressive stage.

The reference holds no progressive files and no progressive writer, and libjpeg is absent from the GPU
box, so config 4's full-size inputs are made by re-emitting the baseline writer's quantised data units
as SOF2 scans.  Self-check: the same coefficients must come back out of the progressive stream (host
stage planes identical to the baseline stream's) and the oracle must decode both to the same pixels;
where the real reference is built (this container) it is asked too."""
import numpy as np
import pytest

import helpers

CASES = [(1, 1, 90), (8, 8, 95), (17, 33, 90), (67, 45, 95), (130, 70, 92), (200, 133, 50)]


@pytest.mark.parametrize("script", [0, 1])
def test_progressive_stream_carries_the_same_coefficients(ica, oracle, script):
    ref = helpers.Reference() if helpers.Reference.available() else None
    for (w, h, q) in CASES:
        img = ica.synth_rgb(w, h, seed=w + h)
        plan, du = ica.host_transform(img, q)
        base = ica.emit_jpeg(plan, du)
        prog = helpers.progressive_from_du(plan, du, script)
        assert prog[:2] == b"\xff\xd8" and b"\xff\xc2" in prog[:300]
        kind_b, want, _ = oracle.load(base, 3)
        kind_p, got, _ = oracle.load(prog, 3)
        assert kind_b == kind_p == "ok"
        assert np.array_equal(got, want), (w, h, q)
        if ref is not None:
            assert np.array_equal(ref.load(prog, 3)[1], want), (w, h, q)
        d_b, a_b = ica.HostDecoder.decode(base, 3)
        d_p, a_p = ica.HostDecoder.decode(prog, 3)
        assert d_b.flags == d_p.flags
        # non-interleaved AC scans visit ceil(x/8) x ceil(y/8) blocks only (codec/jpeg.c:1272-1273): the
        # padding blocks of the MCU grid carry AC data in the baseline stream alone, and nothing reads them
        for ci, (p_b, p_p) in enumerate(zip(ica.detile_coefficients(d_b, a_b), ica.detile_coefficients(d_p, a_p))):
            cw, ch = (d_b.comp[ci].x + 7) >> 3, (d_b.comp[ci].y + 7) >> 3
            assert np.array_equal(p_b[:ch, :cw], p_p[:ch, :cw]), (w, h, q, ci)
            assert np.array_equal(p_b[:, :, 0, 0], p_p[:, :, 0, 0]), (w, h, q, ci)


def test_progressive_refinement_scans_use_eob_runs(ica):
    """The point of per-scan optimal tables: EOBn symbols (run<<4, run in 1..14) must occur, otherwise the
    host stage's eob_run bookkeeping (codec/jpeg.c:470-493, :520-533) is never exercised at scale."""
    plan, du = ica.host_transform(ica.synth_rgb(256, 256, 2), 95)
    prog = helpers.progressive_from_du(plan, du, 1)
    pos, found = 0, False
    while True:
        pos = prog.find(b"\xff\xc4", pos)
        if pos < 0:
            break
        length = int.from_bytes(prog[pos + 2:pos + 4], "big")
        tc_th = prog[pos + 4]
        vals = prog[pos + 5 + 16:pos + 2 + length]
        if tc_th >> 4 == 1 and any((v & 15) == 0 and 0 < (v >> 4) < 15 for v in vals):
            found = True
        pos += 2 + length
    assert found


def test_writer_streams_pin_the_oracle_to_the_live_reference(ica, oracle):
    """Build container only (the reference cannot travel): every kind of stream the test-side writers make --
    progressive scripts, 4:2:2, grey, baseline with optimal tables and restart intervals -- decodes in the real
    reference exactly as in the oracle, for every req_comp.  This is what lets the GPU box trust the oracle on them."""
    if not helpers.Reference.available():
        pytest.skip("the reference build exists only where /root/reference does")
    ref = helpers.Reference()
    rng = np.random.default_rng(5)
    n = 0
    for k in range(36):
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 160))
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8) if k % 2 else ica.synth_rgb(w, h, k)
        plan, du = ica.host_transform(img, int(rng.choice([30, 90, 95, 100])) if k % 3 else 95)
        kind = k % 6
        if kind == 0:
            data = helpers.progressive_from_du(plan, du, k & 1)
        elif kind == 1 and plan.du_per_mcu == 3:
            data = helpers.progressive_422_from_444(plan, du, k & 1)
        elif kind == 2 and plan.du_per_mcu == 3:
            data = helpers.progressive_grey_from_444(plan, du, k & 1)
        elif kind == 3:
            data = helpers.baseline_from_du(plan, du, int(rng.choice([0, 1, 3, 50])), "native")
        elif kind == 4 and plan.du_per_mcu == 3:
            data = helpers.baseline_from_du(plan, du, int(rng.choice([1, 2, 7])), "422")
        elif plan.du_per_mcu == 3:
            data = helpers.baseline_from_du(plan, du, int(rng.choice([0, 2])), "grey")
        else:
            data = helpers.baseline_from_du(plan, du, 5, "native")
        for req in (0, 1, 3, 4):
            ko, po, _ = oracle.load(data, req)
            kr, pr, _ = ref.load(data, req)
            assert ko == kr == "ok", (k, req)
            assert np.array_equal(po, pr), (k, req)
            n += 1
    assert n == 36 * 4
