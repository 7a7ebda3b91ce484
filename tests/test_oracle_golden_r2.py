"""CPU: the oracle against the round-2 golden vectors of the real reference (late JFIF / Adobe markers, config 1),
and the host half of the product on the same inputs (colour mode decided after the last marker, codec/jpeg.c:2244)."""
import numpy as np
import pytest

import helpers


@pytest.fixture(scope="module")
def g2():
    return helpers.GoldenR2()


def test_oracle_follows_late_markers(oracle, g2):
    for name in g2.late_names:
        for req in range(5):
            data, kind, want = g2.late(name, req)
            got = oracle.load(data, req)
            assert got[0] == kind, (name, req, got[1] if got[0] == "fail" else None)
            if kind == "ok":
                assert np.array_equal(got[1], want), (name, req)
            else:
                assert got[1] == want, (name, req)


def test_oracle_config1(oracle, g2):
    data = bytes(g2["cfg1/jpg"])
    for req in (0, 1, 3, 4):
        kind, px, comp = oracle.load(data, req)
        assert kind == "ok" and px.shape[:2] == (512, 512)
        assert np.array_equal(px[:8], g2["cfg1/head%d" % req])
        assert helpers.fnv1a64(px) == int(g2["cfg1/fnv%d" % req][0]), req


def test_host_stage_colour_mode_after_late_markers(ica, g2):
    """mjh_probe_memory sees only the markers in front of SOF; mjh_decode_memory must report the colour branch the
    reference takes once every marker has been seen.  Moving a marker behind SOF must not change the decision."""
    color = {"adobe_rgb_after_sof": 2, "jfif_then_adobe0_after_sof": 1, "nojfif_adobe0_after_sof": 2, "adobe0_then_jfif_after_sof": 1,
             "cmyk_adobe2_after_sof": 4, "cmyk_adobe0_after_sof": 3, "b420_adobe0_after_sof": 2, "prog_adobe0_between_scans": 2}
    for name in g2.late_names:
        data, kind, _ = g2.late(name, 3)
        assert kind == "ok"
        desc, _ = ica.HostDecoder.decode(data, 3)
        assert desc.color == color[name], (name, desc.color)


def _layout_cases():
    import os
    z = np.load(os.path.join(helpers.ROOT, "tests", "golden", "jpeg_golden_r2b.npz"))
    return z, int(z["lay/count"][0])


def test_oracle_sampling_layout_vectors(oracle):
    """4:4:0, 4:1:1, 4:1:0, h2v4, h1v4, RGB-tagged, CMYK, YCCK, four-component YCbCr, sub-sampled luma: the reference's own
    outputs (tests/golden/make_golden_r2b.py), every req_comp."""
    z, n = _layout_cases()
    assert n == 48
    for k in range(n):
        data = bytes(z["lay/%d/jpg" % k])
        for req in range(5):
            kind, px, _ = oracle.load(data, req)
            assert kind == "ok", (k, req)
            assert helpers.fnv1a64(px) == int(z["lay/%d/fnv" % k][req]), (k, req, z["lay/%d/hv" % k].tolist())
            if req == 3:
                assert np.array_equal(px, z["lay/%d/out3" % k]), k


@pytest.mark.gpu
def test_gpu_sampling_layout_vectors(ica, gpu_ctx):
    """The same vectors through stbi_load_from_memory on the GPU (default kernel choice per layout)."""
    z, n = _layout_cases()
    for k in range(n):
        data = bytes(z["lay/%d/jpg" % k])
        for req in range(5):
            got = ica.stbi_load_from_memory(data, req)
            assert got is not None, (k, req, ica.stbi_failure_reason())
            assert helpers.fnv1a64(got[0]) == int(z["lay/%d/fnv" % k][req]), (k, req, z["lay/%d/hv" % k].tolist())
            if req == 3:
                assert np.array_equal(got[0], z["lay/%d/out3" % k]), k
