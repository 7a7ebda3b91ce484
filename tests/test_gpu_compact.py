"""Compact coefficient planes (include/mij.h, "compact coefficient planes"): the default format of the
coefficients in HBM.  The reference's coefficients are 16-bit (codec/jpeg.c:250-265, :325-365), the planes hold
low bytes plus escape bytes for blocks that need them: these tests put blocks on both sides of the escape -- and
at the extremes of int16 -- through every decode kernel family, through both producers (the GPU Huffman walk,
which writes the format itself, and the host walk, whose int16 staging k_pack_c8 packs on the device), and compare
with the oracle.  No stream may fall back to the host for the size of its coefficients."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

EDGE = [127, -128, 128, -129, 255, -256, 256, 1000, -1000, 32767, -32767, 32511, -32640, 383, -384]


def _streams(ica, seed, extreme):
    """Baseline streams (4:2:0, 4:4:4, 4:2:2, grey) whose quantised coefficients were replaced by values around
    and far beyond the byte range; DC stays small enough for its 11-bit category."""
    rng = np.random.default_rng(seed)
    out = []
    for (w, h, q, layout) in ((96, 64, 90, "native"), (72, 40, 95, "native"), (80, 48, 95, "422"), (64, 56, 95, "grey"), (200, 120, 50, "native")):
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        plan, du = ica.host_transform(img, q)
        du = du.copy()
        nblk = du.shape[0]
        hit = rng.random(nblk) < 0.3  # most blocks stay inside a byte: escaped and plain blocks share tiles and waves
        for b in np.nonzero(hit)[0]:
            for _ in range(int(rng.integers(1, 6))):
                k = int(rng.integers(1, 64))
                if extreme:
                    du[b, k] = EDGE[int(rng.integers(0, len(EDGE)))]
                else:
                    du[b, k] = int(rng.integers(128, 400)) * (1 if rng.random() < 0.5 else -1)
        du[:, 0] = np.clip(du[:, 0], -900, 900)
        out.append(helpers.baseline_from_du(plan, du, restart_mcus=(5 if layout == "native" and w == 96 else 0), layout=layout))
    return out


@pytest.mark.parametrize("extreme", [False, True])
def test_escaped_blocks_decode_exactly_through_both_producers(ica, oracle, gpu_ctx, extreme):
    datas = _streams(ica, 21 + extreme, extreme)
    for req in (3, 4, 1):
        want = []
        for d in datas:
            kind, px, _ = oracle.load(d, req)
            assert kind == "ok", px
            want.append(px)
        # producer 1: the GPU Huffman walk writes compact planes straight into HBM
        b = ica.Batch(gpu_ctx, len(datas), 64 << 20, 64 << 20, 64 << 20)
        b.entropy_reserve(8 << 20)
        ok, slots, reasons = b.decode_jpegs(datas, req, threads=2, gpu_entropy=True)
        assert ok == len(datas), reasons
        b.submit()
        b.wait()
        for i, s in enumerate(slots):
            assert b.slot_coef_bytes(s) == 1
            assert b.slot_escapes(s) > 0, i
            assert np.array_equal(b.fetch(s), want[i]), ("gpu walk", i, req)
        planes_gpu = [b.fetch_coef(s) for s in slots]
        b.close()
        # producer 2: the host walk, which since round 3 stages compact planes itself (mjh_decode_memory_fmt); the k_pack_c8 route is covered below
        b = ica.Batch(gpu_ctx, len(datas), 64 << 20, 64 << 20, 64 << 20)
        ok, slots, reasons = b.decode_jpegs(datas, req, threads=2, gpu_entropy=False)
        assert ok == len(datas), reasons
        b.submit()
        b.wait()
        for i, s in enumerate(slots):
            assert b.slot_coef_bytes(s) == 1
            assert b.slot_escapes(s) > 0, i
            assert np.array_equal(b.fetch(s), want[i]), ("host walk + pack", i, req)
            # the packed planes, expanded again, are the staged int16 planes -- and what the GPU walk wrote
            desc, staged = ica.HostDecoder.decode(datas[i], req)
            for pa, pb, pc in zip(ica.detile_coefficients(desc, b.fetch_coef(s)), ica.detile_coefficients(desc, staged), ica.detile_coefficients(desc, planes_gpu[i])):
                assert np.array_equal(pa, pb) and np.array_equal(pa, pc), i
        # the same batch through the general two-pass kernels, and once more as int16 planes
        b.force_generic(True)
        b.submit()
        b.wait()
        for i, s in enumerate(slots):
            assert b.slot_path(s) == 2 and np.array_equal(b.fetch(s), want[i]), ("two-pass", i, req)
        b.close()
        # and once more as int16 planes (the format is the batch's to ask for before the walk stages anything)
        b = ica.Batch(gpu_ctx, len(datas), 64 << 20, 64 << 20, 64 << 20)
        b.set_coef_format("int16")
        ok, slots, reasons = b.decode_jpegs(datas, req, threads=2, gpu_entropy=False)
        assert ok == len(datas), reasons
        b.submit()
        b.wait()
        for i, s in enumerate(slots):
            assert b.slot_coef_bytes(s) == 0 and np.array_equal(b.fetch(s), want[i]), ("int16", i, req)
        b.close()


def test_escapes_through_stbi_load_and_progressive(ica, oracle, gpu_ctx):
    """stbi_load_from_memory (one-image batch, host walk, pack) and a progressive stream (planes re-staged over ten
    scans, then packed) with coefficients far beyond a byte."""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (88, 120, 3)).astype(np.uint8)
    for q in (92, 100):
        plan, du = ica.host_transform(img, q)
        du = du.copy()
        du[::3, 1] = 300
        du[1::5, 2] = -700
        du[:, 0] = np.clip(du[:, 0], -900, 900)
        for data in (helpers.baseline_from_du(plan, du), helpers.progressive_from_du(plan, du)):
            for req in (0, 3, 4):
                kind, want, _ = oracle.load(data, req)
                assert kind == "ok"
                got = ica.stbi_load_from_memory(data, req)
                assert got is not None, ica.stbi_failure_reason()
                assert np.array_equal(got[0], want), (q, req)


def test_host_staged_compact_planes_equal_the_packed_ones(ica, oracle, gpu_ctx):
    """Round 3: the host walk writes compact planes itself (mjh_decode_memory_fmt) and upload copies them as they are -- no k_pack_c8,
    1.6 instead of 3 bytes per pixel over PCIe.  The planes in HBM equal the ones k_pack_c8 makes from the same walk's int16 staging
    (expanded element for element, escape counts), the pixels equal the oracle's, a batch asked for int16 planes still gets int16
    staging, and a progressive file next to baseline ones still goes through the pack."""
    datas = _streams(ica, 77, True) + [ica.synth_jpeg(1920, 1080, 0), ica.synth_jpeg(333, 211, 3, quality=95)]
    plan, du = ica.host_transform(ica.synth_rgb(120, 88, 4), 92)
    datas.append(helpers.progressive_from_du(plan, du))
    want = [oracle.load(d, 3)[1] for d in datas]
    direct = ica.Batch(gpu_ctx, len(datas), 128 << 20, 128 << 20, 128 << 20)
    packed = ica.Batch(gpu_ctx, len(datas), 128 << 20, 128 << 20, 128 << 20)
    sd = [direct.add_jpeg(d, 3) for d in datas]
    sp = [packed.add_jpeg(d, 3, stage="int16") for d in datas]
    for b in (direct, packed):
        b.submit()
        b.wait()
    for i in range(len(datas)):
        assert direct.slot_coef_bytes(sd[i]) == 1 and packed.slot_coef_bytes(sp[i]) == 1
        assert np.array_equal(direct.fetch(sd[i]), want[i]), ("direct", i)
        assert np.array_equal(packed.fetch(sp[i]), want[i]), ("packed", i)
        assert np.array_equal(direct.fetch_coef(sd[i]), packed.fetch_coef(sp[i])), i
        assert direct.slot_escapes(sd[i]) == packed.slot_escapes(sp[i]), i
    # the flags the walk raised: staged compact for the baseline files, not for the progressive one
    assert [bool(direct.descs[s].flags & 4) for s in sd] == [True] * (len(datas) - 1) + [False]
    # the front end on host threads takes the same route; int16 batches keep int16 staging
    for fmt in ("compact", "int16"):
        b = ica.Batch(gpu_ctx, len(datas), 128 << 20, 128 << 20, 128 << 20)
        b.set_coef_format(fmt)
        ok, slots, reasons = b.decode_jpegs(datas, 3, threads=3, gpu_entropy=False)
        assert ok == len(datas), reasons
        b.submit()
        b.wait()
        for i, s in enumerate(slots):
            assert b.slot_coef_bytes(s) == (1 if fmt == "compact" else 0)
            assert np.array_equal(b.fetch(s), want[i]), (fmt, i)
        b.close()
    direct.close()
    packed.close()


def test_progressive_l1_bound_is_taken_by_the_pack_kernel(ica, oracle, gpu_ctx):
    """Round 3: for a progressive file staged for a compact batch the host no longer makes a pass over the finished planes for the per-block
    L1 bound behind MIJ_FLAG_WIDE_IDCT (MIJ_FLAG_L1_ON_DEVICE): k_pack_c8 takes it while it packs and mij_batch_upload raises the flag.  The
    verdict equals the host's own (mjh_decode_memory still computes it), for tame streams, streams just around the limit and wild ones; pixels
    equal the oracle's either way; clones follow their source."""
    rng = np.random.default_rng(9)
    datas = []
    for (w, h, q, boost) in ((120, 88, 92, 0), (96, 64, 100, 0), (64, 64, 95, 300), (200, 120, 90, 1000), (72, 40, 100, 32767), (40, 24, 75, 40)):
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        plan, du = ica.host_transform(img, q)
        du = du.copy()
        if boost:
            du[::3, 1] = boost
            du[1::5, 5] = -boost
            du[:, 0] = np.clip(du[:, 0], -900, 900)
        datas.append(helpers.progressive_from_du(plan, du, 1 if w != 96 else 2))
    host_wide = [bool(ica.HostDecoder.decode(d, 3)[0].flags & 1) for d in datas]
    assert any(host_wide) and not all(host_wide)
    b = ica.Batch(gpu_ctx, 2 * len(datas), 64 << 20, 64 << 20, 64 << 20)
    slots = [b.add_jpeg(d, 3) for d in datas]
    assert all(b.slot_flags(s) & 16 for s in slots), "the walk did not leave the L1 bound to the device"
    clones = [b.add_clone(s) for s in slots]
    b.submit()
    b.wait()
    for i, s in enumerate(slots):
        f = b.slot_flags(s)
        assert not (f & 16) and bool(f & 1) == host_wide[i], (i, f, host_wide[i])
        kind, want, _ = oracle.load(datas[i], 3)
        assert np.array_equal(b.fetch(s), want), i
        assert np.array_equal(b.fetch(clones[i]), want), ("clone", i)
    # a second upload (another kernel family) leaves the packed planes alone
    b.force_generic(True)
    b.submit()
    b.wait()
    for i, s in enumerate(slots):
        assert np.array_equal(b.fetch(s), oracle.load(datas[i], 3)[1]), ("two-pass", i)
    b.close()
    # the front end on host threads and stbi_load take the same route
    b = ica.Batch(gpu_ctx, len(datas), 64 << 20, 64 << 20, 64 << 20)
    ok, slots, reasons = b.decode_jpegs(datas, 3, threads=3, gpu_entropy=False)
    assert ok == len(datas), reasons
    b.submit()
    b.wait()
    for i, s in enumerate(slots):
        assert bool(b.slot_flags(s) & 1) == host_wide[i] and np.array_equal(b.fetch(s), oracle.load(datas[i], 3)[1]), i
    b.close()
    for i, d in enumerate(datas):
        got = ica.stbi_load_from_memory(d, 3)
        assert got is not None and np.array_equal(got[0], oracle.load(d, 3)[1]), i
