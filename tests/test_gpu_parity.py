"""Parity tests proper (-m gpu): the HIP path, called through the C-ABI, against
  (1) the golden vectors the real reference produced (tests/golden),
  (2) the oracle (oracle/) on seeded inputs at sizes it finishes in seconds,
  (3) size-independent properties at BASELINE.json's full size (1080p batches).
Bit-exact everywhere: this is integer/byte work."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

# streams on which the reference reads never-written (uninitialised) planes: pixels are compared
# with the oracle's definition instead (undecoded block == all-zero coefficients)
UNINIT_IN_REFERENCE = {"dri_without_rst"}


def test_native_library_is_the_one_running(ica, gpu_ctx):
    arch, cus, mem = gpu_ctx.info()
    assert arch.startswith("gfx950"), arch
    assert cus == 256
    loaded = [line for line in open("/proc/self/maps") if "libimagecodecs_mi355x.so" in line]
    assert loaded, "the in-tree HIP library is not mapped into this process"


def test_stbi_load_matches_reference_golden(golden, ica, oracle, gpu_ctx):
    n = 0
    for name in golden.names:
        data = golden.jpg(name)
        for req in range(5):
            kind, want = golden.expect(name, req)
            if kind == "skip":
                n += 1
                continue
            got = ica.stbi_load_from_memory(data, req)
            if kind == "fail":
                assert got is None, (name, req)
                assert ica.stbi_failure_reason() == want, (name, req, ica.stbi_failure_reason(), want)
            else:
                assert got is not None, (name, req, ica.stbi_failure_reason())
                pixels, w, h, comp = got
                if name in UNINIT_IN_REFERENCE:
                    want = oracle.load(data, req)[1]
                assert pixels.shape == want.shape, (name, req)
                assert np.array_equal(pixels, want), (name, req, int((pixels != want).sum()))
                assert comp == int(golden[name + "/info"][3])
            n += 1
    assert n == 5 * len(golden.names)


def test_bad_req_comp(golden, ica, gpu_ctx):
    assert ica.stbi_load_from_memory(golden.jpg("b420_64x64_q90"), 5) is None
    assert ica.stbi_failure_reason() == "bad req_comp"
    assert ica.stbi_load_from_memory(golden.jpg("b420_64x64_q90"), -1) is None


def test_flip_and_16bit(golden, ica, gpu_ctx):
    data = golden.jpg("b420_31x47_q60")
    base = ica.stbi_load_from_memory(data, 3)[0]
    ica.stbi_set_flip_vertically_on_load(1)
    try:
        assert np.array_equal(ica.stbi_load_from_memory(data, 3)[0], base[::-1])
    finally:
        ica.stbi_set_flip_vertically_on_load(0)
    wide = ica.stbi_load_16_from_memory(data, 3)[0]
    assert np.array_equal(wide, base.astype(np.uint16) * 257)


def test_stbi_load_from_file(golden, ica, gpu_ctx, tmp_path):
    p = tmp_path / "x.jpg"
    p.write_bytes(golden.jpg("b422_37x21"))
    got = ica.stbi_load(str(p), 0)
    assert np.array_equal(got[0], golden.expect("b422_37x21", 0)[1])
    assert ica.stbi_load(str(tmp_path / "missing.jpg"), 0) is None
    assert ica.stbi_failure_reason() == "can't fopen"


def _batch_for(ica, ctx, datas, req, clones=0):
    descs = [ica.HostDecoder.probe(d, req) for d in datas]
    cb = sum(ica.Batch.coef_bytes(d) for d in descs)
    ob = sum(ica.Batch.out_bytes(d) for d in descs)
    b = ica.Batch(ctx, len(datas) * (1 + clones), cb, cb * (1 + clones), ob * (1 + clones))
    slots = [b.add_jpeg(d, req) for d in datas]
    for _ in range(clones):
        for s in list(slots):
            b.add_clone(s)
    return b, slots


def test_seeded_images_vs_oracle_fused_and_generic(ica, oracle, gpu_ctx):
    """Random sizes/qualities; every image through BOTH kernel families, both equal to the oracle."""
    rng = np.random.default_rng(11)
    datas, wants = [], []
    for i in range(48):
        w, h = int(rng.integers(1, 400)), int(rng.integers(1, 300))
        q = int(rng.choice([10, 50, 75, 90, 90, 90, 95]))
        img = ica.synth_rgb(w, h, seed=i) if i % 3 else rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        datas.append(ica.stbi_write_jpg_to_memory(img, q))
    for req in (3, 4):
        wants = [oracle.load(d, req)[1] for d in datas]
        for generic in (0, 1, 2):
            b, slots = _batch_for(ica, gpu_ctx, datas, req)
            b.force_generic(generic)
            b.submit()
            b.wait()
            paths = set()
            for s, want in zip(slots, wants):
                assert np.array_equal(b.fetch(s), want), (s, req, generic)
                paths.add(b.slot_path(s))
            assert paths == ({2} if generic else {1, 3}), paths  # q>90 files are 4:4:4 -> the register-resident kernel
            b.close()


def test_band_height_does_not_change_pixels(ica, oracle, gpu_ctx, monkeypatch):
    """Fused kernel with 1, 2, 3 and 'all' MCU rows per workgroup: halo handling is exact."""
    datas = [ica.synth_jpeg(200, 150, 3), ica.synth_jpeg(16, 16, 4), ica.synth_jpeg(33, 97, 5), ica.synth_jpeg(640, 360, 6)]
    wants = [oracle.load(d, 3)[1] for d in datas]
    for rows in ("1", "2", "3", "1000"):
        monkeypatch.setenv("MIJ_BAND_ROWS", rows)
        b, slots = _batch_for(ica, gpu_ctx, datas, 3)
        b.submit()
        b.wait()
        for s, want in zip(slots, wants):
            assert b.slot_path(s) == 1
            assert np.array_equal(b.fetch(s), want), (rows, s)
        b.close()


def test_wide_idct_path_is_exact(golden, ica, oracle, gpu_ctx):
    """Streams whose first IDCT pass overflows int16: the exact 32-bit second pass must equal the
    reference's wrapping int arithmetic (oracle), in both kernel families."""
    hits = 0
    for name in ("b420_64x64_q90", "b444_40x24_q95", "pil420_130x50"):
        data = bytearray(golden.jpg(name))
        i = bytes(data).index(b"\xff\xdb")
        rng = np.random.default_rng(len(name))
        for k in range(64):
            data[i + 5 + k] = int(rng.integers(100, 256))
        data = bytes(data)
        d, _ = ica.HostDecoder.decode(data, 3)
        hits += d.flags & 1
        want = oracle.load(data, 3)[1]
        assert np.array_equal(ica.stbi_load_from_memory(data, 3)[0], want), name
        b, slots = _batch_for(ica, gpu_ctx, [data], 3)
        b.force_generic(True)
        b.submit()
        assert np.array_equal(b.fetch(slots[0]), want), name
        b.close()
    assert hits >= 2


def test_fuzzed_streams_vs_oracle(golden, ica, oracle, gpu_ctx):
    """Mutated entropy data through the whole GPU path: same accept/reject and reason as the oracle,
    same pixels (including blocks the scan never reached, defined as zero coefficients)."""
    n_ok = n_fail = 0
    for name in ("b420_64x64_q90", "b444_40x24_q95", "grey_33x20", "b422_37x21", "prog_420_64x64", "rst_blocks_64x48", "s41_35x19"):
        base = golden.jpg(name)
        for seed in range(30):
            data = helpers.mutate(base, seed * 31337 + len(name), allow_markers=(seed % 3 == 0))
            o = oracle.load(data, 3)
            got = ica.stbi_load_from_memory(data, 3)
            if o[0] == "fail":
                assert got is None, (name, seed)
                assert ica.stbi_failure_reason() == o[1], (name, seed)
                n_fail += 1
            else:
                assert got is not None, (name, seed, ica.stbi_failure_reason())
                assert np.array_equal(got[0], o[1]), (name, seed)
                n_ok += 1
    assert n_ok > 100 and n_fail > 3


def test_full_size_1080p_batch_properties(ica, oracle, gpu_ctx):
    """BASELINE configuration at full image size: 1920x1080 4:2:0 q=90.
    (a) four distinct images equal the oracle byte for byte, (b) 8 clones of each: every clone's
    hash equals its source's (independent buffers, identical results), (c) re-launching is
    idempotent, (d) fused == two-pass."""
    datas = [ica.synth_jpeg(1920, 1080, s, 90) for s in range(4)]
    wants = [oracle.load(d, 3)[1] for d in datas]
    b, slots = _batch_for(ica, gpu_ctx, datas, 3, clones=8)
    b.submit()
    b.wait()
    n = len(datas)
    for s in range(n):
        assert b.slot_path(s) == 1
        assert np.array_equal(b.fetch(s), wants[s]), s
    hashes = [b.hash_out(s) for s in range(n * 9)]
    for s in range(n * 9):
        assert hashes[s] == hashes[s % n], s
    b.launch()
    b.wait()
    assert [b.hash_out(s) for s in range(n)] == hashes[:n]
    b.close()
    b2, slots2 = _batch_for(ica, gpu_ctx, datas[:2], 3)
    b2.force_generic(True)
    b2.submit()
    for s in slots2:
        assert np.array_equal(b2.fetch(s), wants[s])
    b2.close()


def test_wide_images_take_the_band_kernel_in_column_segments(ica, oracle, gpu_ctx, monkeypatch):
    """A row of 4:2:0 MCUs beyond 5840 pixels (4:4:0: 4300) does not fit the LDS of a CU: since round 3 such pictures are cut into column
    segments, each transformed with one MCU column of halo on either side (k_fused420c / k_fused440c, fused_band SEG), instead of falling to
    the two-pass kernels.  Widths on both sides of the switch and of the segment counts, odd and unaligned widths (the careful strips of the
    last segment), pictures a few rows high and several bands high, both output widths, both plane formats, both producers: all equal to
    the CPU checker, all on the fused path."""
    import helpers
    datas = []
    for w, h, q in ((5840, 40, 90), (5841, 33, 90), (5856, 40, 90), (6000, 40, 90), (8191, 50, 85), (8192, 17, 90), (11615, 36, 90), (11616, 16, 80), (11632, 48, 90), (16385, 20, 90)):
        datas.append(ica.synth_jpeg(w, h, w & 7, q))  # 4:2:0 (the writer's layout up to quality 90)
    for w, h in ((4296, 40), (4312, 33), (6001, 24), (9000, 40)):  # 4:4:0: 304 B per 8 pixels, 538 MCU columns fit
        plan, du = ica.host_transform(ica.synth_rgb(w, h, w & 7), 92)
        datas.append(helpers.baseline_layout_from_444(plan, du, [(1, 2), (1, 1), (1, 1)], -1))
    for req in (3, 4):
        wants = [oracle.load(d, req)[1] for d in datas]
        for band_rows, fmt, gpu_walk in ((0, "compact", False), (1, "int16", False), (0, "compact", True)):
            if band_rows:
                monkeypatch.setenv("MIJ_BAND_ROWS", str(band_rows))
            else:
                monkeypatch.delenv("MIJ_BAND_ROWS", raising=False)
            b = ica.Batch(gpu_ctx, len(datas), 96 << 20, 96 << 20, 96 << 20)
            b.set_coef_format(fmt)
            if gpu_walk:
                b.entropy_reserve(32 << 20)
            ok, slots, reasons = b.decode_jpegs(datas, req, threads=2, gpu_entropy=gpu_walk)
            assert ok == len(datas), reasons
            b.submit()
            for i, (s, want) in enumerate(zip(slots, wants)):
                assert b.slot_path(s) in (1, 6), (i, b.slot_path(s))
                got = b.fetch(s)
                assert np.array_equal(got, want), (i, want.shape, req, band_rows, fmt, gpu_walk, int((got != want).sum()), np.argwhere((got != want).any(axis=2))[:4].tolist())
            b.close()


def test_band_kernels_by_workgroups_per_cu(ica, oracle, gpu_ctx, monkeypatch):
    """The band kernels come with one, two, four, eight or sixteen waves per workgroup, chosen by the picture's width (narrow rows do
    not fill four waves; of wide ones only two or one workgroups fit a CU's LDS: k_fused420t / s / - / w / x, k_fused440 / w): widths on both sides of each switch,
    odd sizes, both output widths, one band and many, mixed in one batch -- all equal to the CPU checker."""
    import helpers
    datas = []
    for w in (1904, 1920, 2288, 2304, 2320, 2848, 2864, 3840, 4097):  # 4:2:0: 448 B per 16 pixels; 3 x fit up to 1904, 2 x up to 2848
        datas.append(ica.synth_jpeg(w, 70, w & 7, 90))
    for w, h in ((1, 1), (17, 9), (383, 40), (384, 33), (385, 50), (400, 16), (895, 30), (896, 47), (897, 20), (912, 64)):  # one wave up to 24 MCU columns, two up to 56
        datas.append(ica.synth_jpeg(w, h, (w + h) & 7, 85))
    plan, du = ica.host_transform(ica.synth_rgb(2160, 64, 3), 95)  # 4:4:0 layouts: 304 B per 8 pixels; 3 x fit up to 1432, 2 x up to 2152
    plan2, du2 = ica.host_transform(ica.synth_rgb(1424, 64, 4), 95)
    datas.append(helpers.baseline_layout_from_444(plan, du, [(1, 2), (1, 1), (1, 1)], -1))
    datas.append(helpers.baseline_layout_from_444(plan2, du2, [(1, 2), (1, 1), (1, 1)], -1))
    for w in (3392, 3424, 5104, 5136):  # 4:2:2: 256 B per 16 pixels + 16; three workgroups fit up to 3408 pixels, two up to 5104
        p422, d422 = ica.host_transform(ica.synth_rgb(w, 40, w & 7), 95)
        datas.append(helpers.baseline_layout_from_444(p422, d422, [(2, 1), (1, 1), (1, 1)], -1))
    for req in (3, 4):
        wants = [oracle.load(d, req)[1] for d in datas]
        for band_rows in (0, 1, 2):
            if band_rows:
                monkeypatch.setenv("MIJ_BAND_ROWS", str(band_rows))
            else:
                monkeypatch.delenv("MIJ_BAND_ROWS", raising=False)
            b, slots = _batch_for(ica, gpu_ctx, datas, req)
            b.submit()
            for s, want, d in zip(slots, wants, datas):
                assert b.slot_path(s) in (1, 4, 6), b.slot_path(s)
                assert np.array_equal(b.fetch(s), want), (s, req, band_rows, want.shape)
            b.close()


def test_progressive_and_444(ica, oracle, gpu_ctx, golden):
    """Config-4 shape at reduced size: progressive 4:4:4 (multi-scan coefficient re-staging on the
    host, register-resident 4:4:4 kernel on the GPU), plus larger libjpeg-made fixtures."""
    data = golden.jpg("prog_444_64x64")
    assert np.array_equal(ica.stbi_load_from_memory(data, 3)[0], golden.expect("prog_444_64x64", 3)[1])
    for name in ("big_prog_444_256x256", "big_prog_420_320x200", "big_b422_320x240", "big_b444_rst_250x130"):
        got = ica.stbi_load_from_memory(golden.jpg(name), 3)
        assert got is not None, name
        assert np.array_equal(got[0], golden.expect(name, 3)[1]), name
    # seeded 4:4:4 images of awkward sizes, both output widths, fused vs two-pass vs oracle
    rng = np.random.default_rng(5)
    datas = [ica.stbi_write_jpg_to_memory(rng.integers(0, 256, (h, w, 3)).astype(np.uint8), 95) for (w, h) in ((8, 8), (9, 7), (250, 131), (64, 200), (1, 1))]
    datas.append(ica.synth_jpeg(1024, 768, 3, quality=95))
    for req in (3, 4):
        wants = [oracle.load(d, req)[1] for d in datas]
        for generic in (0, 1, 2):
            b, slots = _batch_for(ica, gpu_ctx, datas, req)
            b.force_generic(generic)
            b.submit()
            for s, want in zip(slots, wants):
                assert b.slot_path(s) == (2 if generic else 3)
                assert np.array_equal(b.fetch(s), want), (s, req, generic)
            b.close()


def test_default_front_end_routes_small_pictures_to_the_host_walk(golden, ica, oracle, gpu_ctx, monkeypatch):
    """mjh_decode_batch sends pictures below a size threshold (2200 pixels per host thread, MIJ_GPU_WALK_BATCH_MIN_PIXELS) to the
    host walk and the rest to the GPU walk in one call: pictures on both sides of the threshold, short files that are not small
    pictures (flat content), a rejected header and a rejected stream among them, every threshold -- same slots, reasons and pixels."""
    datas = [ica.synth_jpeg(w, h, (w + h) & 7, q) for (w, h, q) in ((64, 64, 90), (200, 180, 50), (512, 512, 90), (33, 17, 95), (1000, 700, 75), (96, 96, 10), (640, 480, 90))]
    datas.append(ica.stbi_write_jpg_to_memory(np.full((900, 1200, 3), 77, np.uint8), 90))  # 1 Mpix in a few KB: a short file, not a small picture
    datas.insert(2, golden.jpg("garbage"))
    datas.insert(5, golden.jpg("trunc_noeoi"))
    wants = [oracle.load(d, 3) for d in datas]
    for thr in (None, "0", "5000", "100000", "100000000"):
        if thr is None:
            monkeypatch.delenv("MIJ_GPU_WALK_BATCH_MIN_PIXELS", raising=False)
        else:
            monkeypatch.setenv("MIJ_GPU_WALK_BATCH_MIN_PIXELS", thr)
        b = ica.Batch(gpu_ctx, len(datas), 24 << 20, 24 << 20, 24 << 20)
        ok, slots, reasons = b.decode_jpegs(datas, 3, threads=3)
        assert ok == len(datas) - 2, (thr, reasons)
        assert slots[2] == -1 and reasons[2] == "unknown image type", thr
        assert slots[5] < -1 and reasons[5] == "expected marker", (thr, slots[5], reasons[5])
        b.submit()
        b.wait()
        for i, (d, want) in enumerate(zip(datas, wants)):
            if slots[i] >= 0:
                assert np.array_equal(b.fetch(slots[i]), want[1]), (thr, i)
        b.close()


def test_batch_front_end_thread_pool(golden, ica, oracle, gpu_ctx):
    """mjh_decode_batch: host stage on a thread pool; a rejected header costs no slot, a rejected
    entropy segment keeps its slot but is skipped by the launch; everything else equals the oracle."""
    datas = [ica.synth_jpeg(96 + 16 * i, 64 + 8 * i, i) for i in range(10)]
    datas.insert(3, golden.jpg("garbage"))          # not a JPEG: rejected at the header
    datas.insert(7, golden.jpg("trunc_noeoi"))      # header fine, stream rejected ("expected marker")
    datas.append(golden.jpg("prog_420_64x64"))
    datas.append(golden.jpg("grey_33x20"))
    b = ica.Batch(gpu_ctx, len(datas), 64 << 20, 64 << 20, 64 << 20)
    ok, slots, reasons = b.decode_jpegs(datas, 3, threads=4)
    assert ok == len(datas) - 2
    assert slots[3] == -1 and reasons[3] == "unknown image type"
    assert slots[7] < -1 and reasons[7] == "expected marker"
    b.submit()
    b.wait()
    for i, d in enumerate(datas):
        if slots[i] >= 0:
            assert np.array_equal(b.fetch(slots[i]), oracle.load(d, 3)[1]), i
    with pytest.raises(ica.MijError):
        b.fetch(-1 - slots[7])
    # the whole output arena in one asynchronous D2H into pinned memory
    pin = ica.PinnedBuffer(b.out_total_bytes())
    b.fetch_all_async(pin.ptr, pin.nbytes)
    b.wait()
    for i, d in enumerate(datas):
        if slots[i] >= 0:
            want = oracle.load(d, 3)[1].reshape(-1)
            off = b.out_offset(slots[i])
            assert np.array_equal(pin.array[off:off + want.size], want), i
    with pytest.raises(ica.MijError):
        b.fetch_all_async(pin.ptr, 16)
    pin.close()
    b.close()


def test_batch_reuse_with_other_images(ica, oracle, gpu_ctx):
    """A batch that is reset and filled again: the front ends add images without clearing the staging planes
    (mij_batch_add_uncleared), so every worker must clear its own -- busy pictures first, flat ones of other sizes
    second, through the host walk and through the GPU-walk front end (whose host-walk slots take the same route)."""
    rng = np.random.default_rng(11)
    busy = [ica.stbi_write_jpg_to_memory(rng.integers(0, 256, (96, 128, 3)).astype(np.uint8), 95) for _ in range(6)]
    flat = [ica.stbi_write_jpg_to_memory(np.full((h, w, 3), 40 + 20 * i, np.uint8), 50) for i, (w, h) in enumerate(((128, 96), (64, 64), (200, 40), (16, 16), (96, 128), (33, 17)))]
    prog = [helpers.progressive_from_du(*ica.host_transform(rng.integers(0, 256, (64, 80, 3)).astype(np.uint8), 92), 0)]
    for gpu_entropy in (False, True):
        b = ica.Batch(gpu_ctx, 8, 16 << 20, 16 << 20, 16 << 20)
        if gpu_entropy:
            b.entropy_reserve(4 << 20)
        for datas in (busy + prog, flat + prog, busy[:3] + flat[:3]):
            b.reset()
            ok, slots, reasons = b.decode_jpegs(datas, 3, threads=3, gpu_entropy=gpu_entropy)
            assert ok == len(datas), reasons
            b.submit()
            b.wait()
            for d, s_ in zip(datas, slots):
                assert np.array_equal(b.fetch(s_), oracle.load(d, 3)[1]), (gpu_entropy, s_)
        b.close()


def test_batch_api_errors(ica, gpu_ctx):
    d = ica.HostDecoder.probe(ica.synth_jpeg(32, 32, 0), 3)
    b = ica.Batch(gpu_ctx, 1, 1 << 20, 1 << 20, 1 << 20)
    with pytest.raises(ica.MijError):
        b.launch()  # nothing uploaded
    b.add(d)
    with pytest.raises(ica.MijError):
        b.add(d)  # batch full
    bad = ica.ImageDesc()
    b.reset()
    with pytest.raises(ica.MijError):
        b.add(bad)
    b.close()


def test_config4_full_size_progressive_444(ica, oracle, gpu_ctx):
    """BASELINE config 4 at its real size: a 4096x4096 progressive 4:4:4 stream (ten scans, spectral
    selection + successive approximation, EOB runs) made by tests/support/prog_writer.c from the same
    data units as the baseline stream.  Host stage on the thread pool, register-resident kernel on the
    GPU; whole image equal to the oracle's decode of the same stream, and to the baseline twin."""
    img = ica.synth_rgb(4096, 4096, 1)
    plan, du = ica.host_transform(img, 95)
    base = ica.emit_jpeg(plan, du)
    prog = helpers.progressive_from_du(plan, du, 1)
    want = oracle.load(prog, 3)[1]
    assert want.shape == (4096, 4096, 3)
    d = ica.HostDecoder.probe(prog, 3)
    cb, ob = ica.Batch.coef_bytes(d), ica.Batch.out_bytes(d)
    b = ica.Batch(gpu_ctx, 2, 2 * cb, 2 * cb, 2 * ob)
    ok, slots, reasons = b.decode_jpegs([prog, base], 3, threads=2)
    assert ok == 2, reasons
    b.submit()
    b.wait()
    for s in slots:
        assert b.slot_path(s) == 3
        assert np.array_equal(b.fetch(s), want), s
    b.close()


def test_progressive_420_full_hd_takes_the_fused_kernel(ica, oracle, gpu_ctx):
    """A progressive 1080p 4:2:0 stream lands in the same staging layout, so the headline kernel serves it."""
    plan, du = ica.host_transform(ica.synth_rgb(1920, 1080, 7), 90)
    prog = helpers.progressive_from_du(plan, du, 1)
    for req in (3, 4):
        b, slots = _batch_for(ica, gpu_ctx, [prog], req)
        b.submit()
        b.wait()
        assert b.slot_path(slots[0]) == 1
        assert np.array_equal(b.fetch(slots[0]), oracle.load(prog, req)[1]), req
        b.close()


def test_fused_422_kernel_vs_oracle_and_two_pass(ica, oracle, gpu_ctx, golden):
    """h2v1 streams (libjpeg-made fixtures and seeded ones re-emitted by the test-side writer) through the
    fused 4:2:2 kernel and through the two-pass family: both equal to the oracle, including the
    reference's last-but-one-column form (codec/jpeg.c:1805), odd widths and one-MCU images."""
    for name in ("big_b422_320x240",):
        assert np.array_equal(ica.stbi_load_from_memory(golden.jpg(name), 3)[0], golden.expect(name, 3)[1]), name
    datas = []
    for i, (w, h) in enumerate(((16, 8), (17, 9), (1, 1), (2, 2), (4, 4), (33, 7), (64, 64), (250, 131), (640, 480), (1920, 1080), (36, 20), (18, 40))):
        plan, du = ica.host_transform(ica.synth_rgb(w, h, 20 + i), 95)
        datas.append(helpers.progressive_422_from_444(plan, du, i & 1))
    for req in (3, 4):
        wants = [oracle.load(d, req)[1] for d in datas]
        for generic in (0, 1, 2):
            b, slots = _batch_for(ica, gpu_ctx, datas, req)
            b.force_generic(generic)
            b.submit()
            for s, want in zip(slots, wants):
                assert b.slot_path(s) == (2 if generic else 4)
                got = b.fetch(s)
                assert np.array_equal(got, want), (s, req, generic, want.shape, int((got != want).sum()))
            b.close()


def test_fused_grey_kernel_vs_oracle_and_two_pass(ica, oracle, gpu_ctx, golden):
    """Single-component streams: IDCT straight into the pixel buffer for every req_comp (y | y,255 | y,y,y |
    y,y,y,255), aligned and unaligned widths, against the oracle and the two-pass family."""
    datas = [golden.jpg("grey_33x20")]
    for i, (w, h) in enumerate(((8, 8), (16, 9), (1, 1), (7, 3), (64, 64), (250, 131), (640, 480), (1920, 1080), (36, 20))):
        plan, du = ica.host_transform(ica.synth_rgb(w, h, 40 + i), 95)
        datas.append(helpers.progressive_grey_from_444(plan, du, i & 1))
    for req in (0, 1, 2, 3, 4):
        wants = [oracle.load(d, req)[1] for d in datas]
        for generic in (0, 1, 2):
            b, slots = _batch_for(ica, gpu_ctx, datas, req)
            b.force_generic(generic)
            b.submit()
            for s, want in zip(slots, wants):
                assert b.slot_path(s) == (2 if generic else 5)
                got = b.fetch(s)
                assert np.array_equal(got.reshape(-1), want.reshape(-1)), (s, req, generic, want.shape)
            b.close()


def test_two_pass_layouts_specialised_and_general_pass2(ica, oracle, gpu_ctx):
    """Every layout the fused kernels do not take -- 4:4:0, 4:1:1, 4:1:0, h2v4, h1v4, RGB-tagged, CMYK, YCCK, four-component
    YCbCr with sub-sampled chroma -- through the two-pass family with the pass 2 compiled per resampler (k_resample_fast:
    row_1 / v_2 / h_2 / hv_2 / generic x2 / generic x4, codec/jpeg.c:1765-1840, :1962-1971) and with the run-time-general
    pass 2 (k_resample_color), both equal to the oracle; widths that are not a multiple of four stay on the general one."""
    layouts = [  # (factors per component, Adobe transform or -1)
        ([(1, 2), (1, 1), (1, 1)], -1),          # 4:4:0  v_2
        ([(4, 1), (1, 1), (1, 1)], -1),          # 4:1:1  generic x4
        ([(4, 2), (1, 1), (1, 1)], -1),          # 4:1:0  generic x4, two rows
        ([(2, 4), (1, 1), (1, 1)], -1),          # h2v4   generic x2
        ([(1, 4), (1, 1), (1, 1)], -1),          # h1v4   generic x1 = the near row
        ([(2, 2), (1, 1), (1, 1)], -1),          # 4:2:0 (forced off the fused kernel below)
        ([(2, 1), (1, 1), (1, 1)], -1),          # 4:2:2
        ([(1, 1), (1, 1), (1, 1)], 0),           # RGB-tagged
        ([(2, 2), (1, 1), (1, 1)], 0),           # RGB-tagged with sub-sampled G / B planes
        ([(1, 1), (1, 1), (1, 1), (1, 1)], 0),   # CMYK
        ([(1, 1), (1, 1), (1, 1), (1, 1)], 2),   # YCCK
        ([(2, 2), (1, 1), (1, 1), (2, 2)], 2),   # YCCK, chroma sub-sampled
        ([(2, 1), (1, 1), (1, 1), (2, 1)], 0),   # CMYK, h2v1 on the middle planes
        ([(1, 1), (1, 1), (1, 1), (1, 1)], 1),   # four components, other transform: YCbCr, fourth ignored
        ([(2, 2), (1, 1), (1, 1), (1, 1)], 2),   # fourth component not at full resolution: general pass 2 only
        ([(1, 1), (2, 2), (2, 2)], -1),          # luma below the chroma resolution: general pass 2 only
    ]
    sizes = ((64, 48), (36, 20), (128, 72), (4, 4), (30, 17), (200, 97), (1040, 24))  # the last: 130 blocks a row, wavefronts inside one block row (LDS-transposed stores)
    datas, fast = [], []
    for li, (hv, app14) in enumerate(layouts):
        for si, (w, h) in enumerate(sizes):
            plan, du = ica.host_transform(ica.synth_rgb(w, h, 300 + 7 * li + si), 92)
            datas.append(helpers.baseline_layout_from_444(plan, du, hv, app14, restart_mcus=(3 if si == 2 else 0)))
            fast.append(w % 4 == 0 and li < 14)
    for req in (3, 4):
        wants = [oracle.load(d, req) for d in datas]
        assert all(k == "ok" for k, _, _ in wants)
        for generic in (1, 2):
            b, slots = _batch_for(ica, gpu_ctx, datas, req)
            b.force_generic(generic)
            b.submit()
            b.wait()
            for i, (s, (_, want, _)) in enumerate(zip(slots, wants)):
                assert b.slot_path(s) == 2
                got = b.fetch(s)
                assert np.array_equal(got, want), (i, layouts[i // len(sizes)], sizes[i % len(sizes)], req, generic, int((got != want).sum()))
            b.close()
    # default choice: the same pixels again (fused kernels where they apply, two-pass elsewhere).  Round 3: RGB-tagged, CMYK and YCCK files
    # whose components are all 1x1 take k_fused1x1c (path 7), four-component YCbCr with the fourth ignored takes k_fused444 (path 3)
    for req in (3, 4):
        b, slots = _batch_for(ica, gpu_ctx, datas, req)
        b.submit()
        b.wait()
        for i, s in enumerate(slots):
            li = i // len(sizes)
            assert np.array_equal(b.fetch(s), oracle.load(datas[i], req)[1]), (i, layouts[li], sizes[i % len(sizes)], req)
            if li in (7, 9, 10):
                assert b.slot_path(s) == 7, (li, b.slot_path(s))
            if li == 13:
                assert b.slot_path(s) == 3, (li, b.slot_path(s))
        b.close()


def test_fused_440_kernel_vs_oracle_and_two_pass(ica, oracle, gpu_ctx, monkeypatch):
    """h1v2 (4:4:0) streams through the band kernel's H2 = false form (k_fused440: vertical filter only, codec/jpeg.c:1774-1782) and
    through both pass-2 forms of the two-pass family: all equal to the oracle -- odd widths and heights, one-MCU images, every band
    height (1 / 2 / 3 MCU rows per workgroup / whole image: halo rows, saved rows, the last odd row), 3 and 4 output channels."""
    datas = []
    for i, (w, h) in enumerate(((8, 16), (9, 17), (1, 1), (2, 2), (4, 4), (33, 7), (64, 64), (250, 131), (640, 480), (1920, 1080), (36, 20), (18, 40), (52, 33))):
        plan, du = ica.host_transform(ica.synth_rgb(w, h, 60 + i), 93)
        datas.append(helpers.baseline_layout_from_444(plan, du, [(1, 2), (1, 1), (1, 1)], -1, restart_mcus=(5 if i % 4 == 1 else 0)))
    for req in (3, 4):
        wants = [oracle.load(d, req)[1] for d in datas]
        for generic in (0, 1, 2):
            b, slots = _batch_for(ica, gpu_ctx, datas, req)
            b.force_generic(generic)
            b.submit()
            for s, want in zip(slots, wants):
                assert b.slot_path(s) == (2 if generic else 6)
                got = b.fetch(s)
                assert np.array_equal(got, want), (s, req, generic, want.shape, int((got != want).sum()))
            b.close()
    wants = [oracle.load(d, 3)[1] for d in datas]
    for rows in ("1", "2", "3", "1000"):
        monkeypatch.setenv("MIJ_BAND_ROWS", rows)
        b, slots = _batch_for(ica, gpu_ctx, datas, 3)
        b.submit()
        b.wait()
        for s, want in zip(slots, wants):
            assert b.slot_path(s) == 6
            assert np.array_equal(b.fetch(s), want), (rows, s)
        b.close()
