"""GPU entropy stage (experimental): the baseline Huffman walk on the GPU.
The coefficient planes it leaves in HBM must equal, element for element, what the host walk stages; the
decoded pixels must equal the oracle's; streams it refuses must come back in the fallback list."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def _host_planes(ica, data, req):
    return ica.HostDecoder.decode(data, req)


@pytest.mark.parametrize("fmt", ["compact", "int16"])
def test_gpu_walk_equals_host_walk(ica, oracle, gpu_ctx, golden, fmt):
    """Both plane formats the write pass knows: compact planes (the default: low bytes + escape bytes + DC array)
    and the int16 tile layout.  fetch_coef expands compact planes, so both compare with the host walk's staging."""
    datas = [ica.synth_jpeg(w, h, i, q) for i, (w, h, q) in enumerate(((64, 48, 90), (16, 16, 50), (200, 120, 90), (33, 17, 75), (640, 480, 90), (1920, 1080, 90),
                                                                        (8, 8, 95), (250, 131, 95), (1, 1, 90), (1024, 768, 30)))]
    rng = np.random.default_rng(3)
    datas.append(ica.stbi_write_jpg_to_memory(rng.integers(0, 256, (211, 307, 3)).astype(np.uint8), 92))   # noise: long codes, big coefficients
    datas.append(ica.stbi_write_jpg_to_memory(rng.integers(0, 256, (120, 160, 3)).astype(np.uint8), 100))  # q=100: WIDE candidates
    b = ica.Batch(gpu_ctx, len(datas), 64 << 20, 64 << 20, 64 << 20)
    b.set_coef_format(fmt)
    b.entropy_reserve(16 << 20)
    slots = []
    for d in datas:
        st, slot = b.add_jpeg_stream(d, 3)
        assert st == 1, (st, b.last_reason)
        slots.append(slot)
    fallback = b.entropy_run()
    assert fallback == [], fallback
    assert 1 <= b.entropy_rounds() <= 24
    assert {b.slot_coef_bytes(s) for s in slots} == {1 if fmt == "compact" else 0}
    if fmt == "compact":  # the two noise images hold coefficients beyond a byte: escaped blocks, not fallbacks
        assert b.slot_escapes(slots[-1]) > 0 and b.slot_escapes(slots[-2]) > 0
    for d, s in zip(datas, slots):
        desc, want = _host_planes(ica, d, 3)
        got = b.fetch_coef(s)
        # block by block over the MCU grid (tile slots past the last block are never read and never written)
        for ci, (pg, pw) in enumerate(zip(ica.detile_coefficients(desc, got), ica.detile_coefficients(desc, want))):
            assert np.array_equal(pg, pw), (s, ci, int((pg != pw).sum()))
    b.submit()
    b.wait()
    for d, s in zip(datas, slots):
        assert np.array_equal(b.fetch(s), oracle.load(d, 3)[1]), s
    b.close()


def test_gpu_walk_refuses_what_it_should(ica, oracle, gpu_ctx, golden):
    """Layouts outside its scope are not taken (status 2), damaged entropy data is reported for the host walk,
    and the host walk's result for those slots is what the oracle says."""
    b = ica.Batch(gpu_ctx, 16, 64 << 20, 64 << 20, 64 << 20)
    b.entropy_reserve(8 << 20)
    for name in ("prog_420_64x64", "dri_without_rst"):
        st, _ = b.add_jpeg_stream(golden.jpg(name), 3)
        assert st == 2, name
    st, _ = b.add_jpeg_stream(golden.jpg("garbage"), 3)
    assert st == 0 and b.last_reason == "unknown image type"
    good = ica.synth_jpeg(320, 200, 5, 90)
    bad = bytearray(good)
    pos = len(bad) // 2
    for k in range(24):  # a burst of damage in the middle of the entropy segment, no 0xff created or destroyed
        if bad[pos + k] != 0xFF and bad[pos + k - 1] != 0xFF:
            bad[pos + k] = (bad[pos + k] * 7 + 13) % 255
    bad = bytes(bad)
    s_good = b.add_jpeg_stream(good, 3)[1]
    st, s_bad = b.add_jpeg_stream(bad, 3)
    assert st == 1
    fallback = b.entropy_run()
    assert s_good not in fallback
    kind, want, reason = oracle.load(bad, 3)
    if s_bad in fallback:
        b.fallback_prepare(s_bad)
        try:
            d2, _ = ica.HostDecoder.decode(bad, 3, out=b.staging(s_bad))
            if d2.flags:
                b.set_flags(s_bad, d2.flags)
            host_ok = True
        except ica.MijError:
            host_ok = False
        assert host_ok == (kind == "ok")
        if not host_ok:
            b.set_flags(s_bad, 2)  # MIJ_FLAG_SKIP
    b.submit()
    b.wait()
    assert np.array_equal(b.fetch(s_good), oracle.load(good, 3)[1])
    if kind == "ok":
        assert np.array_equal(b.fetch(s_bad), want)
    b.close()


def test_batch_front_end_with_gpu_walk(golden, ica, oracle, gpu_ctx):
    """mjh_decode_batch_gpu: same contract as the host-only front end -- rejected headers cost no slot, rejected
    streams keep theirs and are skipped, layouts the GPU walk does not take (progressive, restart markers,
    grey 4-component ...) and streams it refuses go through the host walk; every decoded image equals the oracle."""
    datas = [ica.synth_jpeg(96 + 16 * i, 64 + 8 * i, i) for i in range(8)]
    datas.insert(2, golden.jpg("garbage"))
    datas.insert(5, golden.jpg("trunc_noeoi"))
    datas.append(golden.jpg("prog_420_64x64"))
    datas.append(golden.jpg("grey_33x20"))
    datas.append(golden.jpg("big_b444_rst_250x130"))
    datas.append(golden.jpg("big_b422_320x240"))
    datas.append(ica.synth_jpeg(640, 480, 9, 95))
    good = ica.synth_jpeg(320, 200, 5, 90)
    bad = bytearray(good)
    for k in range(len(bad) // 2, len(bad) // 2 + 24):
        if bad[k] != 0xFF and bad[k - 1] != 0xFF:
            bad[k] = (bad[k] * 7 + 13) % 255
    datas.append(bytes(bad))
    for req in (3, 4):
        b = ica.Batch(gpu_ctx, len(datas), 64 << 20, 64 << 20, 64 << 20)
        b.entropy_reserve(8 << 20)
        ok, slots, reasons = b.decode_jpegs(datas, req, threads=4, gpu_entropy=True)
        assert slots[2] == -1 and reasons[2] == "unknown image type"
        b.submit()
        b.wait()
        n_ok = 0
        for i, d in enumerate(datas):
            kind, want, why = oracle.load(d, req)
            if slots[i] >= 0:
                assert kind == "ok", (i, why)
                assert np.array_equal(b.fetch(slots[i]), want), (i, req)
                n_ok += 1
            elif slots[i] < -1:
                assert kind == "fail" and reasons[i] == want, (i, reasons[i], want)
        assert n_ok == ok
        b.close()


def test_gpu_walk_fuzzed_streams_vs_oracle(golden, ica, oracle, gpu_ctx):
    """Damaged entropy segments through the GPU front end: whatever the GPU walk accepts must be what the
    reference semantics give, everything else must have come back for the host walk -- accept/reject, reason
    and pixels equal to the oracle for every stream."""
    import helpers
    bases = [ica.synth_jpeg(160, 120, 1, 90), ica.synth_jpeg(97, 131, 2, 75), ica.synth_jpeg(64, 64, 3, 95), golden.jpg("b420_64x64_q90"), golden.jpg("grey_33x20")]
    datas = []
    for k in range(180):
        base = bases[k % len(bases)]
        datas.append(helpers.mutate(base, 1000 + k, n_mut=1 + k % 4, allow_markers=(k % 3 == 0)))
    n_gpu = n_fail = 0
    for lo in range(0, len(datas), 60):
        part = datas[lo:lo + 60]
        b = ica.Batch(gpu_ctx, len(part), 16 << 20, 16 << 20, 16 << 20)
        b.entropy_reserve(4 << 20)
        ok, slots, reasons = b.decode_jpegs(part, 3, threads=4, gpu_entropy=True)
        b.submit()
        b.wait()
        for i, d in enumerate(part):
            kind, want, _ = oracle.load(d, 3)
            if slots[i] >= 0:
                assert kind == "ok", (lo + i, want)
                assert np.array_equal(b.fetch(slots[i]), want), lo + i
                n_gpu += 1
            else:
                assert kind == "fail", lo + i
                assert reasons[i] == want, (lo + i, reasons[i], want)
                n_fail += 1
        b.close()
    assert n_gpu > 100


def test_wide_streams_in_every_fused_family_and_through_the_gpu_walk(golden, ica, oracle, gpu_ctx):
    """Quantisation tables blown up so that the first IDCT pass overflows int16: the exact (WIDE) instantiation of
    each fused kernel -- 4:2:0, 4:2:2, 4:4:4, grey -- must equal the reference's wrapping arithmetic, whether the
    flag was raised by the host walk or by the GPU walk's per-block bound."""
    datas = []
    for name in ("b420_64x64_q90", "big_b422_320x240", "b444_40x24_q95", "grey_33x20"):
        data = bytearray(golden.jpg(name))
        rng = np.random.default_rng(len(name))
        pos = 0
        while True:  # every DQT segment
            pos = bytes(data).find(b"\xff\xdb", pos)
            if pos < 0:
                break
            length = int.from_bytes(data[pos + 2:pos + 4], "big")
            for t in range((length - 2) // 65):
                for k in range(64):
                    data[pos + 5 + 65 * t + k] = int(rng.integers(100, 256))
            pos += 2 + length
        datas.append(bytes(data))
    wants = [oracle.load(d, 3)[1] for d in datas]
    flags = [ica.HostDecoder.decode(d, 3)[0].flags & 1 for d in datas]
    assert sum(flags) >= 3
    expect_path = [1, 4, 3, 5]
    for gpu_entropy in (False, True):
        b = ica.Batch(gpu_ctx, len(datas), 16 << 20, 16 << 20, 16 << 20)
        b.entropy_reserve(1 << 20)
        ok, slots, reasons = b.decode_jpegs(datas, 3, threads=2, gpu_entropy=gpu_entropy)
        assert ok == len(datas), reasons
        b.submit()
        b.wait()
        for s, want, path in zip(slots, wants, expect_path):
            assert b.slot_path(s) == path
            assert np.array_equal(b.fetch(s), want), (gpu_entropy, path)
        b.close()


def test_gpu_walk_restart_intervals_and_own_tables(ica, oracle, gpu_ctx, golden):
    """Baseline streams with optimal (not the writer's standard) Huffman tables and restart intervals from one MCU
    to more than the image holds, in 4:2:0, 4:4:4, 4:2:2 and grey: every interval is its own chain with fresh DC
    predictors; planes equal to the host walk's, pixels to the oracle's, nothing handed back."""
    import helpers
    datas = [golden.jpg("big_b444_rst_250x130")]
    for i, (w, h, q, dri, lay) in enumerate(((64, 48, 90, 0, "native"), (64, 48, 90, 1, "native"), (200, 133, 90, 4, "native"), (250, 131, 95, 17, "native"),
                                             (250, 131, 95, 3, "422"), (97, 51, 95, 2, "grey"), (16, 16, 90, 1, "native"), (640, 480, 90, 40, "native"),
                                             (640, 480, 75, 1000, "native"), (1920, 1080, 90, 120, "native"), (1920, 1080, 95, 60, "422"), (33, 17, 100, 2, "native"))):
        plan, du = ica.host_transform(ica.synth_rgb(w, h, 60 + i) if i % 3 else np.random.default_rng(i).integers(0, 256, (h, w, 3)).astype(np.uint8), q)
        datas.append(helpers.baseline_from_du(plan, du, dri, lay))
    b = ica.Batch(gpu_ctx, len(datas), 8 << 20, 96 << 20, 96 << 20)
    b.entropy_reserve(16 << 20)
    slots = []
    for d in datas:
        st, slot = b.add_jpeg_stream(d, 3)
        assert st == 1, (st, b.last_reason)
        slots.append(slot)
    assert b.entropy_run() == []
    for d, s_ in zip(datas, slots):
        desc, want = ica.HostDecoder.decode(d, 3)
        got = b.fetch_coef(s_)
        for ci, (pg, pw) in enumerate(zip(ica.detile_coefficients(desc, got), ica.detile_coefficients(desc, want))):
            assert np.array_equal(pg, pw), (s_, ci, int((pg != pw).sum()))
    b.submit()
    b.wait()
    for d, s_ in zip(datas, slots):
        assert np.array_equal(b.fetch(s_), oracle.load(d, 3)[1]), s_
    b.close()
    # damaged restart streams through the front end: verdict, reason and pixels as the oracle says
    fuzz = [helpers.mutate(datas[1 + k % 9], 2000 + k, n_mut=1 + k % 3, allow_markers=(k % 4 == 0)) for k in range(120)]
    b = ica.Batch(gpu_ctx, len(fuzz), 64 << 20, 64 << 20, 64 << 20)
    b.entropy_reserve(16 << 20)
    ok, slots, reasons = b.decode_jpegs(fuzz, 3, threads=4, gpu_entropy=True)
    b.submit()
    b.wait()
    for i, d in enumerate(fuzz):
        kind, want, _ = oracle.load(d, 3)
        if slots[i] >= 0:
            assert kind == "ok", (i, want)
            assert np.array_equal(b.fetch(slots[i]), want), i
        else:
            assert kind == "fail" and reasons[i] == want, (i, reasons[i], want)
    b.close()


def test_compact_planes_are_the_default(ica, oracle, gpu_ctx):
    """Without any knob the GPU walk leaves every image it takes as compact planes -- whatever the sampling, the
    quantiser or the size of the coefficients -- and every decode kernel family reads them; an image with
    coefficients outside -128..127 gets escape bytes instead of a trip through the host walk."""
    rng = np.random.default_rng(8)
    datas = [ica.synth_jpeg(w, h, i, q) for i, (w, h, q) in enumerate(((64, 48, 90), (200, 120, 90), (33, 17, 75), (640, 480, 90), (1920, 1080, 90), (250, 131, 50)))]
    datas.append(ica.synth_jpeg(320, 200, 7, 95))                                                         # 4:4:4
    datas.append(ica.stbi_write_jpg_to_memory(rng.integers(0, 256, (120, 160, 3)).astype(np.uint8), 90))   # noise: coefficients beyond a byte
    datas.append(ica.synth_jpeg(96, 96, 9, 10))                                                            # quantisers up to 255
    plan, du = ica.host_transform(rng.integers(0, 256, (64, 80, 3)).astype(np.uint8), 95)
    datas.append(helpers.baseline_from_du(plan, du, layout="grey"))                                        # one component
    for req in (3, 4):
        b = ica.Batch(gpu_ctx, len(datas) + 2, 64 << 20, 64 << 20, 64 << 20)
        b.entropy_reserve(8 << 20)
        ok, slots, reasons = b.decode_jpegs(datas, req, threads=2, gpu_entropy=True)
        assert ok == len(datas), reasons
        c1 = b.add_clone(slots[4])
        c2 = b.add_clone(slots[7])
        b.submit()
        b.wait()
        for d, s_ in zip(datas, slots):
            assert np.array_equal(b.fetch(s_), oracle.load(d, req)[1]), (s_, req)
        assert np.array_equal(b.fetch(c1), oracle.load(datas[4], req)[1])
        assert np.array_equal(b.fetch(c2), oracle.load(datas[7], req)[1])
        assert [b.slot_coef_bytes(s_) for s_ in slots] == [1] * len(datas)
        assert b.slot_escapes(slots[7]) > 0
        assert {b.slot_path(s_) for s_ in slots} == {1, 3, 5}
        b.close()


def test_gpu_walk_is_the_default_front_end(ica, oracle, gpu_ctx, golden, monkeypatch):
    """mjh_decode_batch without any preparation: the batch gets its entropy arena on first use, baseline single-scan files
    are walked on the GPU (no staging needed for them), everything else and every damaged stream by the host walk --
    verdicts, reasons and pixels are the oracle's.  MIJ_ENTROPY=host turns the default off."""
    import helpers
    good = [ica.synth_jpeg(200 + 16 * i, 120 + 8 * i, i, (90, 95, 50)[i % 3]) for i in range(6)]
    datas = good + [golden.jpg(n) for n in ("prog_420_64x64", "cmyk_40x30", "grey_33x20", "garbage", "trunc_noeoi", "rst_blocks_64x48")]
    datas += [helpers.mutate(good[k % 6], 900 + k, n_mut=1 + k % 3, allow_markers=(k % 5 == 0)) for k in range(40)]
    for mode in ("default", "host"):
        if mode == "host":
            monkeypatch.setenv("MIJ_ENTROPY", "host")
        b = ica.Batch(gpu_ctx, len(datas), 64 << 20, 64 << 20, 64 << 20)
        ok, slots, reasons = b.decode_jpegs(datas, 3, threads=4)
        b.submit()
        b.wait()
        n_ok = 0
        for i, d in enumerate(datas):
            kind, want, _ = oracle.load(d, 3)
            if slots[i] >= 0:
                assert kind == "ok", (mode, i, want)
                assert np.array_equal(b.fetch(slots[i]), want), (mode, i)
                n_ok += 1
            else:
                assert kind == "fail" and reasons[i] == want, (mode, i, reasons[i], want)
        assert ok == n_ok
        b.close()
    monkeypatch.delenv("MIJ_ENTROPY")


def test_stbi_load_from_memory_takes_the_gpu_walk_for_large_pictures(ica, oracle, gpu_ctx, golden, monkeypatch):
    """>= 800x600 pixels from memory (MIJ_GPU_WALK_MIN_PIXELS): header on the host, Huffman walk + everything else on the GPU.  With the threshold
    at zero every golden stream goes through the same entry: what the GPU walk does not take or reports back falls
    through to the host walk, so results and failure reasons stay the reference's."""
    for (w, h, q) in ((512, 512, 90), (1920, 1080, 90), (800, 600, 95), (1280, 1024, 90), (2048, 1536, 85)):
        data = ica.synth_jpeg(w, h, seed=w, quality=q)
        for req in (0, 3, 4, 1):
            got = ica.stbi_load_from_memory(data, req)
            assert got is not None, ica.stbi_failure_reason()
            assert np.array_equal(got[0], oracle.load(data, req)[1]), (w, h, req)
    monkeypatch.setenv("MIJ_GPU_WALK_MIN_PIXELS", "0")
    for name in golden.names:
        data = golden.jpg(name)
        for req in (0, 3):
            kind, want = golden.expect(name, req)
            if kind == "skip":
                continue
            got = ica.stbi_load_from_memory(data, req)
            if kind == "fail":
                assert got is None and ica.stbi_failure_reason() == want, (name, req, ica.stbi_failure_reason())
            else:
                assert got is not None, (name, req, ica.stbi_failure_reason())
                if name == "dri_without_rst":
                    want = oracle.load(data, req)[1]
                assert np.array_equal(got[0], want), (name, req)


@pytest.mark.parametrize("bits", [256, 512, 1024, 2048])
def test_gpu_walk_subsequence_lengths(ica, oracle, gpu_ctx, golden, monkeypatch, bits):
    """The subsequence length is a property of the batch's entropy arena (DevScan.sub_bits): one-picture batches -- stbi_load_from_memory --
    cut the stream into MIJ_ES_BITS_SINGLE = 1024 bits per lane (latency), everything else into 4096.  Every length must leave the host
    walk's planes and the oracle's pixels, damaged streams included (mutations: accepted ones equal the oracle, the rest comes back)."""
    monkeypatch.setenv("MIJ_ES_BITS_OVERRIDE", str(bits))
    rng = np.random.default_rng(bits)
    datas = [ica.synth_jpeg(w, h, i, q) for i, (w, h, q) in enumerate(((64, 48, 90), (200, 120, 90), (640, 480, 90), (1920, 1080, 90), (8, 8, 95), (1, 1, 90), (1024, 768, 30)))]
    datas.append(ica.stbi_write_jpg_to_memory(rng.integers(0, 256, (211, 307, 3)).astype(np.uint8), 92))
    datas.append(ica.stbi_write_jpg_to_memory(rng.integers(0, 256, (120, 160, 3)).astype(np.uint8), 100))
    plan, du = ica.host_transform(ica.synth_rgb(250, 131, 5), 92)
    datas.append(helpers.baseline_from_du(plan, du, restart_mcus=7))
    b = ica.Batch(gpu_ctx, len(datas), 64 << 20, 64 << 20, 64 << 20)
    b.entropy_reserve(16 << 20)
    slots = []
    for d in datas:
        st, slot = b.add_jpeg_stream(d, 3)
        assert st == 1, (st, b.last_reason)
        slots.append(slot)
    fallback = b.entropy_run()
    # a lane that cannot finish a block inside its subsequence moves its chain one subsequence per round: noise at q=100 in 256-bit
    # pieces may run out of rounds and come back for the host walk (which is what the fallback list is for); from 1024 bits on nothing may
    assert fallback == [] or bits < 1024, (bits, fallback)
    for d, s in zip(datas, slots):
        desc, want = _host_planes(ica, d, 3)
        if s in fallback:
            b.fallback_prepare(s)
            d2, _ = ica.HostDecoder.decode(d, 3, out=b.staging(s))
            if d2.flags:
                b.set_flags(s, d2.flags)
            continue
        got = b.fetch_coef(s)
        for ci, (pg, pw) in enumerate(zip(ica.detile_coefficients(desc, got), ica.detile_coefficients(desc, want))):
            assert np.array_equal(pg, pw), (bits, s, ci, int((pg != pw).sum()))
    b.submit()
    b.wait()
    for d, s in zip(datas, slots):
        assert np.array_equal(b.fetch(s), oracle.load(d, 3)[1]), (bits, s)
    b.close()
    # damaged streams through the front end
    bases = [ica.synth_jpeg(160, 120, 1, 90), ica.synth_jpeg(97, 131, 2, 75), golden.jpg("b420_64x64_q90")]
    part = [helpers.mutate(bases[k % 3], 5000 + bits + k, n_mut=1 + k % 4, allow_markers=(k % 3 == 0)) for k in range(45)]
    b = ica.Batch(gpu_ctx, len(part), 16 << 20, 16 << 20, 16 << 20)
    b.entropy_reserve(4 << 20)
    ok, slots, reasons = b.decode_jpegs(part, 3, threads=4, gpu_entropy=True)
    b.submit()
    b.wait()
    for i, d in enumerate(part):
        kind, want, why = oracle.load(d, 3)
        if slots[i] >= 0:
            assert kind == "ok", (bits, i, why)
            assert np.array_equal(b.fetch(slots[i]), want), (bits, i)
        else:
            assert kind == "fail" and reasons[i] == want, (bits, i, reasons[i], want)
    b.close()
    # and the one-picture path of the public API (its own arena, MIJ_ES_BITS_SINGLE unless overridden as here)
    monkeypatch.setenv("MIJ_GPU_WALK_MIN_PIXELS", "0")
    for d in datas[:4]:
        assert np.array_equal(ica.stbi_load_from_memory(d, 3)[0], oracle.load(d, 3)[1])


def test_default_front_end_twice_without_a_reset(ica, oracle, gpu_ctx, golden):
    """mjh_decode_batch on a batch that already holds the pictures of an earlier call.  The GPU walk's finish step looks at
    every scan of the arena, and only the current call's slots have an owner there: the second call therefore extends the batch through
    the host walk.  Both calls' pictures -- some of which the GPU walk hands back, one of which is rejected -- come out right, and the
    explicit two-half entry on such a batch fails cleanly instead of indexing with an unowned slot."""
    first = [ica.synth_jpeg(320, 200, 1), golden.jpg("prog_420_64x64"), ica.synth_jpeg(1024, 768, 2), golden.jpg("garbage")]
    second = [ica.synth_jpeg(640, 480, 3, quality=95), ica.synth_jpeg(64, 64, 4), golden.jpg("grey_33x20")]
    b = ica.Batch(gpu_ctx, 16, 64 << 20, 64 << 20, 64 << 20)
    ok1, s1, r1 = b.decode_jpegs(first, 3, threads=3)           # default front end: GPU walk where it applies
    ok2, s2, r2 = b.decode_jpegs(second, 3, threads=3)          # no reset in between
    assert ok1 == 3 and ok2 == 3, (r1, r2)
    b.submit()
    b.wait()
    for datas, slots in ((first, s1), (second, s2)):
        for d, s in zip(datas, slots):
            kind, want, _ = oracle.load(d, 3)
            if kind != "ok":
                assert s < 0
            else:
                assert np.array_equal(b.fetch(s), want)
    b.close()


def test_gpu_walk_with_a_third_pair_of_tables(ica, oracle, gpu_ctx):
    """The walk's entry tables (EsUni, EsPair) hold the first two DC and the first two AC tables a scan uses; a baseline file may name a
    third pair (the reference takes ids 0..3).  Its blocks go through the search path of both the state-only passes and the write pass: same
    planes as the host walk, same pixels as the oracle, nothing handed back."""
    datas = []
    for i, (w, h, q) in enumerate(((200, 120, 90), (640, 480, 85), (1920, 1080, 90), (333, 77, 95))):
        base = ica.synth_jpeg(w, h, i, q)
        datas.append(helpers.third_tables(base))
        assert np.array_equal(oracle.load(datas[-1], 3)[1], oracle.load(base, 3)[1])
    for fmt in ("compact", "int16"):
        b = ica.Batch(gpu_ctx, len(datas), 64 << 20, 64 << 20, 64 << 20)
        b.set_coef_format(fmt)
        b.entropy_reserve(16 << 20)
        slots = []
        for d in datas:
            st, slot = b.add_jpeg_stream(d, 3)
            assert st == 1, (st, b.last_reason)
            slots.append(slot)
        assert b.entropy_run() == []
        for d, s in zip(datas, slots):
            desc, want = ica.HostDecoder.decode(d, 3)
            for ci, (pg, pw) in enumerate(zip(ica.detile_coefficients(desc, b.fetch_coef(s)), ica.detile_coefficients(desc, want))):
                assert np.array_equal(pg, pw), (fmt, len(d), ci, int((pg != pw).sum()))
        b.submit()
        b.wait()
        for d, s in zip(datas, slots):
            assert np.array_equal(b.fetch(s), oracle.load(d, 3)[1]), (fmt, len(d))
        b.close()


def test_gpu_walk_zigzag_image_form_of_the_write_pass(ica, oracle, gpu_ctx, monkeypatch):
    """Compact planes normally come out of the write pass as a record stream (k_es_writer / k_es_pack2, round 3); an arena whose record
    indices would not fit 32 bits -- or MIJ_ES_RECORDS=0, read when the arena is reserved -- keeps the older form (k_es_write / k_es_tails /
    k_es_pack over a zigzag image).  Same planes, same pixels: escapes, restart intervals, straddling blocks, a third pair of tables."""
    rng = np.random.default_rng(5)
    datas = [ica.synth_jpeg(640, 480, 1, 90), ica.synth_jpeg(1920, 1080, 2, 90), ica.stbi_write_jpg_to_memory(rng.integers(0, 256, (211, 307, 3)).astype(np.uint8), 92),
             helpers.third_tables(ica.synth_jpeg(333, 77, 3, 85))]
    plan, du = ica.host_transform(ica.synth_rgb(400, 300, 4), 95)
    datas.append(helpers.baseline_from_du(plan, du, 5, "native"))
    planes = {}
    for form in ("0", "1"):
        monkeypatch.setenv("MIJ_ES_RECORDS", form)
        b = ica.Batch(gpu_ctx, len(datas), 64 << 20, 64 << 20, 64 << 20)
        b.entropy_reserve(16 << 20)
        slots = []
        for d in datas:
            st, slot = b.add_jpeg_stream(d, 3)
            assert st == 1, (st, b.last_reason)
            slots.append(slot)
        assert b.entropy_run() == []
        planes[form] = []
        for d, s in zip(datas, slots):
            desc, want = ica.HostDecoder.decode(d, 3)
            got = ica.detile_coefficients(desc, b.fetch_coef(s))
            for ci, (pg, pw) in enumerate(zip(got, ica.detile_coefficients(desc, want))):
                assert np.array_equal(pg, pw), (form, len(d), ci)
            planes[form].append(got)
        b.submit()
        b.wait()
        for d, s in zip(datas, slots):
            assert np.array_equal(b.fetch(s), oracle.load(d, 3)[1]), (form, len(d))
        b.close()


def test_gpu_walk_densest_record_stream(ica, oracle, gpu_ctx):
    """The record form of the write pass sizes a subsequence's region for one record per two bits of stream -- the densest a stream can be:
    a flat picture with optimal tables spends two bits per block (a one-bit DC code for category 0, a one-bit EOB).  Such streams, with and without restart intervals, through the walk at two
    subsequence lengths: planes == host walk,
    pixels == oracle, nothing handed back."""
    flat = np.full((1080, 1920, 3), 117, np.uint8)
    plan, du = ica.host_transform(flat, 90)
    datas = [helpers.baseline_from_du(plan, du, 0, "native"), helpers.baseline_from_du(plan, du, 64, "native")]
    # (Blocks full of +-1 coefficients come as close to two bits per record from the other side, but a stream of near-equal symbol lengths
    # has little for a wrong start to re-synchronise on: its chain settles a subsequence per round, and the walk -- in either form of the
    # write pass, since round 1 -- hands such a picture back to the host after 96 rounds.  Seen while writing this test)
    for d in datas:
        assert oracle.load(d, 3)[0] == "ok"
    for n_batch in (len(datas), 1):  # a batch of one picture walks with 1024-bit subsequences
        for d in (datas if n_batch == 1 else [None]):
            group = [d] if n_batch == 1 else datas
            b = ica.Batch(gpu_ctx, len(group), 64 << 20, 64 << 20, 64 << 20)
            b.entropy_reserve(8 << 20)
            slots = []
            for x in group:
                st, slot = b.add_jpeg_stream(x, 3)
                assert st == 1, (st, b.last_reason)
                slots.append(slot)
            assert b.entropy_run() == []
            for x, sl in zip(group, slots):
                desc, want = ica.HostDecoder.decode(x, 3)
                for ci, (pg, pw) in enumerate(zip(ica.detile_coefficients(desc, b.fetch_coef(sl)), ica.detile_coefficients(desc, want))):
                    assert np.array_equal(pg, pw), (n_batch, len(x), ci)
            b.submit()
            b.wait()
            for x, sl in zip(group, slots):
                assert np.array_equal(b.fetch(sl), oracle.load(x, 3)[1]), (n_batch, len(x))
            b.close()
