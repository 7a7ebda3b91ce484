"""CPU-side tests of the product: C-ABI surface, host entropy stage, JPEG writer, loud failure
without a GPU.  No GPU compute calls here (-m "not gpu")."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers

ROOT = helpers.ROOT


def _declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set()
    for m in re.finditer(r"^[A-Za-z_][\w\s\*]*?\b(stbi_\w+|mij_\w+|mjh_\w+)\s*\(", src, flags=re.M):
        names.add(m.group(1))
    # static inline helpers are not exported
    for m in re.finditer(r"static\s+inline\s+[\w\s\*]*?\b(\w+)\s*\(", src):
        names.discard(m.group(1))
    for m in re.finditer(r"typedef\s+[\w\s\*]*?\b(\w+)\s*\(", src):  # function typedefs (stbi_write_func)
        names.discard(m.group(1))
    return sorted(names)


def test_library_exports_every_declared_symbol(ica):
    L = ica.lib()
    declared = _declared_functions("image_api.h") + _declared_functions("mij.h") + _declared_functions("mij_host.h")
    assert len(declared) >= 48
    assert "mjh_decode_batch" in declared
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, missing
    assert L.mij_abi_version() == 1


def test_info_matches_reference(golden, ica):
    for name in golden.names:
        ok, w, h, c = ica.stbi_info_from_memory(golden.jpg(name))
        want = golden[name + "/info"]
        assert bool(ok) == bool(want[0]), name
        if ok:
            assert (w, h, c) == tuple(int(v) for v in want[1:]), name
        else:
            assert ica.stbi_failure_reason() == "unknown image type"


def _dequantised_in_call_order(ica, desc, arena, progressive_order=False):
    """tile-layout planes -> the reference's IDCT call order (MCU-interleaved for multi-component
    baseline files; plane order for single-component and progressive files), de-quantised."""
    planes = ica.detile_coefficients(desc, arena)
    dq = [np.array(desc.dequant[desc.comp[i].tq][:], dtype=np.int32).reshape(8, 8) for i in range(desc.ncomp)]
    blocks = []
    if progressive_order or desc.ncomp == 1:
        for ci in range(desc.ncomp):
            cp = desc.comp[ci]
            for j in range((cp.y + 7) >> 3):
                for i in range((cp.x + 7) >> 3):
                    blocks.append((planes[ci][j, i].astype(np.int32) * dq[ci]).astype(np.int16))
    else:
        for my in range(desc.mcu_y):
            for mx in range(desc.mcu_x):
                for ci in range(desc.ncomp):
                    cp = desc.comp[ci]
                    for y in range(cp.v):
                        for x in range(cp.h):
                            blocks.append((planes[ci][my * cp.v + y, mx * cp.h + x].astype(np.int32) * dq[ci]).astype(np.int16))
    return np.stack(blocks).reshape(-1)


def test_host_entropy_stage_matches_reference_coefficients(golden, ica, oracle):
    n = 0
    for name in golden.names:
        if not golden.has(name + "/coef"):
            continue
        data = golden.jpg(name)
        desc, arena = ica.HostDecoder.decode(data, 0)
        prog = name.startswith("prog")
        if name == "pil_base_30x22":
            prog = False
        got = _dequantised_in_call_order(ica, desc, arena, progressive_order=prog)
        want = golden[name + "/coef"]
        assert got.shape == want.shape, (name, got.shape, want.shape)
        assert np.array_equal(got, want), name
        n += 1
    assert n >= 8


def test_host_stage_failure_reasons_match_reference(golden, ica):
    for name in golden.names:
        kind, want = golden.expect(name, 3)
        data = golden.jpg(name)
        if kind == "fail":
            with pytest.raises(ica.MijError) as e:
                ica.HostDecoder.decode(data, 3)
            assert str(e.value) == want, (name, str(e.value), want)
        else:
            ica.HostDecoder.decode(data, 3)
    with pytest.raises(ica.MijError) as e:
        ica.HostDecoder.decode(golden.jpg("b420_64x64_q90"), 5)
    assert str(e.value) == "bad req_comp"


def test_describe_follows_load_jpeg_image(golden, ica):
    # codec/jpeg.c:2241-2249 and the colour branches :2320-2431
    d = ica.HostDecoder.probe(golden.jpg("b420_64x64_q90"), 0)
    assert (d.ncomp, d.n_out, d.color) == (3, 3, 1)
    assert (d.comp[0].h, d.comp[0].v, d.comp[1].h, d.comp[1].v) == (2, 2, 1, 1)
    assert (d.mcu_x, d.mcu_y, d.comp[0].bw, d.comp[1].bw) == (4, 4, 8, 4)
    d = ica.HostDecoder.probe(golden.jpg("b420_64x64_q90"), 1)
    assert (d.n_out, d.color) == (1, 0)  # luma only
    d = ica.HostDecoder.probe(golden.jpg("grey_33x20"), 4)
    assert (d.ncomp, d.n_out, d.color) == (1, 4, 0)
    d = ica.HostDecoder.probe(golden.jpg("rgb_tagged_24x24"), 0)
    assert d.color == 2
    d = ica.HostDecoder.probe(golden.jpg("adobe_rgb_20x12"), 2)
    assert (d.color, d.n_out) == (2, 2)
    assert ica.HostDecoder.probe(golden.jpg("cmyk_transform0_40x30"), 0).color == 3
    assert ica.HostDecoder.probe(golden.jpg("cmyk_transform2_40x30"), 0).color == 4
    assert ica.HostDecoder.probe(golden.jpg("cmyk_transform1_40x30"), 0).color == 5
    d = ica.HostDecoder.probe(golden.jpg("s41_35x19"), 0)
    assert (d.h_max, d.v_max, d.comp[0].h, d.comp[1].h) == (4, 1, 4, 1)
    d = ica.HostDecoder.probe(golden.jpg("s12_35x19"), 0)
    assert (d.h_max, d.v_max, d.comp[0].v, d.comp[1].v) == (1, 2, 2, 1)


def test_tile_layout_index_math(ica):
    # include/mij.h: P = 8*col + rowslot[row]; plane index = (L>>6)<<12 | (P>>3)<<9 | (L&63)<<3 | P&7
    d = ica.ImageDesc()
    d.ncomp = 1
    d.comp[0].bw, d.comp[0].bh = 9, 9  # 81 blocks -> 2 tiles
    n = d.plane_elems(0)
    assert n == 2 * 4096
    arena = np.arange(n, dtype=np.int16)
    nat = ica.detile_coefficients(d, arena)[0]
    for (L, row, col) in [(0, 0, 0), (0, 4, 0), (5, 2, 3), (63, 7, 7), (64, 0, 0), (80, 1, 6)]:
        P = 8 * col + (0, 4, 2, 5, 1, 6, 3, 7)[row]
        idx = ((L >> 6) << 12) + ((P >> 3) << 9) + ((L & 63) << 3) + (P & 7)
        assert nat[L // 9, L % 9, row, col] == arena[idx]


def test_wide_idct_flag(golden, ica):
    # ordinary images never need the exact 32-bit second pass
    for name in ("b420_64x64_q90", "b444_64x64_q92", "prog_420_64x64", "grey_33x20"):
        d, _ = ica.HostDecoder.decode(golden.jpg(name), 3)
        assert d.flags == 0, name
    # a stream whose quantisation table is blown up to 65535 per step must raise it
    data = bytearray(golden.jpg("b420_64x64_q90"))
    i = bytes(data).index(b"\xff\xdb")
    for k in range(64):
        data[i + 5 + k] = 255
    d, _ = ica.HostDecoder.decode(bytes(data), 3)
    assert d.flags & 1


def test_writer_matches_reference_bytes(golden, ica):
    for nm in golden.enc_names:
        got = ica.stbi_write_jpg_to_memory(golden[nm + "/rgb"], int(golden[nm + "/q"][0]))
        assert got == bytes(golden[nm + "/jpg"]), nm
    assert ica.stbi_write_jpg_to_memory(np.zeros((0, 4, 3), np.uint8), 90) is None


def test_writer_vs_oracle_seeded(ica, oracle):
    rng = np.random.default_rng(7)
    for i in range(25):
        w, h, c = int(rng.integers(1, 70)), int(rng.integers(1, 50)), int(rng.choice([1, 2, 3, 4]))
        q = int(rng.choice([1, 25, 50, 90, 91, 100]))
        img = rng.integers(0, 256, (h, w, c)).astype(np.uint8)
        assert ica.stbi_write_jpg_to_memory(img, q) == oracle.encode(img, q), (w, h, c, q)


def test_host_stage_vs_oracle_fuzz(golden, ica, oracle):
    """Mutated entropy data: the product's host stage and the oracle agree on accept/reject, on the
    failure reason and on every de-quantised coefficient block."""
    n_ok = n_fail = 0
    for name in ("b420_64x64_q90", "b444_40x24_q95", "grey_33x20", "b422_37x21", "s41_35x19"):
        base = golden.jpg(name)
        for seed in range(40):
            data = helpers.mutate(base, seed * 104729 + len(name), allow_markers=(seed % 4 == 0))
            o = oracle.load(data, 0)
            if o[0] == "fail":
                with pytest.raises(ica.MijError) as e:
                    ica.HostDecoder.decode(data, 0)
                assert str(e.value) == o[1], (name, seed)
                n_fail += 1
            else:
                desc, arena = ica.HostDecoder.decode(data, 0)
                got = _dequantised_in_call_order(ica, desc, arena)
                assert np.array_equal(got, oracle.coef(data)), (name, seed)
                n_ok += 1
    assert n_ok > 50 and n_fail > 5


def test_synthetic_generator_is_deterministic(ica):
    a = ica.synth_rgb(37, 21, seed=3)
    # direct evaluation of the SURVEY 8d definition
    state = (12345 + 3) & 0xFFFFFFFF
    ref = np.zeros((21, 37, 3), np.uint8)
    for y in range(21):
        for x in range(37):
            state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
            n = (state >> 24) & 15
            ref[y, x] = (min(255, x * 255 // 37 + n), min(255, y * 255 // 21 + n), min(255, (x + y) * 255 // 58 + n))
    assert np.array_equal(a, ref)
    assert ica.synth_jpeg(64, 48, 1) == ica.synth_jpeg(64, 48, 1)


def test_decode_fails_loudly_without_gpu(golden, ica):
    if ica.gpu_available():
        pytest.skip("a GPU is present: the no-device behaviour is covered on the CPU box")
    assert ica.stbi_load_from_memory(golden.jpg("b420_64x64_q90"), 3) is None
    assert ica.stbi_failure_reason() == "no gpu device"
    with pytest.raises(ica.MijError):
        ica.Context()


def _walk_extracted_scan(scan, stream):
    """Plain sequential Huffman walk over what mjh_extract_scan hands to the GPU stage (tables as copied,
    unstuffed bytes): -> int16 [nblocks, 64] in zigzag order, DC predicted per component."""
    bits = int.from_bytes(stream + b"\0" * 16, "big")
    total = (len(stream) + 16) * 8

    def window(p):
        return (bits >> (total - p - 64)) & ((1 << 64) - 1)

    tabs = [(bytes(h.fast), bytes(h.size), bytes(h.values), list(h.maxcode), list(h.delta)) for h in scan.huff]

    def symbol(t, win):
        fast, size, values, maxcode, delta = t
        top16 = win >> 48
        k = fast[top16 >> 7]
        if k < 255:
            return values[k], size[k]
        n = 10
        while top16 >= maxcode[n]:
            n += 1
        return values[((top16 >> (16 - n)) & ((1 << n) - 1)) + delta[n]], n

    def extend(win, ln, n):
        v = (win >> (64 - ln - n)) & ((1 << n) - 1)
        return v if v >> (n - 1) else v - (1 << n) + 1

    bpm = scan.blocks_per_mcu
    out = np.zeros((scan.nblocks, 64), np.int64)
    table = np.frombuffer(stream, np.uint32, count=2 * max(1, scan.n_seg), offset=scan.seg_table_off).reshape(-1, 2)
    per_seg = scan.restart_mcus * bpm if scan.n_seg else scan.nblocks
    pred = [0, 0, 0, 0]
    p = 0
    slack = []
    for b in range(scan.nblocks):
        if b % per_seg == 0:  # a restart interval starts byte aligned with fresh predictors
            seg = b // per_seg
            if seg:
                slack.append(int(table[seg - 1, 0] + table[seg - 1, 1]) * 8 - p)
            p = int(table[seg, 0]) * 8
            pred = [0, 0, 0, 0]
        ci = scan.blk_comp[b % bpm]
        t, ln = symbol(tabs[scan.dc_tab[ci]], window(p))
        diff = extend(window(p), ln, t) if t else 0
        p += ln + t
        pred[ci] += diff
        out[b, 0] = pred[ci]
        k = 1
        while k < 64:
            rs, ln = symbol(tabs[scan.ac_tab[ci]], window(p))
            r, n = rs >> 4, rs & 15
            if n == 0:
                p += ln
                if rs != 0xF0:
                    break
                k += 16
                continue
            k += r
            out[b, k] = extend(window(p), ln, n)
            p += ln + n
            k += 1
    slack.append(int(table[-1, 0] + table[-1, 1]) * 8 - p)
    return out.astype(np.int16), max(slack)


def test_extracted_scan_walks_to_the_host_walk_coefficients(golden, ica):
    """What the GPU entropy stage is given (mjh_extract_scan: copied tables, unstuffed segment, block order)
    decodes, with a plain sequential walk, to exactly the host walk's coefficients; layouts outside its scope
    are declined, not guessed at."""
    import ctypes as C
    from image_codecs_amd.binding import GpuScan, lib
    L = lib()
    L.mjh_extract_scan.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(GpuScan), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_char_p)]
    zig = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50,
                    43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])
    cases = [golden.jpg("b420_64x64_q90"), golden.jpg("b444_40x24_q95"), golden.jpg("grey_33x20"), ica.synth_jpeg(97, 51, 3, 75), ica.synth_jpeg(16, 16, 4, 100),
             golden.jpg("big_b444_rst_250x130")]  # the last one carries restart markers: one segment per interval
    for data in cases:
        scan, n, why = GpuScan(), C.c_size_t(), C.c_char_p()
        buf = np.zeros(len(data) + 64, np.uint8)
        assert L.mjh_extract_scan(data, len(data), 3, C.byref(scan), buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(n), C.byref(why)) == 1
        got, slack = _walk_extracted_scan(scan, bytes(buf[:n.value]))
        assert 0 <= slack < 8  # only the byte-alignment padding is left in every segment
        desc, arena = ica.HostDecoder.decode(data, 3)
        planes = ica.detile_coefficients(desc, arena)  # [bh, bw, 8, 8] natural order
        bpm, mcu_x = scan.blocks_per_mcu, desc.mcu_x
        for b in range(scan.nblocks):
            m, c = divmod(b, bpm)
            ci = scan.blk_comp[c]
            bx = (m % mcu_x) * desc.comp[ci].h + scan.blk_dx[c]
            by = (m // mcu_x) * desc.comp[ci].v + scan.blk_dy[c]
            want = planes[ci][by, bx].reshape(64)[zig]
            assert np.array_equal(got[b], want), (b, ci, bx, by)
    for name in ("prog_420_64x64", "trunc_noeoi", "dri_without_rst"):
        d = golden.jpg(name)
        scan, n, why = GpuScan(), C.c_size_t(), C.c_char_p()
        buf = np.zeros(len(d) + 64, np.uint8)
        assert L.mjh_extract_scan(d, len(d), 3, C.byref(scan), buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(n), C.byref(why)) == 2, name


def test_host_stage_is_clean_under_sanitizers(golden, ica, tmp_path):
    """jpeg_entropy.c (probe, walk, scan extraction) built with -fsanitize=address,undefined and fed damaged files --
    entropy mutations with and without new markers, header bytes hit, truncations, insertions: no report."""
    import subprocess
    import helpers
    root = helpers.ROOT
    exe = str(tmp_path / "san_harness")
    cmd = ["gcc", "-std=gnu11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-I" + root + "/include", "-I" + root + "/image-codecs_amd/csrc", "-o", exe,
           root + "/tests/support/san_harness.c", root + "/image-codecs_amd/csrc/jpeg_entropy.c"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("no sanitizer runtime in this toolchain")
    assert build.returncode == 0, build.stderr
    bases = [golden.jpg(n) for n in golden.names if len(golden.jpg(n)) > 16]
    plan, du = ica.host_transform(ica.synth_rgb(97, 51, 1), 95)
    bases += [helpers.baseline_from_du(plan, du, 3, "native"), helpers.progressive_from_du(plan, du, 1)]
    rng = np.random.default_rng(1)
    files = []
    for k in range(400):
        b = bases[k % len(bases)]
        d = bytearray(b)
        mode = k % 4
        if mode == 0:
            d = bytearray(helpers.mutate(b, k, n_mut=1 + k % 5, allow_markers=True))
        elif mode == 1:
            for _ in range(1 + k % 4):
                d[int(rng.integers(0, len(d)))] = int(rng.integers(0, 256))
        elif mode == 2:
            d = d[:int(rng.integers(2, len(d)))]
        else:
            pos = int(rng.integers(0, len(d)))
            d[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 8))).astype(np.uint8))
        f = tmp_path / ("%04d.jpg" % k)
        f.write_bytes(bytes(d))
        files.append(str(f))
    run = subprocess.run([exe] + files, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert run.returncode == 0, run.stderr[-2000:]
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-2000:]
    assert "decoded ok" in run.stdout


@pytest.mark.skipif(not helpers.Reference.available(), reason="oracle/_ref not built (reference absent on this box)")
def test_host_writer_vs_live_reference_seeded(ica):
    """stbi_write_jpg_to_func (SSE transform: four 1-D passes per instruction in the reference's operation order; Huffman emission
    through a 64-bit accumulator) against the reference's writer itself: one to four channels, qualities on both sides of the 4:2:0 /
    4:4:4 switch, noise (many 0xFF bytes to stuff), flat and gradient pictures, sizes that are not multiples of the MCU."""
    ref = helpers.Reference()
    rng = np.random.default_rng(77)
    for i in range(150):
        w, h = int(rng.integers(1, 160)), int(rng.integers(1, 120))
        c = int(rng.choice([1, 2, 3, 4]))
        q = int(rng.choice([1, 10, 50, 75, 90, 91, 95, 100]))
        if i % 3 == 0:
            img = rng.integers(0, 256, (h, w, c)).astype(np.uint8)
        elif i % 3 == 1:
            img = np.full((h, w, c), int(rng.integers(0, 256)), np.uint8)
        else:
            img = np.clip(np.linspace(0, 255, w)[None, :, None] + rng.normal(0, 30, (h, w, c)), 0, 255).astype(np.uint8)
        assert ica.stbi_write_jpg_to_memory(img, q) == ref.encode(img, q), (w, h, c, q, i % 3)


def test_progressive_host_walk_vs_oracle_fuzz(golden, ica, oracle):
    """Progressive streams (DC / AC first and refinement scans, EOB runs, both scan scripts of the test-side writer, the reference-made
    PIL goldens), intact and with mutated entropy data: the product's host walk -- whose refinement scans take a combined code + sign-bit
    table since round 3 (build_fast_refine) -- and the oracle agree on accept / reject, on the failure reason and on every de-quantised
    coefficient.  Where the reference itself is available the oracle's verdict is compared with it on the way."""
    rng = np.random.default_rng(23)
    bases = [golden.jpg(n) for n in golden.names if n.startswith("prog")]
    for (w, h, q, script, kind) in ((97, 51, 95, 1, "noise"), (64, 64, 92, 2, "noise"), (120, 88, 100, 1, "noise"), (200, 120, 95, 2, "synth"), (33, 17, 91, 1, "synth")):
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8) if kind == "noise" else ica.synth_rgb(w, h, w)
        plan, du = ica.host_transform(img, q)
        bases.append(helpers.progressive_from_du(plan, du, script))
    ref = helpers.Reference() if helpers.Reference.available() else None
    n_ok = n_fail = 0
    for bi, base in enumerate(bases):
        for seed in range(-1, 36):
            if seed >= 30:  # cut short inside the entropy data, with and without an EOI behind the cut: the bit register runs dry inside a refinement scan
                cut = int(rng.integers(len(base) // 3, len(base) - 2))
                data = base[:cut] + (b"\xff\xd9" if seed % 2 else b"")
            else:
                data = base if seed < 0 else helpers.mutate(base, seed * 7919 + bi, n_mut=1 + seed % 4, allow_markers=(seed % 5 == 0))
            o = oracle.load(data, 0)
            if ref is not None:
                r = ref.load(data, 0)
                # a reason of None: the reference fails there without calling stbi__err (its stbi_failure_reason() keeps whatever an earlier call left)
                assert r[0] == o[0] and (r[0] == "fail" and (r[1] == o[1] or o[1] is None) or r[0] == "ok"), (bi, seed, r[:2] if r[0] == "fail" else "ok", o[:2] if o[0] == "fail" else "ok")
            if o[0] == "fail":
                with pytest.raises(ica.MijError) as e:
                    ica.HostDecoder.decode(data, 0)
                assert str(e.value) == (o[1] if o[1] is not None else "decode failed"), (bi, seed)
                n_fail += 1
            else:
                desc, arena = ica.HostDecoder.decode(data, 0)
                got = _dequantised_in_call_order(ica, desc, arena, progressive_order=True)
                assert np.array_equal(got, oracle.coef(data)), (bi, seed)
                n_ok += 1
    assert n_ok > 150 and n_fail > 3, (n_ok, n_fail)


def test_progressive_refinement_without_pdep():
    """The AC refinement scans read their correction bits a run at a time where the CPU has BMI2 pdep (jpeg_entropy.c, refine_symbols_wide)
    and bit by bit elsewhere; MIJ_NO_PDEP=1 (read once, when the library initialises its tables) forces the second form: the same fuzz, in
    a process of its own, must come out the same."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, MIJ_NO_PDEP="1")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(here, "test_host_cpu.py") + "::test_progressive_host_walk_vs_oracle_fuzz",
                        os.path.join(here, "test_progressive_writer.py")], env=env, cwd=os.path.dirname(here), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_third_pair_of_huffman_tables_host_walk(ica, oracle):
    """A baseline file may name table ids up to 3 (codec/jpeg.c:1016-1040, :1070-1085); helpers.third_tables rewrites the writer's streams to
    use a third DC / AC pair for the last component.  The oracle (and the reference, where present) decode them to the pixels of the
    original stream, and the product's host walk stages the oracle's coefficients."""
    ref = helpers.Reference() if helpers.Reference.available() else None
    for i, (w, h, q) in enumerate(((64, 48, 90), (200, 120, 75), (333, 77, 95))):
        base = ica.synth_jpeg(w, h, i, q)
        data = helpers.third_tables(base)
        kind, want, _ = oracle.load(data, 3)
        assert kind == "ok" and np.array_equal(want, oracle.load(base, 3)[1])
        if ref is not None:
            assert np.array_equal(ref.load(data, 3)[1], want)
        desc, arena = ica.HostDecoder.decode(data, 0)
        assert np.array_equal(_dequantised_in_call_order(ica, desc, arena), oracle.coef(data)), (w, h, q)
