#!/usr/bin/env python3
"""Generates tests/golden/jpeg_golden.npz from the REAL reference.

Runs only in the build container: it needs oracle/_ref/libstbref.so (the reference compiled in
place from /root/reference by oracle/Makefile) and Pillow (to synthesise stream kinds the
reference's own writer cannot produce: progressive, 4:2:2, 4:1:1, restart markers, grey, CMYK).
The reference ships no tests, fixtures or golden vectors of its own, so these
vectors -- inputs plus the outputs the reference itself produced here -- are what pins parity on
the GPU box, where neither the reference nor Pillow exists.

Everything stored is DATA: input byte strings and the reference's outputs (pixels, coefficient
dumps, encoder byte streams, failure reasons).  No reference source text is stored.

Layout of the .npz (numpy arrays only, loadable with allow_pickle=False):
  names                  "\n"-joined case names
  <name>/jpg             uint8  the input file
  <name>/out<r>          uint8  [h, w, n] pixels the reference returned for req_comp r (0..4)
  <name>/fail<r>         uint8  ascii failure reason when the reference returned NULL
  <name>/info            int32  [ok, w, h, comp] from stbi_info_from_memory
  <name>/coef            int16  de-quantised blocks in the reference's IDCT call order (some cases)
  idct/in, idct/out      int16 [k,64] / uint8 [k,64]  known-answer vectors through stbi__idct_block
  resample/<kind>/...    rows through the four resamplers
  ycc/in, ycc/out3, ycc/out4   stbi__YCbCr_to_RGB_row vectors
  enc/<name>/rgb|q|jpg   encoder inputs and the byte stream stbi_write_jpg_to_func produced
"""
import ctypes as C
import io
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libstbref.so"))

P_INT = C.POINTER(C.c_int)
REF.stbi_load_from_memory.restype = C.POINTER(C.c_ubyte)
REF.stbi_load_from_memory.argtypes = [C.c_char_p, C.c_int, P_INT, P_INT, P_INT, C.c_int]
REF.stbi_info_from_memory.argtypes = [C.c_char_p, C.c_int, P_INT, P_INT, P_INT]
REF.stbi_failure_reason.restype = C.c_char_p
REF.stbi_image_free.argtypes = [C.c_void_p]
REF.ref_encode.restype = C.c_long
REF.ref_encode.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
REF.ref_decode_capture.restype = C.POINTER(C.c_ubyte)
REF.ref_decode_capture.argtypes = [C.c_char_p, C.c_int, P_INT, P_INT, P_INT, C.c_int, C.c_void_p, C.c_long, C.POINTER(C.c_long)]
REF.ref_idct_block.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
REF.ref_resample_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
REF.ref_ycbcr_to_rgb_row.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]


def ref_load(data, req):
    x, y, c = C.c_int(), C.c_int(), C.c_int()
    p = REF.stbi_load_from_memory(data, len(data), x, y, c, req)
    if not p:
        return None, REF.stbi_failure_reason().decode()
    n = req if req else c.value
    a = np.ctypeslib.as_array(p, shape=(y.value * x.value * n,)).reshape(y.value, x.value, n).copy()
    REF.stbi_image_free(p)
    return a, c.value


def ref_info(data):
    x, y, c = C.c_int(), C.c_int(), C.c_int()
    ok = REF.stbi_info_from_memory(data, len(data), x, y, c)
    return np.array([ok, x.value, y.value, c.value], dtype=np.int32)


def ref_coef(data, req=0):
    cap = np.zeros(1 << 22, dtype=np.int16)
    n = C.c_long()
    x, y, c = C.c_int(), C.c_int(), C.c_int()
    p = REF.ref_decode_capture(data, len(data), x, y, c, req, cap.ctypes.data, cap.size, C.byref(n))
    if p:
        REF.stbi_image_free(p)
    return cap[: n.value].copy()


def ref_encode(img, q):
    h, w, c = img.shape
    buf = np.zeros(w * h * 4 + 8192, np.uint8)
    n = REF.ref_encode(buf.ctypes.data, buf.size, w, h, c, np.ascontiguousarray(img).ctypes.data, q)
    assert 0 < n <= buf.size
    return bytes(buf[:n])


def test_image(w, h, seed, mode="RGB"):
    """smooth gradients + texture + a few hard edges: exercises every coefficient band"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 255) // max(w - 1, 1), (yy * 255) // max(h - 1, 1), ((xx + yy) * 255) // max(w + h - 2, 1)], -1).astype(np.int32)
    tex = rng.integers(-24, 25, (h, w, 3))
    img = np.clip(base + tex, 0, 255)
    if w > 8 and h > 8:
        img[h // 3: h // 3 + 3, :, :] = 255 - img[h // 3: h // 3 + 3, :, :]
        img[:, w // 2: w // 2 + 2, :] = rng.integers(0, 2, 3) * 255
    img = img.astype(np.uint8)
    if mode == "L":
        return img[:, :, 0]
    if mode == "CMYK":
        return np.concatenate([img, rng.integers(0, 256, (h, w, 1)).astype(np.uint8)], -1)
    return img


def pil_jpeg(arr, mode="RGB", **kw):
    bio = io.BytesIO()
    Image.fromarray(arr, mode).save(bio, "JPEG", **kw)
    return bio.getvalue()


def patch(data, find, repl, which=0):
    idx = -1
    for _ in range(which + 1):
        idx = data.index(find, idx + 1)
    return data[:idx] + repl + data[idx + len(find):]


def main():
    out = {}
    names = []

    def add_case(name, data, coef=False):
        names.append(name)
        out[name + "/jpg"] = np.frombuffer(data, dtype=np.uint8)
        out[name + "/info"] = ref_info(data)
        for req in range(5):
            a, extra = ref_load(data, req)
            if a is None:
                out["%s/fail%d" % (name, req)] = np.frombuffer(extra.encode(), dtype=np.uint8)
            else:
                out["%s/out%d" % (name, req)] = a
        if coef:
            out[name + "/coef"] = ref_coef(data)

    # (1) baseline 4:2:0 from the reference's own writer
    for (w, h, q) in [(1, 1, 90), (2, 3, 90), (17, 33, 75), (33, 17, 50), (64, 64, 90), (16, 16, 90), (15, 15, 90), (31, 47, 60), (8, 8, 10), (100, 20, 90)]:
        add_case("b420_%dx%d_q%d" % (w, h, q), ref_encode(test_image(w, h, w * 1000 + h), q), coef=(w <= 33))
    # (2) baseline 4:4:4 (quality > 90)
    for (w, h, q) in [(40, 24, 95), (9, 7, 100), (64, 64, 92)]:
        add_case("b444_%dx%d_q%d" % (w, h, q), ref_encode(test_image(w, h, w * 7 + h), q), coef=(w <= 40))
    # (3) other sampling layouts via libjpeg
    for (w, h) in [(2, 2), (4, 5), (37, 21), (256, 64), (1, 7)]:
        add_case("b422_%dx%d" % (w, h), pil_jpeg(test_image(w, h, w + h), quality=85, subsampling="4:2:2"), coef=(w == 37))
    # Sampling layouts no writer here produces (luma 4x1, 1x4, 1x2): re-tag the luma sampling byte of
    # a stream with the same number of luma blocks per MCU.  The picture is scrambled but the stream
    # stays decodable, and it drives the reference's generic / v_2 resamplers (codec/jpeg.c:2282-2289).
    def retag_luma(data, samp):
        sof = data.index(b"\xff\xc0")
        return data[: sof + 11] + bytes([samp]) + data[sof + 12:]
    for (w, h) in [(35, 19), (64, 16), (3, 3)]:
        src420 = pil_jpeg(test_image(w, h, w * 3 + h), quality=85, subsampling="4:2:0")
        add_case("s41_%dx%d" % (w, h), retag_luma(src420, 0x41))
        add_case("s14_%dx%d" % (w, h), retag_luma(src420, 0x14))
        src422 = pil_jpeg(test_image(w, h, w * 5 + h), quality=85, subsampling="4:2:2")
        add_case("s12_%dx%d" % (w, h), retag_luma(src422, 0x12))
    for (w, h) in [(33, 33), (130, 50)]:
        add_case("pil420_%dx%d" % (w, h), pil_jpeg(test_image(w, h, 5 * w + h), quality=80, subsampling="4:2:0", optimize=True))
    # 4:4:0 (h1 v2) by re-tagging the sampling factors of a 4:2:2 stream is not a valid stream; use a
    # hand-built one instead: PIL cannot write it, so take 4:2:0 luma-only variants further below.
    # (4) progressive
    for (w, h, ss) in [(64, 64, "4:4:4"), (64, 64, "4:2:0"), (23, 41, "4:2:0"), (50, 30, "4:2:2"), (8, 8, "4:4:4")]:
        add_case("prog_%s_%dx%d" % (ss.replace(":", ""), w, h), pil_jpeg(test_image(w, h, w * 11 + h), quality=88, subsampling=ss, progressive=True),
                 coef=(w == 23))
    add_case("prog_grey_40x40", pil_jpeg(test_image(40, 40, 77, "L"), "L", quality=90, progressive=True))
    # (5) restart intervals
    add_case("rst_blocks_64x48", pil_jpeg(test_image(64, 48, 5), quality=85, subsampling="4:2:0", restart_marker_blocks=3))
    add_case("rst_rows_70x40_444", pil_jpeg(test_image(70, 40, 6), quality=85, subsampling="4:4:4", restart_marker_rows=1))
    add_case("rst_prog_48x48", pil_jpeg(test_image(48, 48, 8), quality=85, subsampling="4:2:0", progressive=True, restart_marker_blocks=2))
    # (6) colour layouts
    add_case("grey_33x20", pil_jpeg(test_image(33, 20, 9, "L"), "L", quality=90), coef=True)
    add_case("grey_1x1", pil_jpeg(test_image(1, 1, 9, "L"), "L", quality=90))
    cmyk = pil_jpeg(test_image(40, 30, 10, "CMYK"), "CMYK", quality=90)
    add_case("cmyk_40x30", cmyk)
    # Adobe APP14: 'Adobe\0' ver(2) flags0(2) flags1(2) transform(1); flip the transform byte
    ai = cmyk.index(b"Adobe")
    tpos = ai + 5 + 2 + 2 + 2
    for t in (0, 1, 2):
        add_case("cmyk_transform%d_40x30" % t, cmyk[:tpos] + bytes([t]) + cmyk[tpos + 1:])
    base444 = ref_encode(test_image(24, 24, 12), 95)
    # component ids 1,2,3 -> 'R','G','B' in SOF0 and SOS => z->rgb == 3 (codec/jpeg.c:1587)
    sof = base444.index(b"\xff\xc0")
    rgb_tagged = bytearray(base444)
    for k, ch in enumerate(b"RGB"):
        rgb_tagged[sof + 10 + 3 * k] = ch
    sos = bytes(rgb_tagged).index(b"\xff\xda")
    for k, ch in enumerate(b"RGB"):
        rgb_tagged[sos + 5 + 2 * k] = ch
    add_case("rgb_tagged_24x24", bytes(rgb_tagged))
    # Adobe transform 0 with 3 components and no JFIF => is_rgb too (codec/jpeg.c:2244)
    adobe_rgb = pil_jpeg(test_image(20, 12, 13), quality=90, subsampling="4:4:4")
    if b"JFIF" in adobe_rgb:
        j0 = adobe_rgb.index(b"\xff\xe0")
        jl = (adobe_rgb[j0 + 2] << 8) + adobe_rgb[j0 + 3]
        app14 = b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x00"
        adobe_rgb = adobe_rgb[:j0] + app14 + adobe_rgb[j0 + 2 + jl:]
    add_case("adobe_rgb_20x12", adobe_rgb)
    # (8) damaged / odd streams
    good = ref_encode(test_image(48, 32, 14), 90)
    add_case("trunc_noeoi", good[: len(good) * 2 // 3])
    add_case("trunc_eoi", good[: len(good) * 2 // 3] + b"\xff\xd9")
    add_case("trunc_header", good[:100])
    add_case("empty", b"")
    add_case("garbage", bytes(range(256)) * 2)
    add_case("soi_only", b"\xff\xd8")
    add_case("padded_tail", good[:-2] + b"\x00" * 37 + b"\xff\xd9")
    add_case("fill_bytes", good.replace(b"\xff\xdb", b"\xff\xff\xff\xdb", 1))
    add_case("com_segment", good[:2] + b"\xff\xfe\x00\x07hello" + good[2:])
    add_case("bad_com_len", good[:2] + b"\xff\xfe\x00\x01" + good[2:])
    dri = good[: good.index(b"\xff\xda")] + b"\xff\xdd\x00\x04\x00\x02" + good[good.index(b"\xff\xda"):]
    add_case("dri_without_rst", dri)  # restart interval announced, no RST in the data: scan stops early (:1184)
    add_case("sixteen_bit_dqt", patch(good, b"\xff\xdb\x00\x84\x00", b"\xff\xdb\x00\x84\x00"))
    add_case("twelve_bit", patch(good, b"\xff\xc0\x00\x11\x08", b"\xff\xc0\x00\x11\x0c"))
    add_case("zero_height", good[: good.index(b"\xff\xc0") + 5] + b"\x00\x00" + good[good.index(b"\xff\xc0") + 7:])
    add_case("bad_ncomp", good[: good.index(b"\xff\xc0") + 9] + b"\x02" + good[good.index(b"\xff\xc0") + 10:])
    add_case("unknown_marker", good[:2] + b"\xff\xc9\x00\x02" + good[2:])
    add_case("dnl_ok", good[: good.index(b"\xff\xda")] + b"\xff\xdc\x00\x04" + bytes([0, 32]) + good[good.index(b"\xff\xda"):])
    # DNL / second table definitions between scans of a non-interleaved baseline file
    nonint = pil_jpeg(test_image(30, 22, 15), quality=85, subsampling="4:2:0", progressive=False, optimize=False)
    add_case("pil_base_30x22", nonint, coef=True)

    # (8b) larger libjpeg-made streams (the GPU box has no Pillow): only req_comp 3 is stored
    def add_big(name, data):
        names.append(name)
        out[name + "/jpg"] = np.frombuffer(data, dtype=np.uint8)
        out[name + "/info"] = ref_info(data)
        a, _ = ref_load(data, 3)
        out[name + "/out3"] = a
    add_big("big_prog_444_256x256", pil_jpeg(test_image(256, 256, 21), quality=90, subsampling="4:4:4", progressive=True))
    add_big("big_prog_420_320x200", pil_jpeg(test_image(320, 200, 22), quality=85, subsampling="4:2:0", progressive=True))
    add_big("big_b422_320x240", pil_jpeg(test_image(320, 240, 23), quality=85, subsampling="4:2:2"))
    add_big("big_b444_rst_250x130", pil_jpeg(test_image(250, 130, 24), quality=92, subsampling="4:4:4", restart_marker_rows=2))

    # (9) known-answer vectors for the individual stages
    rng = np.random.default_rng(99)
    blocks = []
    z = np.zeros(64, np.int16)
    for dc in (0, 1, -1, 8, -8, 1016, -1024, 2047, -2048, 32767, -32768):
        b = z.copy(); b[0] = dc; blocks.append(b)
    for k in range(1, 64):
        for v in (64, -300):
            b = z.copy(); b[k] = v; blocks.append(b)
    for amp in (4, 32, 256, 1024, 4096, 32767):
        for _ in range(24):
            blocks.append(rng.integers(-amp, amp + 1, 64).astype(np.int16))
    for _ in range(64):  # realistic: decaying spectrum
        b = (rng.standard_normal(64) * 400 / (1 + np.arange(64))).astype(np.int16); b[0] = rng.integers(-1024, 1024); blocks.append(b)
    blocks = np.stack(blocks).astype(np.int16)
    outs = np.zeros((len(blocks), 64), np.uint8)
    for i in range(len(blocks)):
        tmp = np.ascontiguousarray(blocks[i])
        o = np.zeros(64, np.uint8)
        REF.ref_idct_block(o.ctypes.data, 8, tmp.ctypes.data)
        outs[i] = o
    out["idct/in"] = blocks
    out["idct/out"] = outs
    for kind, kname in ((1, "v2"), (2, "h2"), (3, "hv2"), (4, "generic3")):
        for w in (1, 2, 3, 8, 17):
            near = rng.integers(0, 256, w).astype(np.uint8)
            far = rng.integers(0, 256, w).astype(np.uint8)
            hs = 3 if kind == 4 else 2
            o = np.zeros(w * 4 + 8, np.uint8)
            n = REF.ref_resample_row(kind, o.ctypes.data, near.ctypes.data, far.ctypes.data, w, hs)
            out["resample/%s/%d/near" % (kname, w)] = near
            out["resample/%s/%d/far" % (kname, w)] = far
            out["resample/%s/%d/out" % (kname, w)] = o[:n].copy()
    # all (y, cb, cr) triples on a coarse lattice plus the extremes
    vals = np.array(sorted(set(list(range(0, 256, 12)) + [255, 1, 127, 128, 129])), np.uint8)
    yy, bb, rr = np.meshgrid(vals, vals, vals, indexing="ij")
    yv, bv, rv = [np.ascontiguousarray(a.reshape(-1)) for a in (yy, bb, rr)]
    o3 = np.zeros(len(yv) * 3 + 4, np.uint8)
    o4 = np.zeros(len(yv) * 4 + 4, np.uint8)
    REF.ref_ycbcr_to_rgb_row(o3.ctypes.data, yv.ctypes.data, bv.ctypes.data, rv.ctypes.data, len(yv), 3)
    REF.ref_ycbcr_to_rgb_row(o4.ctypes.data, yv.ctypes.data, bv.ctypes.data, rv.ctypes.data, len(yv), 4)
    out["ycc/in"] = np.stack([yv, bv, rv], -1)
    out["ycc/out3"] = o3[: len(yv) * 3].reshape(-1, 3).copy()
    out["ycc/out4"] = o4[: len(yv) * 4].reshape(-1, 4).copy()

    # (10) encoder
    enc_names = []
    for (w, h, c, q) in [(16, 16, 3, 90), (17, 9, 3, 90), (64, 64, 3, 95), (33, 31, 3, 50), (5, 7, 1, 90), (20, 20, 2, 100), (12, 40, 4, 75), (1, 1, 3, 1), (8, 8, 3, 0)]:
        img = test_image(w, h, w * 13 + h)
        if c == 1:
            img = img[:, :, :1]
        elif c == 2:
            img = img[:, :, :2]
        elif c == 4:
            img = np.concatenate([img, img[:, :, :1]], -1)
        img = np.ascontiguousarray(img)
        nm = "enc/%dx%dx%d_q%d" % (w, h, c, q)
        enc_names.append(nm)
        out[nm + "/rgb"] = img
        out[nm + "/q"] = np.array([q], np.int32)
        out[nm + "/jpg"] = np.frombuffer(ref_encode(img, q), dtype=np.uint8)
    out["names"] = np.frombuffer("\n".join(names).encode(), dtype=np.uint8)
    out["enc_names"] = np.frombuffer("\n".join(enc_names).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "jpeg_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d decode cases, %d encoder cases, %.1f KiB" % (path, len(names), len(enc_names), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    sys.exit(main())
