"""Shared test plumbing: golden-vector access and ctypes views of the checkers under oracle/.

Only tests (and smoke / bench's cpu_baseline leg) may touch oracle/; the product never does."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "jpeg_golden.npz")
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libstbref.so")

P_INT = C.POINTER(C.c_int)


class Golden:
    def __init__(self, z):
        self.z = z
        self.names = bytes(z["names"]).decode().split("\n")
        self.enc_names = bytes(z["enc_names"]).decode().split("\n")

    def jpg(self, name):
        return bytes(self.z[name + "/jpg"])

    def expect(self, name, req):
        """-> ('ok', pixels) or ('fail', reason)"""
        k = "%s/out%d" % (name, req)
        if k in self.z:
            return "ok", self.z[k]
        k = "%s/fail%d" % (name, req)
        if k in self.z:
            return "fail", bytes(self.z[k]).decode()
        return "skip", None  # the larger fixtures only store req_comp 3

    def has(self, key):
        return key in self.z

    def __getitem__(self, key):
        return self.z[key]


def load_golden():
    return Golden(np.load(GOLDEN, allow_pickle=False))


def build_oracle():
    if not os.path.exists(ORACLE_SO):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True)
    return ORACLE_SO


class Oracle:
    """oracle/_build/liboracle.so -- our CPU restatement (the checker)."""

    def __init__(self):
        L = C.CDLL(build_oracle())
        L.orc_load_from_memory.restype = C.POINTER(C.c_ubyte)
        L.orc_load_from_memory.argtypes = [C.c_char_p, C.c_int, P_INT, P_INT, P_INT, C.c_int, C.POINTER(C.c_char_p)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_info_from_memory.argtypes = [C.c_char_p, C.c_int, P_INT, P_INT, P_INT]
        L.orc_idct_block.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_resample_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_ycbcr_to_rgb_row.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_decode_capture.restype = C.POINTER(C.c_ubyte)
        L.orc_decode_capture.argtypes = [C.c_char_p, C.c_int, P_INT, P_INT, P_INT, C.c_int, C.c_void_p, C.c_long, C.POINTER(C.c_long)]
        L.orc_encode.restype = C.c_long
        L.orc_encode.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        self.L = L

    def load(self, data, req=0):
        """-> ('ok', pixels, comp) or ('fail', reason, None)"""
        x, y, c = C.c_int(), C.c_int(), C.c_int()
        why = C.c_char_p()
        p = self.L.orc_load_from_memory(bytes(data), len(data), x, y, c, req, C.byref(why))
        if not p:
            return "fail", (why.value.decode() if why.value else None), None
        n = req if req else c.value
        a = np.ctypeslib.as_array(p, shape=(y.value * x.value * n,)).reshape(y.value, x.value, n).copy()
        self.L.orc_free(p)
        return "ok", a, c.value

    def info(self, data):
        x, y, c = C.c_int(), C.c_int(), C.c_int()
        ok = self.L.orc_info_from_memory(bytes(data), len(data), x, y, c)
        return ok, x.value, y.value, c.value

    def idct(self, block):
        b = np.ascontiguousarray(block, dtype=np.int16)
        o = np.zeros(64, np.uint8)
        self.L.orc_idct_block(o.ctypes.data, 8, b.ctypes.data)
        return o

    def resample(self, kind, near, far, hs):
        near = np.ascontiguousarray(near, np.uint8)
        far = np.ascontiguousarray(far, np.uint8)
        o = np.zeros(len(near) * 4 + 8, np.uint8)
        n = self.L.orc_resample_row(kind, o.ctypes.data, near.ctypes.data, far.ctypes.data, len(near), hs)
        return o[:n].copy()

    def ycc(self, y, cb, cr, step):
        y, cb, cr = [np.ascontiguousarray(a, np.uint8) for a in (y, cb, cr)]
        o = np.zeros(len(y) * step + 4, np.uint8)
        self.L.orc_ycbcr_to_rgb_row(o.ctypes.data, y.ctypes.data, cb.ctypes.data, cr.ctypes.data, len(y), step)
        return o[: len(y) * step].reshape(-1, step).copy()

    def coef(self, data, req=0):
        cap = np.zeros(1 << 22, dtype=np.int16)
        n = C.c_long()
        x, y, c = C.c_int(), C.c_int(), C.c_int()
        p = self.L.orc_decode_capture(bytes(data), len(data), x, y, c, req, cap.ctypes.data, cap.size, C.byref(n))
        if p:
            self.L.orc_free(p)
        return cap[: n.value].copy()

    def encode(self, img, q):
        img = np.ascontiguousarray(img, np.uint8)
        h, w, c = img.shape
        buf = np.zeros(w * h * 4 + 8192, np.uint8)
        n = self.L.orc_encode(buf.ctypes.data, buf.size, w, h, c, img.ctypes.data, q)
        if n < 0:
            return None
        return bytes(buf[:n])


class Reference:
    """oracle/_ref/libstbref.so -- the real reference compiled in place (build container only)."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self):
        L = C.CDLL(REF_SO)
        L.stbi_load_from_memory.restype = C.POINTER(C.c_ubyte)
        L.stbi_load_from_memory.argtypes = [C.c_char_p, C.c_int, P_INT, P_INT, P_INT, C.c_int]
        L.stbi_failure_reason.restype = C.c_char_p
        L.stbi_image_free.argtypes = [C.c_void_p]
        L.ref_encode.restype = C.c_long
        L.ref_encode.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.ref_idct_block.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self.L = L

    def load(self, data, req=0):
        x, y, c = C.c_int(), C.c_int(), C.c_int()
        p = self.L.stbi_load_from_memory(bytes(data), len(data), x, y, c, req)
        if not p:
            return "fail", self.L.stbi_failure_reason().decode(), None
        n = req if req else c.value
        a = np.ctypeslib.as_array(p, shape=(y.value * x.value * n,)).reshape(y.value, x.value, n).copy()
        self.L.stbi_image_free(p)
        return "ok", a, c.value

    def encode(self, img, q):
        img = np.ascontiguousarray(img, np.uint8)
        h, w, c = img.shape
        buf = np.zeros(w * h * 4 + 8192, np.uint8)
        n = self.L.ref_encode(buf.ctypes.data, buf.size, w, h, c, img.ctypes.data, q)
        return None if n < 0 else bytes(buf[:n])

    def idct(self, block):
        b = np.ascontiguousarray(block, dtype=np.int16)
        o = np.zeros(64, np.uint8)
        self.L.ref_idct_block(o.ctypes.data, 8, b.ctypes.data)
        return o


def entropy_ranges(data):
    """[(start, end)) byte ranges of the entropy-coded segments of a JPEG (after each SOS header)."""
    out = []
    i = 2
    n = len(data)
    while i + 4 <= n:
        if data[i] != 0xFF:
            i += 1
            continue
        m = data[i + 1]
        if m == 0xFF:
            i += 1
            continue
        if m == 0xD9:
            break
        if m == 0x01 or 0xD0 <= m <= 0xD7:
            i += 2
            continue
        seglen = (data[i + 2] << 8) + data[i + 3]
        i += 2 + seglen
        if m == 0xDA:
            start = i
            while i + 1 < n and not (data[i] == 0xFF and data[i + 1] != 0x00 and not (0xD0 <= data[i + 1] <= 0xD7)):
                i += 1
            out.append((start, i))
    return out


def mutate(data, seed, n_mut=3, allow_markers=False):
    """Deterministic in-place byte mutations confined to the entropy-coded segments: the headers and
    tables stay intact, so both decoders always work from defined tables (the reference reads
    uninitialised tables otherwise -- undefined behaviour no restatement can match).  Unless
    allow_markers, no 0xFF byte is created or destroyed, so the marker structure is preserved too."""
    rng = np.random.default_rng(seed)
    b = bytearray(data)
    ranges = [r for r in entropy_ranges(data) if r[1] - r[0] > 4]
    if not ranges:
        return bytes(b)
    for _ in range(n_mut):
        s, e = ranges[int(rng.integers(0, len(ranges)))]
        for _try in range(32):
            pos = int(rng.integers(s, e))
            new = int(rng.integers(0, 256)) if rng.integers(0, 2) else b[pos] ^ (1 << int(rng.integers(0, 8)))
            if allow_markers:
                if pos + 1 < e:
                    b[pos] = new
                    break
                continue
            if b[pos] == 0xFF or new == 0xFF or (pos > s and b[pos - 1] == 0xFF):
                continue
            b[pos] = new
            break
    return bytes(b)


# ---------------------------------------------------------------- progressive test-stream writer

PROGW_SRC = os.path.join(ROOT, "tests", "support", "prog_writer.c")
PROGW_SO = os.path.join(ROOT, "tests", "support", "libprogwriter.so")


def build_prog_writer():
    if not os.path.exists(PROGW_SO) or os.path.getmtime(PROGW_SO) < os.path.getmtime(PROGW_SRC):
        subprocess.run(["gcc", "-O2", "-std=gnu99", "-shared", "-fPIC", "-o", PROGW_SO, PROGW_SRC], check=True)
    return PROGW_SO


def du_to_planes(plan, du):
    """The writer's data units (MCU order: Y.. U V, zigzag) -> per-component planes [bh][bw][64]."""
    du = np.asarray(du, dtype=np.int16).reshape(plan.mcu_y, plan.mcu_x, plan.du_per_mcu, 64)
    if plan.du_per_mcu == 6:
        y = du[:, :, :4].reshape(plan.mcu_y, plan.mcu_x, 2, 2, 64).transpose(0, 2, 1, 3, 4).reshape(plan.mcu_y * 2, plan.mcu_x * 2, 64)
        return [np.ascontiguousarray(y), np.ascontiguousarray(du[:, :, 4]), np.ascontiguousarray(du[:, :, 5])], (2, 1, 1)
    if plan.du_per_mcu == 3:
        return [np.ascontiguousarray(du[:, :, c]) for c in range(3)], (1, 1, 1)
    raise ValueError("unsupported data-unit layout")


def progressive_from_du(plan, du, script=1):
    """Re-emit a writer plan's quantised data units as a progressive (SOF2) stream: same coefficients as
    the baseline stream emit_jpeg(plan, du) carries, so both must decode to identical pixels."""
    L = C.CDLL(build_prog_writer())
    L.pw_write_progressive.restype = C.c_long
    L.pw_write_progressive.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, P_INT, P_INT, C.c_void_p, C.c_int, C.c_void_p, C.c_long]
    planes, samp = du_to_planes(plan, du)
    ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
    hs = (C.c_int * 3)(*samp)
    vs = (C.c_int * 3)(*samp)
    qt = np.concatenate([np.frombuffer(bytes(plan.ytab), np.uint8), np.frombuffer(bytes(plan.ctab), np.uint8)])
    cap = 4096 + sum(p.size for p in planes) * 3  # headers and ten scans' tables dominate tiny pictures
    out = np.empty(cap, np.uint8)
    n = L.pw_write_progressive(ptrs, 3, plan.width, plan.height, hs, vs, qt.ctypes.data_as(C.c_void_p), int(script), out.ctypes.data_as(C.c_void_p), cap)
    assert 0 < n <= cap, n
    return out[:n].tobytes()


def progressive_422_from_444(plan, du, script=1):
    """A 4:2:2 (h2v1) progressive stream from the 4:4:4 data units of the same picture size: full luma,
    every other chroma block column.  The chroma content is not a faithful down-sampling -- only a valid
    stream of that layout is needed (no libjpeg on the GPU box); parity is always against the oracle's
    decode of the same bytes."""
    assert plan.du_per_mcu == 3
    L = C.CDLL(build_prog_writer())
    L.pw_write_progressive.restype = C.c_long
    L.pw_write_progressive.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, P_INT, P_INT, C.c_void_p, C.c_int, C.c_void_p, C.c_long]
    planes444, _ = du_to_planes(plan, du)
    mcu_x, mcu_y = (plan.width + 15) // 16, (plan.height + 7) // 8
    y = np.zeros((mcu_y, 2 * mcu_x, 64), np.int16)
    y[:, :planes444[0].shape[1]] = planes444[0]
    planes = [y]
    for c in (1, 2):
        sub = planes444[c][:, 0::2]
        p = np.zeros((mcu_y, mcu_x, 64), np.int16)
        p[:, :sub.shape[1]] = sub
        planes.append(p)
    ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in planes])
    hs = (C.c_int * 3)(2, 1, 1)
    vs = (C.c_int * 3)(1, 1, 1)
    qt = np.concatenate([np.frombuffer(bytes(plan.ytab), np.uint8), np.frombuffer(bytes(plan.ctab), np.uint8)])
    cap = 4096 + sum(p.size for p in planes) * 3
    out = np.empty(cap, np.uint8)
    n = L.pw_write_progressive(ptrs, 3, plan.width, plan.height, hs, vs, qt.ctypes.data_as(C.c_void_p), int(script), out.ctypes.data_as(C.c_void_p), cap)
    assert 0 < n <= cap, n
    return out[:n].tobytes()


def progressive_grey_from_444(plan, du, script=1):
    """A single-component (grey) progressive stream carrying the luma data units of a 4:4:4 plan."""
    assert plan.du_per_mcu == 3
    L = C.CDLL(build_prog_writer())
    L.pw_write_progressive.restype = C.c_long
    L.pw_write_progressive.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, P_INT, P_INT, C.c_void_p, C.c_int, C.c_void_p, C.c_long]
    planes444, _ = du_to_planes(plan, du)
    y = np.ascontiguousarray(planes444[0])
    ptrs = (C.c_void_p * 1)(y.ctypes.data)
    one = (C.c_int * 1)(1)
    qt = np.concatenate([np.frombuffer(bytes(plan.ytab), np.uint8), np.frombuffer(bytes(plan.ctab), np.uint8)])
    cap = 4096 + y.size * 3
    out = np.empty(cap, np.uint8)
    n = L.pw_write_progressive(ptrs, 1, plan.width, plan.height, one, one, qt.ctypes.data_as(C.c_void_p), int(script), out.ctypes.data_as(C.c_void_p), cap)
    assert 0 < n <= cap, n
    return out[:n].tobytes()


def baseline_from_du(plan, du, restart_mcus=0, layout="native"):
    """A baseline (SOF0) stream from a writer plan's data units through the test-side writer, with optimal tables
    and, when restart_mcus > 0, a DRI segment and RSTn markers.  layout: "native" (the plan's own 4:2:0 or
    4:4:4), "422" or "grey" (both from a 4:4:4 plan, see progressive_422_from_444)."""
    L = C.CDLL(build_prog_writer())
    L.pw_write_baseline.restype = C.c_long
    L.pw_write_baseline.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, P_INT, P_INT, C.c_void_p, C.c_int, C.c_void_p, C.c_long]
    planes, samp = du_to_planes(plan, du)
    hs, vs = list(samp), list(samp)
    if layout == "422":
        assert plan.du_per_mcu == 3
        mcu_x, mcu_y = (plan.width + 15) // 16, (plan.height + 7) // 8
        y = np.zeros((mcu_y, 2 * mcu_x, 64), np.int16)
        y[:, :planes[0].shape[1]] = planes[0]
        new = [y]
        for c in (1, 2):
            sub = planes[c][:, 0::2]
            p = np.zeros((mcu_y, mcu_x, 64), np.int16)
            p[:, :sub.shape[1]] = sub
            new.append(p)
        planes, hs, vs = new, [2, 1, 1], [1, 1, 1]
    elif layout == "grey":
        assert plan.du_per_mcu == 3
        planes, hs, vs = [np.ascontiguousarray(planes[0])], [1], [1]
    n_c = len(planes)
    ptrs = (C.c_void_p * n_c)(*[p.ctypes.data for p in planes])
    qt = np.concatenate([np.frombuffer(bytes(plan.ytab), np.uint8), np.frombuffer(bytes(plan.ctab), np.uint8)])
    cap = 4096 + sum(p.size for p in planes) * 3
    out = np.empty(cap, np.uint8)
    n = L.pw_write_baseline(ptrs, n_c, plan.width, plan.height, (C.c_int * n_c)(*hs), (C.c_int * n_c)(*vs), qt.ctypes.data_as(C.c_void_p),
                            int(restart_mcus), out.ctypes.data_as(C.c_void_p), cap)
    assert 0 < n <= cap, n
    return out[:n].tobytes()


def baseline_layout_from_444(plan, du, hv, app14=-1, restart_mcus=0):
    """A baseline stream with arbitrary sampling factors (and optionally an Adobe APP14 segment / a fourth component)
    from the 4:4:4 data units of a picture of the same size: component c with factors hv[c] = (h, v) gets every
    (hmax/h)-th block column and (vmax/v)-th block row of the 4:4:4 plane (component 3: the luma plane again).  The
    content is not a faithful down-sampling -- only a valid stream of that layout is needed; parity is always against
    the oracle's decode of the same bytes."""
    assert plan.du_per_mcu == 3
    L = C.CDLL(build_prog_writer())
    L.pw_write_baseline_ex.restype = C.c_long
    L.pw_write_baseline_ex.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, P_INT, P_INT, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long]
    planes444, _ = du_to_planes(plan, du)
    n_c = len(hv)
    hmax, vmax = max(h for h, _ in hv), max(v for _, v in hv)
    mcu_x, mcu_y = (plan.width + 8 * hmax - 1) // (8 * hmax), (plan.height + 8 * vmax - 1) // (8 * vmax)
    planes = []
    for c, (h, v) in enumerate(hv):
        src = planes444[c if c < 3 else 0]
        sub = src[0::max(1, vmax // v), 0::max(1, hmax // h)]
        p = np.zeros((mcu_y * v, mcu_x * h, 64), np.int16)
        hh, ww = min(sub.shape[0], p.shape[0]), min(sub.shape[1], p.shape[1])
        p[:hh, :ww] = sub[:hh, :ww]
        planes.append(p)
    ptrs = (C.c_void_p * n_c)(*[p.ctypes.data for p in planes])
    qt = np.concatenate([np.frombuffer(bytes(plan.ytab), np.uint8), np.frombuffer(bytes(plan.ctab), np.uint8)])
    cap = 4096 + sum(p.size for p in planes) * 3
    out = np.empty(cap, np.uint8)
    n = L.pw_write_baseline_ex(ptrs, n_c, plan.width, plan.height, (C.c_int * n_c)(*[h for h, _ in hv]), (C.c_int * n_c)(*[v for _, v in hv]),
                               qt.ctypes.data_as(C.c_void_p), int(restart_mcus), int(app14), out.ctypes.data_as(C.c_void_p), cap)
    assert 0 < n <= cap, n
    return out[:n].tobytes()


# ---------------------------------------------------------------- round-2 golden additions

GOLDEN_R2 = os.path.join(ROOT, "tests", "golden", "jpeg_golden_r2.npz")


class GoldenR2:
    """tests/golden/jpeg_golden_r2.npz (make_golden_r2.py): late JFIF / Adobe markers, config 1, FILE* positions."""

    def __init__(self):
        self.z = np.load(GOLDEN_R2, allow_pickle=False)
        self.late_names = bytes(self.z["late_names"]).decode().split("\n")
        self.filepos_names = bytes(self.z["filepos_names"]).decode().split("\n")

    def late(self, name, req):
        data = bytes(self.z["late/%s/jpg" % name])
        k = "late/%s/out%d" % (name, req)
        if k in self.z:
            return data, "ok", self.z[k]
        return data, "fail", bytes(self.z["late/%s/fail%d" % (name, req)]).decode()

    def __getitem__(self, key):
        return self.z[key]


def fnv1a64(a):
    """FNV-1a 64 of a uint8 array (vectorised: the multiplications by the prime are folded per block of 1)"""
    h = 1469598103934665603
    for v in np.ascontiguousarray(a).reshape(-1).tolist():
        h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


# read() sizes of the k-th callback for the stbi_load_from_callbacks tests (and make_golden_r2.py, which records what
# the real reference does under each).  The reference rewinds to the FIRST buffer it read after its type test
# (stbi__rewind; common.c:10-26 refills in place), so a first read shorter than the two SOI bytes makes it fail with
# "no SOI" -- behaviour the product reproduces; patterns 3 and 4 pin that.
CB_PATTERNS = (
    lambda k: 128 if k == 0 else 1 + (k * 7) % 128,
    lambda k: 2,
    lambda k: 128 if k % 3 else 5,
    lambda k: 1,
    lambda k: 1 + (k * 7) % 128,
)


def third_tables(data):
    """The same baseline picture with a THIRD pair of Huffman tables: every DHT table with id 1 is repeated with id 2 and the scan's third
    component is pointed at DC / AC table 2 (the reference accepts table ids 0..3 in a baseline file, codec/jpeg.c:1016-1040, :1070-1085).
    Decodes to the same pixels; the GPU walk's entry tables hold two tables of each class, so the third takes its search path."""
    out, i, extra = bytearray(data[:2]), 2, bytearray()
    while i < len(data):
        assert data[i] == 0xFF
        m = data[i + 1]
        if m == 0xDA:
            n = (data[i + 2] << 8) | data[i + 3]
            seg = bytearray(data[i:i + 2 + n])
            ns = seg[4]
            assert ns == 3
            seg[5 + 2 * 2 + 1] = 0x22  # third component: Td = Ta = 2
            if extra:
                out += b"\xff\xc4" + bytes([(len(extra) + 2) >> 8, (len(extra) + 2) & 255]) + extra
            out += seg + data[i + 2 + n:]
            return bytes(out)
        n = (data[i + 2] << 8) | data[i + 3]
        seg = data[i:i + 2 + n]
        if m == 0xC4:
            j = 4
            while j < len(seg):
                tc_th = seg[j]
                cnt = sum(seg[j + 1:j + 17])
                if (tc_th & 15) == 1:
                    extra += bytes([(tc_th & 0xF0) | 2]) + seg[j + 1:j + 17 + cnt]
                j += 17 + cnt
        out += seg
        i += 2 + n
    raise AssertionError("no SOS")
