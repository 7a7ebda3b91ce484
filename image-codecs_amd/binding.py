"""ctypes bindings over lib/libimagecodecs_mi355x.so (see package docstring).

Nothing here computes pixels: every function forwards to the C-ABI.  The library is loaded
lazily; a missing library raises (no fallback).  Decode calls need a gfx950 GPU and fail with
the library's own reason ("no gpu device") without one.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# eight hardware queues instead of HIP's default four, unless the user chose (see mij_hip_defaults in mij_runtime.hip): must be in
# the environment before the HIP runtime initialises, i.e. before anything in this process touches the GPU
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
LIB_PATH = os.environ.get("MIJ_LIB") or os.path.join(_HERE, "lib", "libimagecodecs_mi355x.so")  # MIJ_LIB: A/B builds

MIJ_FLAG_WIDE_IDCT = 1
COLOR_NAMES = {0: "grey", 1: "ycbcr", 2: "rgb", 3: "cmyk", 4: "ycck", 5: "ycbcra"}

ROWSLOT = (0, 4, 2, 5, 1, 6, 3, 7)  # include/mij.h mij_rowslot


class MijError(RuntimeError):
    pass


class CompDesc(C.Structure):
    _fields_ = [("h", C.c_int32), ("v", C.c_int32), ("tq", C.c_int32), ("x", C.c_int32), ("y", C.c_int32),
                ("bw", C.c_int32), ("bh", C.c_int32)]


class ImageDesc(C.Structure):
    """mij_image_desc (include/mij.h)."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("ncomp", C.c_int32), ("n_out", C.c_int32),
                ("color", C.c_int32), ("flags", C.c_uint32), ("h_max", C.c_int32), ("v_max", C.c_int32),
                ("mcu_x", C.c_int32), ("mcu_y", C.c_int32), ("comp", CompDesc * 4),
                ("dequant", (C.c_uint16 * 64) * 4)]

    def plane_elems(self, c):
        nblk = self.comp[c].bw * self.comp[c].bh
        return ((nblk + 63) >> 6) << 12

    def coef_elems(self):
        return sum(self.plane_elems(c) for c in range(self.ncomp))


class GpuHuff(C.Structure):
    """mjg_huff (include/mij.h)."""
    _fields_ = [("fast", C.c_uint8 * 512), ("size", C.c_uint8 * 256), ("values", C.c_uint8 * 256), ("maxcode", C.c_uint32 * 18),
                ("delta", C.c_int32 * 18)]


class GpuScan(C.Structure):
    """mjg_scan (include/mij.h): what the GPU entropy stage needs to walk one baseline scan."""
    _fields_ = [("desc", ImageDesc), ("nblocks", C.c_uint32), ("blocks_per_mcu", C.c_uint32), ("blk_comp", C.c_uint8 * 12),
                ("blk_dx", C.c_uint8 * 12), ("blk_dy", C.c_uint8 * 12), ("dc_tab", C.c_uint8 * 4), ("ac_tab", C.c_uint8 * 4),
                ("huff", GpuHuff * 8), ("qz", (C.c_uint16 * 64) * 4), ("n_seg", C.c_uint32), ("restart_mcus", C.c_uint32),
                ("seg_table_off", C.c_uint32), ("reserved", C.c_uint32)]


_lib = None


def _library_stale():
    """True when the .so is missing or older than any source it is built from (csrc/, include/)."""
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    for d in (os.path.join(_HERE, "csrc"), os.path.join(os.path.dirname(_HERE), "include")):
        for name in os.listdir(d):
            if name.endswith((".c", ".h", ".hip")) or name == "Makefile":
                if os.path.getmtime(os.path.join(d, name)) > built:
                    return True
    return False


def build_library(force=False):
    """Compile the HIP/C sources in csrc/ for gfx950 (hipcc cross-compiles without a GPU).  Rebuilds when the
    library is missing or any file under csrc/ or include/ is newer than it (make decides what to recompile)."""
    if os.environ.get("MIJ_LIB"):
        return LIB_PATH  # an A/B build chosen by the caller: never rebuilt behind its back
    if force or _library_stale():
        subprocess.run(["make", "-C", os.path.join(_HERE, "csrc")] + (["-B"] if force else []), check=True)
    return LIB_PATH


def lib():
    """The loaded shared library (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MijError("%s is missing: run `python __graft_entry__.py` (build()) or `make -C image-codecs_amd/csrc`" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    p_int = C.POINTER(C.c_int)
    L.stbi_load_from_memory.restype = C.POINTER(C.c_ubyte)
    L.stbi_load_from_memory.argtypes = [C.c_char_p, C.c_int, p_int, p_int, p_int, C.c_int]
    L.stbi_load.restype = C.POINTER(C.c_ubyte)
    L.stbi_load.argtypes = [C.c_char_p, p_int, p_int, p_int, C.c_int]
    L.stbi_load_16_from_memory.restype = C.POINTER(C.c_ushort)
    L.stbi_load_16_from_memory.argtypes = [C.c_char_p, C.c_int, p_int, p_int, p_int, C.c_int]
    L.stbi_info_from_memory.restype = C.c_int
    L.stbi_info_from_memory.argtypes = [C.c_char_p, C.c_int, p_int, p_int, p_int]
    L.stbi_failure_reason.restype = C.c_char_p
    L.stbi_image_free.argtypes = [C.c_void_p]
    L.stbi_set_flip_vertically_on_load.argtypes = [C.c_int]
    L.stbi_write_jpg_to_func.restype = C.c_int
    L.mij_last_error.restype = C.c_char_p
    L.mij_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.mij_ctx_destroy.argtypes = [C.c_void_p]
    L.mij_ctx_info.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, p_int, C.POINTER(C.c_size_t)]
    L.mij_batch_create.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]
    L.mij_batch_destroy.argtypes = [C.c_void_p]
    L.mij_batch_reset.argtypes = [C.c_void_p]
    L.mij_image_coef_bytes.restype = C.c_size_t
    L.mij_image_coef_bytes.argtypes = [C.POINTER(ImageDesc)]
    L.mij_image_out_bytes.restype = C.c_size_t
    L.mij_image_out_bytes.argtypes = [C.POINTER(ImageDesc)]
    L.mij_batch_add.argtypes = [C.c_void_p, C.POINTER(ImageDesc)]
    L.mij_batch_add_clone.argtypes = [C.c_void_p, C.c_int]
    L.mij_batch_coef.restype = C.c_void_p
    L.mij_batch_coef.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.mij_batch_set_flags.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
    L.mij_batch_set_color.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.mij_batch_set_coef_format.argtypes = [C.c_void_p, C.c_int]
    L.mij_batch_slot_escapes.argtypes = [C.c_void_p, C.c_int]
    L.mij_batch_slot_coef_bytes.argtypes = [C.c_void_p, C.c_int]
    for name in ("mij_batch_upload", "mij_batch_launch", "mij_batch_submit", "mij_batch_wait", "mij_batch_timer_begin",
                 "mij_batch_timer_end", "mij_batch_image_count"):
        getattr(L, name).argtypes = [C.c_void_p]
    L.mij_batch_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.mij_batch_device_out.restype = C.c_void_p
    L.mij_batch_device_out.argtypes = [C.c_void_p, C.c_int]
    L.mij_batch_timer_elapsed_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.mij_batch_hash_out.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    L.mij_batch_slot_path.argtypes = [C.c_void_p, C.c_int]
    L.mij_batch_force_generic.argtypes = [C.c_void_p, C.c_int]
    L.mjh_probe_memory.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(ImageDesc), C.POINTER(C.c_char_p)]
    L.mjh_decode_memory.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(ImageDesc), C.c_void_p, C.c_size_t,
                                    C.POINTER(C.c_char_p)]
    L.mjh_decode_batch.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int,
                                   C.POINTER(C.c_int), C.POINTER(C.c_char_p)]
    _lib = L
    return L


def gpu_available():
    """True when the library sees at least one HIP device (no torch involved)."""
    try:
        return lib().mij_device_count() > 0
    except (MijError, OSError):
        return False


# ---------------------------------------------------------------- stb-style surface

def stbi_failure_reason():
    r = lib().stbi_failure_reason()
    return r.decode() if r else None


def stbi_set_flip_vertically_on_load(flag):
    lib().stbi_set_flip_vertically_on_load(int(flag))


def _take(ptr, shape, dtype):
    n = int(np.prod(shape))
    arr = np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).reshape(shape).copy()
    lib().stbi_image_free(ptr)
    return arr


def stbi_load_from_memory(data, req_comp=0):
    """-> (pixels[h, w, n] uint8, w, h, comp_in_file) or None (see stbi_failure_reason())."""
    x, y, c = C.c_int(), C.c_int(), C.c_int()
    p = lib().stbi_load_from_memory(bytes(data), len(data), C.byref(x), C.byref(y), C.byref(c), int(req_comp))
    if not p:
        return None
    n = req_comp if req_comp else c.value
    return _take(p, (y.value, x.value, n), np.uint8), x.value, y.value, c.value


def stbi_load(filename, req_comp=0):
    x, y, c = C.c_int(), C.c_int(), C.c_int()
    p = lib().stbi_load(os.fsencode(filename), C.byref(x), C.byref(y), C.byref(c), int(req_comp))
    if not p:
        return None
    n = req_comp if req_comp else c.value
    return _take(p, (y.value, x.value, n), np.uint8), x.value, y.value, c.value


def stbi_load_16_from_memory(data, req_comp=0):
    x, y, c = C.c_int(), C.c_int(), C.c_int()
    p = lib().stbi_load_16_from_memory(bytes(data), len(data), C.byref(x), C.byref(y), C.byref(c), int(req_comp))
    if not p:
        return None
    n = req_comp if req_comp else c.value
    cnt = y.value * x.value * n
    arr = np.ctypeslib.as_array(p, shape=(cnt,)).astype(np.uint16).reshape(y.value, x.value, n).copy()
    lib().stbi_image_free(p)
    return arr, x.value, y.value, c.value


def stbi_info_from_memory(data):
    """-> (ok, w, h, comp)"""
    x, y, c = C.c_int(), C.c_int(), C.c_int()
    ok = lib().stbi_info_from_memory(bytes(data), len(data), C.byref(x), C.byref(y), C.byref(c))
    return ok, x.value, y.value, c.value


# ---- the callback / FILE* / file-name variants of the loaders (convert.c:188-211,261-266; image_api.c:74-131)

_READ_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_char), C.c_int)
_SKIP_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int)
_EOF_CB = C.CFUNCTYPE(C.c_int, C.c_void_p)


class IoCallbacks(C.Structure):
    """stbi_io_callbacks (include/image_api.h)."""
    _fields_ = [("read", _READ_CB), ("skip", _SKIP_CB), ("eof", _EOF_CB)]


class _CallbackSource:
    """A byte string behind stbi_io_callbacks whose read() hands out at most `chunk(k)` bytes on its k-th call
    (short reads are legal: the reference refills its 128-byte buffer with whatever comes, common.c:10-26)."""

    def __init__(self, data, chunk=None):
        self.data, self.pos, self.calls = bytes(data), 0, 0
        self.chunk = chunk

        def read(_u, buf, size):
            n = min(size, len(self.data) - self.pos)
            if self.chunk is not None:
                n = min(n, max(1, int(self.chunk(self.calls))))
            self.calls += 1
            C.memmove(buf, self.data[self.pos:self.pos + n], n)
            self.pos += n
            return n

        def skip(_u, n):
            self.pos = min(len(self.data), max(0, self.pos + n))

        def eof(_u):
            return 1 if self.pos >= len(self.data) else 0

        self.cb = IoCallbacks(_READ_CB(read), _SKIP_CB(skip), _EOF_CB(eof))


def stbi_load_from_callbacks(data, req_comp=0, chunk=None):
    """stbi_load_from_callbacks over an in-memory source with (optionally) short reads; result as stbi_load_from_memory."""
    L = lib()
    L.stbi_load_from_callbacks.restype = C.POINTER(C.c_ubyte)
    L.stbi_load_from_callbacks.argtypes = [C.POINTER(IoCallbacks), C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    src = _CallbackSource(data, chunk)
    x, y, c = C.c_int(), C.c_int(), C.c_int()
    p = L.stbi_load_from_callbacks(C.byref(src.cb), None, C.byref(x), C.byref(y), C.byref(c), int(req_comp))
    if not p:
        return None
    n = req_comp if req_comp else c.value
    return _take(p, (y.value, x.value, n), np.uint8), x.value, y.value, c.value


def stbi_info_from_callbacks(data, chunk=None):
    L = lib()
    L.stbi_info_from_callbacks.argtypes = [C.POINTER(IoCallbacks), C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    src = _CallbackSource(data, chunk)
    x, y, c = C.c_int(), C.c_int(), C.c_int()
    ok = L.stbi_info_from_callbacks(C.byref(src.cb), None, C.byref(x), C.byref(y), C.byref(c))
    return ok, x.value, y.value, c.value


_libc = None


def _c_stdio():
    global _libc
    if _libc is None:
        _libc = C.CDLL(None)
        _libc.fopen.restype = C.c_void_p
        _libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
        _libc.fclose.argtypes = [C.c_void_p]
        _libc.ftell.restype = C.c_long
        _libc.ftell.argtypes = [C.c_void_p]
        _libc.fseek.argtypes = [C.c_void_p, C.c_long, C.c_int]
    return _libc


def stbi_load_from_file(filename, req_comp=0, offset=0):
    """fopen + fseek(offset) + stbi_load_from_file + ftell: -> (result as stbi_load_from_memory or None, position the
    FILE* was left at) -- the reference seeks back over what it buffered but did not consume (convert.c:208)."""
    L, libc = lib(), _c_stdio()
    L.stbi_load_from_file.restype = C.POINTER(C.c_ubyte)
    L.stbi_load_from_file.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    f = libc.fopen(os.fsencode(filename), b"rb")
    if not f:
        raise OSError("fopen failed: %s" % filename)
    try:
        libc.fseek(f, offset, 0)
        x, y, c = C.c_int(), C.c_int(), C.c_int()
        p = L.stbi_load_from_file(f, C.byref(x), C.byref(y), C.byref(c), int(req_comp))
        pos = libc.ftell(f)
        if not p:
            return None, pos
        n = req_comp if req_comp else c.value
        return (_take(p, (y.value, x.value, n), np.uint8), x.value, y.value, c.value), pos
    finally:
        libc.fclose(f)


def stbi_info_from_file(filename, offset=0):
    """-> ((ok, w, h, comp), position afterwards): stbi_info_from_file restores the position (image_api.c:85-94)."""
    L, libc = lib(), _c_stdio()
    L.stbi_info_from_file.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    f = libc.fopen(os.fsencode(filename), b"rb")
    if not f:
        raise OSError("fopen failed: %s" % filename)
    try:
        libc.fseek(f, offset, 0)
        x, y, c = C.c_int(), C.c_int(), C.c_int()
        ok = L.stbi_info_from_file(f, C.byref(x), C.byref(y), C.byref(c))
        return (ok, x.value, y.value, c.value), libc.ftell(f)
    finally:
        libc.fclose(f)


def stbi_info(filename):
    L = lib()
    L.stbi_info.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    x, y, c = C.c_int(), C.c_int(), C.c_int()
    ok = L.stbi_info(os.fsencode(filename), C.byref(x), C.byref(y), C.byref(c))
    return ok, x.value, y.value, c.value


def stbi_write_jpg(filename, pixels, quality=90):
    """stbi_write_jpg(filename, ...) (codec/jpeg_write.c:376-388); returns its int result."""
    a = np.ascontiguousarray(pixels, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, comp = a.shape
    L = lib()
    L.stbi_write_jpg.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    return L.stbi_write_jpg(os.fsencode(filename), w, h, comp, a.ctypes.data_as(C.c_void_p), int(quality))


_WRITE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int)


def stbi_write_jpg_to_memory(pixels, quality=90):
    """stbi_write_jpg_to_func into a bytes object.  pixels: uint8 [h, w, comp] (comp 1..4) or [h, w]."""
    a = np.ascontiguousarray(pixels, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, comp = a.shape
    chunks = []

    def sink(_ctx, data, size):
        chunks.append(C.string_at(data, size))

    cb = _WRITE_CB(sink)
    ok = lib().stbi_write_jpg_to_func(cb, None, C.c_int(w), C.c_int(h), C.c_int(comp), a.ctypes.data_as(C.c_void_p), C.c_int(int(quality)))
    if not ok:
        return None
    return b"".join(chunks)


# ---------------------------------------------------------------- host entropy stage

class HostDecoder:
    """mjh_probe_memory / mjh_decode_memory: marker parse + Huffman walk into tile-layout planes."""

    @staticmethod
    def probe(data, req_comp=0):
        d = ImageDesc()
        why = C.c_char_p()
        ok = lib().mjh_probe_memory(bytes(data), len(data), int(req_comp), C.byref(d), C.byref(why))
        if not ok:
            raise MijError(why.value.decode() if why.value else "decode failed")
        return d

    @staticmethod
    def decode(data, req_comp=0, out=None):
        """-> (desc, arena int16[coef_elems]) ; `out` may be a preallocated int16 array (e.g. a view of pinned staging)."""
        d = HostDecoder.probe(data, req_comp)
        n = d.coef_elems()
        arena = out if out is not None else np.empty(n, dtype=np.int16)
        if arena.size < n:
            raise MijError("arena too small")
        why = C.c_char_p()
        ok = lib().mjh_decode_memory(bytes(data), len(data), int(req_comp), C.byref(d), arena.ctypes.data_as(C.c_void_p),
                                     C.c_size_t(arena.size), C.byref(why))
        if not ok:
            raise MijError(why.value.decode() if why.value else "decode failed")
        return d, arena


def host_decode_staged(data, req_comp=0, want_compact=True):
    """mjh_decode_memory_fmt into a plain numpy region (no GPU): -> (desc, region bytes).  desc.flags says which format the walk wrote
    (MIJ_FLAG_STAGED_COMPACT = 4, MIJ_FLAG_HAS_ESCAPES = 8); compact regions follow mij_compact_offsets (compact_offsets below)."""
    d = HostDecoder.probe(data, req_comp)
    L = lib()
    L.mjh_decode_memory_fmt.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(ImageDesc), C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_char_p)]
    tiles = [((d.comp[c].bw * d.comp[c].bh) + 63) // 64 for c in range(d.ncomp)]
    region = np.full(sum(tiles) * (4096 + 128 + 4096), 0xA5, dtype=np.uint8)  # poisoned: what the walk does not clear it must not need
    why = C.c_char_p()
    d2 = ImageDesc()
    if not L.mjh_decode_memory_fmt(bytes(data), len(data), int(req_comp), C.byref(d2), region.ctypes.data_as(C.c_void_p), C.c_size_t(region.size), int(bool(want_compact)), C.byref(why)):
        raise MijError(why.value.decode() if why.value else "decode failed")
    return d2, region


def compact_offsets(desc):
    """mij_compact_offsets: per component (lo, dc, hi) byte offsets inside an image's region, and the main part's size"""
    tiles = [((desc.comp[c].bw * desc.comp[c].bh) + 63) // 64 for c in range(desc.ncomp)]
    main = sum(t * (4096 + 128) for t in tiles)
    out, m, e = [], 0, main
    for t in tiles:
        out.append((m, m + t * 4096, e))
        m += t * (4096 + 128)
        e += t * 4096
    return out, main


def expand_compact_region(desc, region):
    """compact planes (mij.h) -> the int16 tile-layout arena mjh_decode_memory would have written (the inverse of k_pack_c8)"""
    offs, _ = compact_offsets(desc)
    parts = []
    for c, (lo_o, dc_o, hi_o) in enumerate(offs):
        nt = ((desc.comp[c].bw * desc.comp[c].bh) + 63) // 64
        lo = region[lo_o:lo_o + nt * 4096].reshape(nt, 8, 64, 8)          # [tile, chunk, lane, slot]
        dc = region[dc_o:dc_o + nt * 128].view(np.int16).reshape(nt, 64)
        hi = region[hi_o:hi_o + nt * 4096].reshape(nt, 64, 8, 8)          # [tile, lane, chunk, slot] (64 bytes per block, position order)
        val = lo.view(np.int8).astype(np.int32)
        esc = (lo[:, 0, :, 0] & 1).astype(bool)                            # [tile, lane]
        h = np.where(esc[:, None, :, None], hi.view(np.int8).transpose(0, 2, 1, 3).astype(np.int32), 0)
        val = val + 256 * h
        val[:, 0, :, 0] = dc
        parts.append(val.astype(np.int16).reshape(-1))
    return np.concatenate(parts)


def detile_coefficients(desc, arena):
    """Tile layout -> list of per-component arrays [bh, bw, 8, 8] (natural row, col order), for tests."""
    out = []
    off = 0
    pos = np.empty((8, 8), dtype=np.int64)  # [row, col] -> P
    for r in range(8):
        for c in range(8):
            pos[r, c] = 8 * c + ROWSLOT[r]
    for ci in range(desc.ncomp):
        bw, bh = desc.comp[ci].bw, desc.comp[ci].bh
        n = desc.plane_elems(ci)
        plane = np.asarray(arena[off:off + n]).reshape(-1, 8, 64, 8)  # [tile, chunk, lane, j]
        off += n
        L = np.arange(bw * bh)
        blk = plane[L >> 6, :, L & 63, :]  # [nblk, chunk, j]
        flat = blk.reshape(-1, 64)  # index P = 8*chunk + j
        nat = flat[:, pos.reshape(-1)].reshape(bh, bw, 8, 8)
        out.append(nat)
    return out


# ---------------------------------------------------------------- GPU back end

def _check(rc, what):
    if rc < 0:
        raise MijError("%s: %s" % (what, lib().mij_last_error().decode()))
    return rc


class Context:
    """mij_ctx: one per (process, device)."""

    def __init__(self, device=-1):
        self._h = C.c_void_p()
        _check(lib().mij_ctx_create(int(device), C.byref(self._h)), "mij_ctx_create")

    def info(self):
        arch = C.create_string_buffer(64)
        cu = C.c_int()
        mem = C.c_size_t()
        _check(lib().mij_ctx_info(self._h, arch, 64, C.byref(cu), C.byref(mem)), "mij_ctx_info")
        return arch.value.decode(), cu.value, mem.value

    def close(self):
        if self._h:
            lib().mij_ctx_destroy(self._h)
            self._h = C.c_void_p()


class Batch:
    """mij_batch: staging + device arenas + stream.  Thin, order-preserving wrapper."""

    def __init__(self, ctx, max_images, stage_bytes, coef_bytes, out_bytes):
        self.ctx = ctx
        self._h = C.c_void_p()
        _check(lib().mij_batch_create(ctx._h, int(max_images), C.c_size_t(stage_bytes), C.c_size_t(coef_bytes), C.c_size_t(out_bytes),
                                      C.byref(self._h)), "mij_batch_create")
        self.descs = []

    @staticmethod
    def coef_bytes(desc):
        return lib().mij_image_coef_bytes(C.byref(desc))

    @staticmethod
    def out_bytes(desc):
        return lib().mij_image_out_bytes(C.byref(desc))

    def add(self, desc):
        slot = _check(lib().mij_batch_add(self._h, C.byref(desc)), "mij_batch_add")
        self.descs.append(desc)
        return slot

    def _desc(self, slot):
        """Descriptor of a slot; slots added by the C-side front ends keep (data, req_comp) until first use, so that
        a pipelined caller does not pay a Python-side header probe per image."""
        d = self.descs[slot]
        if isinstance(d, tuple):
            d = self.descs[slot] = HostDecoder.probe(d[0], d[1])
        return d

    def add_clone(self, src):
        slot = _check(lib().mij_batch_add_clone(self._h, int(src)), "mij_batch_add_clone")
        self.descs.append(self.descs[src])
        return slot

    def stage_region(self, slot):
        """uint8 numpy view over the slot's whole pinned staging region (mij_batch_stage_region: room for either format)"""
        L = lib()
        L.mij_batch_stage_region.restype = C.c_void_p
        L.mij_batch_stage_region.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
        nb = C.c_size_t()
        p = L.mij_batch_stage_region(self._h, int(slot), C.byref(nb))
        if not p:
            raise MijError("mij_batch_stage_region: %s" % L.mij_last_error().decode())
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nb.value,))

    def set_flags(self, slot, flags):
        _check(lib().mij_batch_set_flags(self._h, int(slot), int(flags)), "mij_batch_set_flags")
        d = self._desc(slot)
        d.flags = int(flags)

    def staging(self, slot):
        """int16 numpy view over the slot's pinned planes (all components, back to back)."""
        d = self._desc(slot)
        p = lib().mij_batch_coef(self._h, int(slot), 0)
        if not p:
            raise MijError("mij_batch_coef: %s" % lib().mij_last_error().decode())
        n = d.coef_elems()
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int16)), shape=(n,))

    # ---- GPU entropy stage (experimental)
    def entropy_reserve(self, stream_bytes):
        L = lib()
        L.mij_batch_entropy_reserve.argtypes = [C.c_void_p, C.c_size_t]
        _check(L.mij_batch_entropy_reserve(self._h, C.c_size_t(stream_bytes)), "mij_batch_entropy_reserve")
        L.mij_batch_entropy_stage.restype = C.c_void_p
        L.mij_batch_entropy_stage.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        cap = C.c_size_t()
        self._es_base = L.mij_batch_entropy_stage(self._h, C.byref(cap))
        self._es_cap, self._es_used = cap.value, 0

    def add_jpeg_stream(self, data, req_comp=0):
        """mjh_extract_scan + mij_batch_add_stream: -> (status, slot).  status 1: the GPU walks this image;
        2: not a layout the GPU walk takes (nothing added); 0: rejected (reason in .last_reason)."""
        L = lib()
        L.mjh_extract_scan.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(GpuScan), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_char_p)]
        L.mij_batch_add_stream.argtypes = [C.c_void_p, C.POINTER(GpuScan), C.c_void_p, C.c_size_t]
        scan, n, why = GpuScan(), C.c_size_t(), C.c_char_p()
        dst = self._es_base + self._es_used
        st = L.mjh_extract_scan(bytes(data), len(data), int(req_comp), C.byref(scan), C.c_void_p(dst), C.c_size_t(self._es_cap - self._es_used), C.byref(n), C.byref(why))
        self.last_reason = why.value.decode() if why.value else None
        if st != 1:
            return st, -1
        slot = _check(L.mij_batch_add_stream(self._h, C.byref(scan), C.c_void_p(dst), n), "mij_batch_add_stream")
        self._es_used += (n.value + 32 + 255) // 256 * 256
        d = ImageDesc()
        C.memmove(C.byref(d), C.byref(scan.desc), C.sizeof(ImageDesc))
        self.descs.append(d)
        return 1, slot

    def entropy_run(self):
        """-> list of slots the host walk must redo."""
        L = lib()
        n = max(1, len(self.descs))
        fb = (C.c_int * n)()
        cnt = C.c_int()
        L.mij_batch_entropy_run.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
        _check(L.mij_batch_entropy_run(self._h, fb, n, C.byref(cnt)), "mij_batch_entropy_run")
        return list(fb[:cnt.value])

    def slot_coef_bytes(self, slot):
        """1 when the slot's coefficients sit in HBM as compact planes (the default), 0 for the int16 tile layout."""
        return lib().mij_batch_slot_coef_bytes(self._h, int(slot))

    def set_coef_format(self, fmt):
        """'compact' (default) or 'int16': the format new coefficient planes of this batch get in HBM."""
        _check(lib().mij_batch_set_coef_format(self._h, {"int16": 0, "compact": 1}[fmt]), "mij_batch_set_coef_format")

    def slot_escapes(self, slot):
        """Blocks of the slot that hold a coefficient outside -128..127 (compact planes only)."""
        return _check(lib().mij_batch_slot_escapes(self._h, int(slot)), "mij_batch_slot_escapes")

    def slot_flags(self, slot):
        L = lib()
        L.mij_batch_slot_flags.restype = C.c_uint32
        L.mij_batch_slot_flags.argtypes = [C.c_void_p, C.c_int]
        return int(L.mij_batch_slot_flags(self._h, int(slot)))

    def pack_ms(self):
        """ms of k_pack_c8 in the last upload, or None when nothing was packed (host-staged compact planes, GPU-walk planes, int16 planes)"""
        L = lib()
        L.mij_batch_pack_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        ms = C.c_float()
        _check(L.mij_batch_pack_ms(self._h, C.byref(ms)), "mij_batch_pack_ms")
        return None if ms.value < 0 else ms.value

    def count_idct_classes(self, on=True):
        """Measurement: make the launches that follow count wavefronts per sparse-block class (clears the counters when switched on)."""
        L = lib()
        L.mij_batch_count_idct_classes.argtypes = [C.c_void_p, C.c_int]
        _check(L.mij_batch_count_idct_classes(self._h, int(bool(on))), "mij_batch_count_idct_classes")

    def idct_class_counts(self):
        """[DC only, inside 2x2, inside 4x4, full] wavefronts since count_idct_classes(True)"""
        L = lib()
        out = (C.c_uint64 * 4)()
        L.mij_batch_idct_class_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        _check(L.mij_batch_idct_class_counts(self._h, out), "mij_batch_idct_class_counts")
        return [int(v) for v in out]

    def entropy_rounds(self):
        L = lib()
        L.mij_batch_entropy_rounds.argtypes = [C.c_void_p]
        return L.mij_batch_entropy_rounds(self._h)

    def fallback_prepare(self, slot):
        _check(lib().mij_batch_fallback_prepare(self._h, int(slot)), "mij_batch_fallback_prepare")

    def fetch_coef(self, slot):
        d = self._desc(slot)
        out = np.empty(d.coef_elems(), dtype=np.int16)
        L = lib()
        L.mij_batch_fetch_coef.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        _check(L.mij_batch_fetch_coef(self._h, int(slot), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size)), "mij_batch_fetch_coef")
        return out

    def add_jpeg(self, data, req_comp=0, stage=None):
        """Host stage of one image straight into a new slot's pinned staging; returns the slot.  stage None: what the batch front ends do
        (mjh_decode_memory_fmt: a baseline file is staged as COMPACT planes by the walk itself when the batch's format is compact, no
        pack pass on the device); "int16": int16 tile-layout staging whatever the file (mjh_decode_memory; compact batches then pack
        it on the device with k_pack_c8 -- the pipeline of rounds 1 and 2, kept for progressive files)."""
        d = HostDecoder.probe(data, req_comp)
        slot = self.add(d)
        L = lib()
        if stage == "int16":
            d2, _ = HostDecoder.decode(data, req_comp, out=self.staging(slot))
        else:
            L.mij_batch_stage_region.restype = C.c_void_p
            L.mij_batch_stage_region.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
            L.mij_batch_coef_format.argtypes = [C.c_void_p]
            L.mjh_decode_memory_fmt.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(ImageDesc), C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_char_p)]
            nb = C.c_size_t()
            region = L.mij_batch_stage_region(self._h, int(slot), C.byref(nb))
            if not region:
                raise MijError("mij_batch_stage_region: " + L.mij_last_error().decode())
            d2, why = ImageDesc(), C.c_char_p()
            if not L.mjh_decode_memory_fmt(bytes(data), len(data), int(req_comp), C.byref(d2), region, nb, int(L.mij_batch_coef_format(self._h) == 1), C.byref(why)):
                raise MijError("mjh_decode_memory_fmt: %s" % (why.value.decode() if why.value else "failed"))
        if d2.flags:
            _check(L.mij_batch_set_flags(self._h, slot, d2.flags), "mij_batch_set_flags")
            d.flags = d2.flags
        if d2.color != d.color:  # a JFIF / Adobe marker behind SOF changed the colour branch (codec/jpeg.c:2244)
            _check(L.mij_batch_set_color(self._h, slot, d2.color), "mij_batch_set_color")
            d.color = d2.color
        return slot

    def decode_jpegs(self, datas, req_comp=0, threads=1, gpu_entropy=None):
        """gpu_entropy None: mjh_decode_batch, the default front end (Huffman walk on the GPU where it applies, host walk
        as the fallback; the batch gets its entropy arena on first use); False: mjh_decode_batch_host (every walk on the
        host threads); True: mjh_decode_batch_gpu (needs entropy_reserve).
        -> (n_ok, slots, reasons); slots[i] < 0 marks a rejected image (reasons[i] says why)."""
        n = len(datas)
        bufs = (C.c_char_p * n)(*[bytes(d) for d in datas])
        lens = (C.c_int * n)(*[len(d) for d in datas])
        slots = (C.c_int * n)()
        reasons = (C.c_char_p * n)()
        first = len(self.descs)
        fn = lib().mjh_decode_batch if gpu_entropy is None else (lib().mjh_decode_batch_gpu if gpu_entropy else lib().mjh_decode_batch_host)
        fn.argtypes = lib().mjh_decode_batch.argtypes
        rc = fn(self._h, bufs, lens, n, int(req_comp), int(threads), slots, reasons)
        if rc < 0:
            raise MijError("mjh_decode_batch: %s" % lib().mij_last_error().decode())
        out_slots = list(slots)
        # mirror the descriptors of the slots the C side added
        for i, sl in enumerate(out_slots):
            real = sl if sl >= 0 else (-1 - sl if sl < -1 else None)
            if real is not None and real >= first:
                self.descs.append((datas[i], req_comp))  # header probed when somebody asks (fetch, staging)
        return rc, out_slots, [r.decode() if r else None for r in reasons]

    def decode_jpegs_gpu_begin(self, datas, req_comp=0, threads=1):
        """mjh_decode_batch_gpu_begin: headers + unstuffing on the host threads, GPU walk queued; returns a job
        to pass to decode_jpegs_gpu_end (which waits for the walk and host-walks what it handed back)."""
        L = lib()
        n = len(datas)
        job = {"datas": datas, "req": req_comp, "first": len(self.descs),
               "bufs": (C.c_char_p * n)(*[bytes(d) for d in datas]), "lens": (C.c_int * n)(*[len(d) for d in datas]),
               "slots": (C.c_int * n)(), "reasons": (C.c_char_p * n)(), "rc": C.c_int()}
        L.mjh_decode_batch_gpu_begin.restype = C.c_void_p
        L.mjh_decode_batch_gpu_begin.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                                 C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
        job["h"] = L.mjh_decode_batch_gpu_begin(self._h, job["bufs"], job["lens"], n, int(req_comp), int(threads), job["slots"], job["reasons"], C.byref(job["rc"]))
        if not job["h"]:
            raise MijError("mjh_decode_batch_gpu_begin: %d %s" % (job["rc"].value, lib().mij_last_error().decode()))
        return job

    def decode_jpegs_gpu_end(self, job):
        L = lib()
        L.mjh_decode_batch_gpu_end.argtypes = [C.c_void_p]
        rc = L.mjh_decode_batch_gpu_end(C.c_void_p(job["h"]))
        if rc < 0:
            raise MijError("mjh_decode_batch_gpu_end: %s" % lib().mij_last_error().decode())
        out_slots = list(job["slots"])
        for i, sl in enumerate(out_slots):
            real = sl if sl >= 0 else (-1 - sl if sl < -1 else None)
            if real is not None and real >= job["first"]:
                self.descs.append((job["datas"][i], job["req"]))
        return rc, out_slots, [r.decode() if r else None for r in job["reasons"]]

    def set_flags(self, slot, flags):
        _check(lib().mij_batch_set_flags(self._h, int(slot), int(flags)), "mij_batch_set_flags")

    def force_generic(self, on=True):
        """True / 1: two-pass path for every image; 2: and the run-time-general pass 2; False / 0: default choice."""
        _check(lib().mij_batch_force_generic(self._h, int(on)), "mij_batch_force_generic")

    def upload(self):
        _check(lib().mij_batch_upload(self._h), "mij_batch_upload")

    def launch(self):
        _check(lib().mij_batch_launch(self._h), "mij_batch_launch")

    def submit(self):
        _check(lib().mij_batch_submit(self._h), "mij_batch_submit")

    def wait(self):
        _check(lib().mij_batch_wait(self._h), "mij_batch_wait")

    def fetch(self, slot):
        d = self._desc(slot)
        out = np.empty((d.height, d.width, d.n_out), dtype=np.uint8)
        _check(lib().mij_batch_fetch(self._h, int(slot), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size)), "mij_batch_fetch")
        return out

    def fetch_all_async(self, dst_ptr, dst_bytes):
        """mij_batch_fetch_all_async into a (pinned) host buffer; wait() completes it."""
        L = lib()
        L.mij_batch_fetch_all_async.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        _check(L.mij_batch_fetch_all_async(self._h, C.c_void_p(dst_ptr), C.c_size_t(dst_bytes)), "mij_batch_fetch_all_async")

    def out_offset(self, slot):
        L = lib()
        L.mij_batch_out_offset.restype = C.c_size_t
        L.mij_batch_out_offset.argtypes = [C.c_void_p, C.c_int]
        return L.mij_batch_out_offset(self._h, int(slot))

    def out_total_bytes(self):
        L = lib()
        L.mij_batch_out_bytes.restype = C.c_size_t
        L.mij_batch_out_bytes.argtypes = [C.c_void_p]
        return L.mij_batch_out_bytes(self._h)

    def hash_out(self, slot):
        h = C.c_uint64()
        _check(lib().mij_batch_hash_out(self._h, int(slot), C.byref(h)), "mij_batch_hash_out")
        return h.value

    def diff_slots(self, pairs):
        """mij_batch_diff_slots: number of 16-byte words in which the device images of the slot pairs differ."""
        L = lib()
        n = len(pairs)
        L.mij_batch_diff_slots.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_uint64)]
        sa = (C.c_int * n)(*[int(p[0]) for p in pairs])
        sb = (C.c_int * n)(*[int(p[1]) for p in pairs])
        out = C.c_uint64()
        _check(L.mij_batch_diff_slots(self._h, sa, sb, n, C.byref(out)), "mij_batch_diff_slots")
        return out.value

    def slot_path(self, slot):
        return lib().mij_batch_slot_path(self._h, int(slot))

    def timer_begin(self):
        _check(lib().mij_batch_timer_begin(self._h), "mij_batch_timer_begin")

    def timer_end(self):
        _check(lib().mij_batch_timer_end(self._h), "mij_batch_timer_end")

    def timer_ms(self):
        ms = C.c_float()
        _check(lib().mij_batch_timer_elapsed_ms(self._h, C.byref(ms)), "mij_batch_timer_elapsed_ms")
        return ms.value

    def reset(self):
        self._es_used = 0
        _check(lib().mij_batch_reset(self._h), "mij_batch_reset")
        self.descs = []

    def close(self):
        if self._h:
            lib().mij_batch_destroy(self._h)
            self._h = C.c_void_p()


def decode_jpegs_multi(batches, datas, req_comp=0, threads=1):
    """mjh_decode_batch_multi: one logical batch sliced over several mij batches (one per device context), host
    walk on a shared pool, every batch submitted as soon as its slice is walked.  -> (n_ok, owner, slots, reasons)."""
    L = lib()
    n, nb = len(datas), len(batches)
    L.mjh_decode_batch_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p)]
    hs = (C.c_void_p * nb)(*[b._h for b in batches])
    bufs = (C.c_char_p * n)(*[bytes(d) for d in datas])
    lens = (C.c_int * n)(*[len(d) for d in datas])
    owner, slots, reasons = (C.c_int * n)(), (C.c_int * n)(), (C.c_char_p * n)()
    rc = L.mjh_decode_batch_multi(hs, nb, bufs, lens, n, int(req_comp), int(threads), owner, slots, reasons)
    if rc < 0:
        raise MijError("mjh_decode_batch_multi: %d %s" % (rc, L.mij_last_error().decode()))
    for i in range(n):
        sl = slots[i]
        real = sl if sl >= 0 else (-1 - sl if sl < -1 else None)
        if real is not None:
            b = batches[owner[i]]
            while len(b.descs) <= real:
                b.descs.append(None)
            b.descs[real] = (datas[i], req_comp)
    return rc, list(owner), list(slots), [r.decode() if r else None for r in reasons]


class PinnedBuffer:
    """mij_host_alloc / mij_host_free: page-locked host memory as a numpy uint8 view."""

    def __init__(self, nbytes):
        L = lib()
        L.mij_host_alloc.restype = C.c_void_p
        L.mij_host_alloc.argtypes = [C.c_size_t]
        L.mij_host_free.argtypes = [C.c_void_p]
        self.ptr = L.mij_host_alloc(C.c_size_t(nbytes))
        if not self.ptr:
            raise MijError("mij_host_alloc(%d) failed" % nbytes)
        self.nbytes = nbytes
        self.array = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_ubyte)), shape=(nbytes,))

    def close(self):
        if self.ptr:
            self.array = None
            lib().mij_host_free(C.c_void_p(self.ptr))
            self.ptr = None


# ---------------------------------------------------------------- encoder (config 5)

class WritePlan(C.Structure):
    """mjw_plan (include/mij_host.h)."""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("comp", C.c_int), ("subsample", C.c_int), ("mcu_x", C.c_int),
                ("mcu_y", C.c_int), ("du_per_mcu", C.c_int), ("ytab", C.c_ubyte * 64), ("ctab", C.c_ubyte * 64),
                ("fdtbl_y", C.c_float * 64), ("fdtbl_c", C.c_float * 64)]

    def du_elems(self):
        return self.mcu_x * self.mcu_y * self.du_per_mcu * 64


def host_transform(pixels, quality=90, flip=False):
    """mjw_plan_init + mjw_transform_host: the writer's data units computed on the HOST
    (int16 [n_du, 64], zigzag order) -- the CPU twin of Encoder, used to check it."""
    a = np.ascontiguousarray(pixels, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, comp = a.shape
    L = lib()
    plan = WritePlan()
    L.mjw_plan_init.argtypes = [C.POINTER(WritePlan), C.c_int, C.c_int, C.c_int, C.c_int]
    L.mjw_transform_host.argtypes = [C.POINTER(WritePlan), C.c_void_p, C.c_int, C.c_void_p]
    if not L.mjw_plan_init(C.byref(plan), w, h, comp, int(quality)):
        return None, None
    du = np.empty(plan.du_elems(), dtype=np.int16)
    L.mjw_transform_host(C.byref(plan), a.ctypes.data_as(C.c_void_p), int(bool(flip)), du.ctypes.data_as(C.c_void_p))
    return plan, du.reshape(-1, 64)


def emit_jpeg(plan, du):
    """mjw_emit: headers + Huffman stage over given data units -> bytes."""
    L = lib()
    L.mjw_emit.argtypes = [C.POINTER(WritePlan), C.c_void_p, _WRITE_CB, C.c_void_p]
    chunks = []
    cb = _WRITE_CB(lambda _c, data, size: chunks.append(C.string_at(data, size)))
    d = np.ascontiguousarray(du, dtype=np.int16)
    ok = L.mjw_emit(C.byref(plan), d.ctypes.data_as(C.c_void_p), cb, None)
    return b"".join(chunks) if ok else None


def mij_write_jpg_to_memory(pixels, quality=90):
    """mij_write_jpg_to_func: the writer with its transform stage on the GPU -> bytes (None on failure)."""
    a = np.ascontiguousarray(pixels, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, comp = a.shape
    L = lib()
    L.mij_write_jpg_to_func.argtypes = [_WRITE_CB, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    chunks = []
    cb = _WRITE_CB(lambda _c, data, size: chunks.append(C.string_at(data, size)))
    ok = L.mij_write_jpg_to_func(cb, None, w, h, comp, a.ctypes.data_as(C.c_void_p), int(quality))
    return b"".join(chunks) if ok else None


def mij_write_jpg_batch(images, quality=90, threads=16):
    """mij_write_jpg_batch: a list of uint8 pictures [h, w, comp] (or [h, w]; None = a NULL pixel pointer) -> list of byte streams
    (None where the picture was refused).  A negative return raises; the C side has released every stream by then."""
    L = lib()
    arrs = []
    for im in images:
        if im is None:
            arrs.append(None)
            continue
        a = np.ascontiguousarray(im, dtype=np.uint8)
        arrs.append(a[:, :, None] if a.ndim == 2 else a)
    n = len(arrs)
    L.mij_write_jpg_batch.restype = C.c_int
    L.mij_write_jpg_batch.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    px = (C.c_void_p * n)(*[(a.ctypes.data if a is not None else None) for a in arrs])
    xs = (C.c_int * n)(*[(a.shape[1] if a is not None else 16) for a in arrs])
    ys = (C.c_int * n)(*[(a.shape[0] if a is not None else 16) for a in arrs])
    cs = (C.c_int * n)(*[(a.shape[2] if a is not None else 3) for a in arrs])
    out = (C.c_void_p * n)()
    lens = (C.c_size_t * n)()
    rc = L.mij_write_jpg_batch(px, xs, ys, cs, n, int(quality), int(threads), out, lens)
    if rc < 0:
        assert not any(out[i] for i in range(n)), "mij_write_jpg_batch returned an error and left streams allocated"
        raise MijError("mij_write_jpg_batch: error %d" % rc)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    res = []
    for i in range(n):
        if out[i]:
            res.append(C.string_at(out[i], lens[i]))
            libc.free(out[i])
        else:
            res.append(None)
    assert rc == sum(r is not None for r in res), "mij_write_jpg_batch: return value and streams disagree"
    return res


class Encoder:
    """mij_encoder: batch colour + subsample + fDCT + quantiser on the GPU."""

    def __init__(self, ctx, max_images, pixel_bytes, du_bytes):
        L = lib()
        L.mij_enc_create.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]
        L.mij_enc_destroy.argtypes = [C.c_void_p]
        L.mij_enc_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.mij_enc_add_clone.argtypes = [C.c_void_p, C.c_int]
        for name in ("mij_enc_reset", "mij_enc_upload", "mij_enc_launch", "mij_enc_wait", "mij_enc_timer_begin", "mij_enc_timer_end"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.mij_enc_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.mij_enc_plan.argtypes = [C.c_void_p, C.c_int, C.POINTER(WritePlan)]
        L.mij_enc_timer_elapsed_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        self._h = C.c_void_p()
        _check(L.mij_enc_create(ctx._h, int(max_images), C.c_size_t(pixel_bytes), C.c_size_t(du_bytes), C.byref(self._h)), "mij_enc_create")

    def add(self, pixels, quality=90, flip=False):
        a = np.ascontiguousarray(pixels, dtype=np.uint8)
        if a.ndim == 2:
            a = a[:, :, None]
        h, w, comp = a.shape
        return _check(lib().mij_enc_add(self._h, a.ctypes.data_as(C.c_void_p), w, h, comp, int(quality), int(bool(flip))), "mij_enc_add")

    def add_clone(self, src):
        return _check(lib().mij_enc_add_clone(self._h, int(src)), "mij_enc_add_clone")

    def upload(self):
        _check(lib().mij_enc_upload(self._h), "mij_enc_upload")

    def launch(self):
        _check(lib().mij_enc_launch(self._h), "mij_enc_launch")

    def wait(self):
        _check(lib().mij_enc_wait(self._h), "mij_enc_wait")

    def force_generic(self, on=True):
        """Per-unit kernels even where the fused 4:2:0 strip kernel applies (tests); call before upload."""
        L = lib()
        L.mij_enc_force_generic.argtypes = [C.c_void_p, C.c_int]
        _check(L.mij_enc_force_generic(self._h, int(bool(on))), "mij_enc_force_generic")

    def plan(self, slot):
        p = WritePlan()
        _check(lib().mij_enc_plan(self._h, int(slot), C.byref(p)), "mij_enc_plan")
        return p

    def fetch(self, slot):
        p = self.plan(slot)
        du = np.empty(p.du_elems(), dtype=np.int16)
        _check(lib().mij_enc_fetch(self._h, int(slot), du.ctypes.data_as(C.c_void_p), C.c_size_t(du.size)), "mij_enc_fetch")
        return du.reshape(-1, 64)

    def timer_begin(self):
        _check(lib().mij_enc_timer_begin(self._h), "mij_enc_timer_begin")

    def timer_end(self):
        _check(lib().mij_enc_timer_end(self._h), "mij_enc_timer_end")

    def timer_ms(self):
        ms = C.c_float()
        _check(lib().mij_enc_timer_elapsed_ms(self._h, C.byref(ms)), "mij_enc_timer_elapsed_ms")
        return ms.value

    def close(self):
        if self._h:
            lib().mij_enc_destroy(self._h)
            self._h = C.c_void_p()
