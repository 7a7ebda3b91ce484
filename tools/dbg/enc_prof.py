import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import image_codecs_amd as ica
ctx = ica.Context()
L, kind = bench.cpu_checker()
W, H = 1920, 1080
for q in (90, 95):
    im = ica.synth_rgb(W, H, 0)
    enc = ica.Encoder(ctx, 2, 32 << 20, 64 << 20)
    s = enc.add(im, q)
    enc.upload(); enc.launch(); enc.wait()
    du = enc.fetch(s)
    host = ica.host_transform(im, q)[1]
    print("q", q, "gpu units == host units:", np.array_equal(du, host), "ndiff", int((du != host).sum()))
    mine = ica.emit_jpeg(enc.plan(s), du)
    fenc = L.ref_encode if kind == "reference" else L.orc_encode
    fenc.restype = C.c_long
    fenc.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    buf = np.zeros(W * H * 3, np.uint8)
    nb = fenc(buf.ctypes.data, buf.size, W, H, 3, np.ascontiguousarray(im).ctypes.data, q)
    print("   checker", kind, "bytes", nb, "mine", len(mine), "equal", mine == bytes(buf[:max(nb, 0)]))
    enc.close()
