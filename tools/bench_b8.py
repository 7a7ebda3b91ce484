"""Fused 4:2:0 kernel on byte-coefficient planes (experimental, MIJ_COEF_BYTES=1) against int16 planes: N x 1080p
walked by the GPU entropy stage, then the kernel timed with the planes resident.  The achieved figure keeps the
ALGORITHMIC bytes (int16 coefficients) in the numerator; actual traffic is lower with byte planes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_codecs_amd as ica  # noqa: E402


def run(n, byte_planes):
    if byte_planes:
        os.environ["MIJ_COEF_BYTES"] = "1"
    else:
        os.environ.pop("MIJ_COEF_BYTES", None)
    w, h = 1920, 1080
    ctx = ica.Context()
    datas = [ica.synth_jpeg(w, h, s, 90) for s in range(16)]
    d = ica.HostDecoder.probe(datas[0], 3)
    cb, ob = ica.Batch.coef_bytes(d), ica.Batch.out_bytes(d)
    b = ica.Batch(ctx, n, cb * 2, cb * n, ob * n)
    b.entropy_reserve(sum(len(x) * 9 // 8 + 4352 for x in datas))
    ok, slots, reasons = b.decode_jpegs(datas, 3, threads=8, gpu_entropy=True)
    assert ok == len(datas), reasons
    for i in range(len(datas), n):
        b.add_clone(slots[i % len(datas)])
    b.submit()
    for _ in range(3):
        b.launch()
    b.wait()
    hashes = [b.hash_out(s) for s in slots[:4]] + [b.hash_out(n - 1)]
    b.timer_begin()
    for _ in range(15):
        b.launch()
    b.timer_end()
    ms = b.timer_ms() / 15
    b.close()
    ctx.close()
    return ms, hashes


def main():
    n = int(os.environ.get("BB8_N", "1024"))
    out = {}
    ref = None
    for name, flag in (("int16", False), ("bytes", True), ("int16_again", False), ("bytes_again", True)):
        ms, hashes = run(n, flag)
        ref = ref or hashes
        assert hashes == ref, "pixels differ between the coefficient formats"
        out[name] = {"ms_per_launch": round(ms, 4), "frac_of_8TBs": round(12487680 * n / ms / 1e6 / 8000, 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
