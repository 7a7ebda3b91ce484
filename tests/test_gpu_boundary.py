"""The rest of the drop-in boundary on the GPU path: BASELINE config 1 through stbi_load(path),
every golden stream through stbi_load_from_callbacks with short reads, the FILE* position after
stbi_load_from_file (convert.c:199-211), stbi_info_from_file / _callbacks (image_api.c:85-131),
stbi_write_jpg(filename) (codec/jpeg_write.c:376-388), and the colour branch after late JFIF / Adobe markers
(codec/jpeg.c:2234-2244) through every front end."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

UNINIT_IN_REFERENCE = {"dri_without_rst"}


@pytest.fixture(scope="module")
def g2():
    return helpers.GoldenR2()


def test_config1_512x512_through_stbi_load_path(ica, oracle, gpu_ctx, g2, tmp_path):
    """BASELINE configs[0]: a single 512x512 baseline 4:2:0 q=90 JPEG through stbi_load(filename)."""
    data = bytes(g2["cfg1/jpg"])
    p = tmp_path / "cfg1.jpg"
    p.write_bytes(data)
    assert ica.stbi_info(str(p)) == (1, 512, 512, 3)
    for req in (0, 1, 3, 4):
        got = ica.stbi_load(str(p), req)
        assert got is not None, ica.stbi_failure_reason()
        px, w, h, comp = got
        assert (w, h, comp) == (512, 512, 3)
        assert np.array_equal(px[:8], g2["cfg1/head%d" % req])               # the reference's own first rows
        assert helpers.fnv1a64(px) == int(g2["cfg1/fnv%d" % req][0])         # and its hash of all of them
        assert np.array_equal(px, oracle.load(data, req)[1])
    # the same picture made here (the product's writer == the reference's, tests/test_gpu_encode.py) decodes the same
    mine = ica.synth_jpeg(512, 512, seed=1, quality=90)
    assert mine == data


def test_every_golden_through_callbacks_with_short_reads(golden, ica, oracle, gpu_ctx, g2):
    """read() returns 1..128 bytes at a time (helpers.CB_PATTERNS): the 128-byte buffer refill of common.c:10-26 must
    land the same bytes, and the verdict must be the REFERENCE's under the same read pattern (recorded by
    make_golden_r2.py) -- including its "no SOI" when the very first read is shorter than the SOI marker, because it
    rewinds into a buffer the second read has overwritten."""
    rows = [r.split("\t") for r in bytes(g2["callbacks"]).decode().split("\n")]
    n_ok = n_fail = n_quirk = 0
    for pi, name, kind, what in rows:
        data = golden.jpg(name)
        got = ica.stbi_load_from_callbacks(data, 3, chunk=helpers.CB_PATTERNS[int(pi)])
        if kind == "fail":
            assert got is None and ica.stbi_failure_reason() == what, (pi, name, ica.stbi_failure_reason(), what)
            n_fail += 1
            n_quirk += what == "no SOI" and golden.expect(name, 3)[0] == "ok"
        else:
            assert got is not None, (pi, name, ica.stbi_failure_reason())
            if name in UNINIT_IN_REFERENCE:
                assert np.array_equal(got[0], oracle.load(data, 3)[1]), (pi, name)
            else:
                assert helpers.fnv1a64(got[0]) == int(what), (pi, name)
                if int(pi) == 0:
                    assert np.array_equal(got[0], golden.expect(name, 3)[1]), name
            n_ok += 1
    assert n_ok > 100 and n_fail > 20 and n_quirk > 20, (n_ok, n_fail, n_quirk)
    # req_comp 0 and 4 through one pattern that the reference survives
    for name in ("b420_64x64_q90", "b422_37x21", "prog_444_64x64", "grey_33x20", "cmyk_40x30"):
        for req in (0, 4):
            got = ica.stbi_load_from_callbacks(golden.jpg(name), req, chunk=helpers.CB_PATTERNS[2])
            assert np.array_equal(got[0], golden.expect(name, req)[1]), (name, req)


def test_file_position_after_load_and_info(ica, gpu_ctx, g2, tmp_path):
    """stbi_load_from_file leaves the FILE* just behind what the decoder consumed (convert.c:208), on failure where the
    reads stopped; stbi_info_from_file puts it back where it was (image_api.c:92)."""
    for key in g2.filepos_names:
        ok, pos_load, ok_info7, pos_info7 = [int(v) for v in g2["filepos/%s" % key]]
        data = bytes(g2["filepos/%s/jpg" % key])
        p = tmp_path / "f.jpg"
        p.write_bytes(data)
        got, pos = ica.stbi_load_from_file(str(p), 3)
        assert (got is not None) == bool(ok), key
        assert pos == pos_load, (key, pos, pos_load)
        (oki, w, h, c), posi = ica.stbi_info_from_file(str(p), 7)
        assert (oki, posi) == (ok_info7, pos_info7), key
        (oki, w, h, c), posi = ica.stbi_info_from_file(str(p), 0)
        assert posi == 0
        if ok:
            assert oki == 1 and (w, h) == (got[1], got[2])
            assert ica.stbi_info_from_callbacks(data, chunk=lambda k: 3 + k % 11) == (1, w, h, c)


def test_stbi_write_jpg_filename(ica, golden, tmp_path):
    """stbi_write_jpg(filename) writes the byte stream stbi_write_jpg_to_func produces: the reference's golden bytes."""
    for name in golden.enc_names:
        rgb, q = golden[name + "/rgb"], int(np.asarray(golden[name + "/q"]).reshape(-1)[0])
        p = tmp_path / "o.jpg"
        assert ica.stbi_write_jpg(str(p), rgb, q) == 1
        assert p.read_bytes() == bytes(golden[name + "/jpg"]), name
    assert ica.stbi_write_jpg(str(tmp_path / "no_such_dir" / "o.jpg"), golden["enc/16x16x3_q90/rgb"], 90) == 0


def test_late_markers_through_every_front_end(ica, oracle, gpu_ctx, g2):
    datas = [g2.late(name, 3)[0] for name in g2.late_names]
    for req in range(5):
        for name in g2.late_names:
            data, kind, want = g2.late(name, req)
            got = ica.stbi_load_from_memory(data, req)
            assert kind == "ok" and got is not None, (name, req)
            assert np.array_equal(got[0], want), (name, req, "stbi_load_from_memory")
    for gpu_entropy in (False, True):
        for req in (3, 4, 1):
            b = ica.Batch(gpu_ctx, len(datas), 16 << 20, 16 << 20, 16 << 20)
            if gpu_entropy:
                b.entropy_reserve(2 << 20)
            ok, slots, reasons = b.decode_jpegs(datas, req, threads=3, gpu_entropy=gpu_entropy)
            assert ok == len(datas), reasons
            b.submit()
            b.wait()
            for name, s in zip(g2.late_names, slots):
                assert np.array_equal(b.fetch(s), g2.late(name, req)[2]), (name, req, gpu_entropy)
            b.close()
    # Batch.add_jpeg (python-side header probe, then the walk) follows the late marker too
    b = ica.Batch(gpu_ctx, len(datas), 16 << 20, 16 << 20, 16 << 20)
    slots = [b.add_jpeg(d, 3) for d in datas]
    b.submit()
    b.wait()
    for name, s in zip(g2.late_names, slots):
        assert np.array_equal(b.fetch(s), g2.late(name, 3)[2]), name
    b.close()


def test_stbi_load_from_many_host_threads(golden, ica, oracle, gpu_ctx):
    """The reference is re-entrant apart from its global reason string (threading): a drop-in may be called from many
    host threads at once, one picture each.  Eight threads decode interleaved lists -- small pictures (host walk), pictures above the
    GPU-walk threshold, damaged and rejected streams -- through stbi_load_from_memory; every result equals the oracle's and every
    failure reason is the one of the calling thread's own last call (thread-local), whatever the other threads were doing."""
    import threading
    good = [ica.synth_jpeg(w, h, seed=w + h, quality=q) for (w, h, q) in ((64, 48, 90), (640, 360, 75), (1920, 1080, 90), (1280, 1024, 92), (333, 211, 95), (2048, 1536, 85))]
    bad = [golden.jpg("garbage"), golden.jpg("b420_64x64_q90")[:200], b"\xff\xd8\xff\xdb\x00\x02"]
    datas = good + bad
    wants = [oracle.load(d, 3) for d in datas]
    errors = []

    def worker(t):
        try:
            for rep in range(6):
                for k in range(len(datas)):
                    i = (k * 5 + t + rep) % len(datas)
                    got = ica.stbi_load_from_memory(datas[i], 3)
                    kind, want, _ = wants[i]
                    if kind == "ok":
                        if got is None or not np.array_equal(got[0], want):
                            errors.append((t, rep, i, "pixels"))
                    else:
                        why = ica.stbi_failure_reason()
                        if got is not None or why != want:
                            errors.append((t, rep, i, why, want))
        except Exception as exc:  # noqa: BLE001
            errors.append((t, repr(exc)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]
