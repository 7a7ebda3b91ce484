"""Randomised parity stress (not part of the pytest suites): many small random pictures of random sizes,
qualities, layouts (4:2:0 / 4:4:4 baseline from the writer, 4:2:2 / grey / progressive from the test-side
writer) and req_comp through every decode route -- fused kernels, the two-pass family, the GPU Huffman walk --
against the oracle.  python tests/stress.py [seconds] [seed] [enc]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # lives under tests/: it uses the oracle, which only test code may
import numpy as np  # noqa: E402
import image_codecs_amd as ica  # noqa: E402
import helpers  # noqa: E402


def picture(rng, w, h):
    kind = int(rng.integers(0, 4))
    if kind == 0:
        return rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    if kind == 1:
        return ica.synth_rgb(w, h, int(rng.integers(0, 1000)))
    if kind == 2:  # flat areas with hard edges (big AC, saturation)
        a = np.zeros((h, w, 3), np.uint8)
        for _ in range(6):
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            a[y0:y0 + int(rng.integers(1, h + 1)), x0:x0 + int(rng.integers(1, w + 1))] = rng.integers(0, 256, 3)
        return a
    g = np.linspace(0, 255, w)[None, :, None] * np.ones((h, 1, 3))
    return np.clip(g + rng.normal(0, 20, (h, w, 3)), 0, 255).astype(np.uint8)


def encoder_stress(budget, rng):
    """GPU data units (strip kernel where it applies, per-unit kernels, both forced) against the host transform."""
    ctx = ica.Context()
    t_end = time.time() + budget
    n = 0
    while time.time() < t_end:
        imgs, qs, flips = [], [], []
        for _ in range(16):
            w = int(rng.integers(1, 80)) * 8 if rng.random() < 0.6 else int(rng.integers(1, 500))  # multiples of 8 / 16: the fused strip kernels
            h = int(rng.integers(1, 400))
            c = int(rng.choice([1, 3, 3, 3, 4]))
            img = picture(rng, w, h)
            img = img[:, :, :c] if c <= 3 else np.concatenate([img, img[:, :, :1]], -1)
            imgs.append(np.ascontiguousarray(img))
            qs.append(int(rng.choice([1, 20, 50, 75, 90, 91, 100])))
            flips.append(bool(rng.integers(0, 2)))
        want = [ica.host_transform(im[::-1] if f else im, q)[1] for im, q, f in zip(imgs, qs, flips)]
        for generic in (False, True):
            enc = ica.Encoder(ctx, len(imgs), 64 << 20, 64 << 20)
            enc.force_generic(generic)
            slots = [enc.add(im, q, flip=f) for im, q, f in zip(imgs, qs, flips)]
            enc.upload()
            enc.launch()
            enc.wait()
            for s_, w_, im in zip(slots, want, imgs):
                if not np.array_equal(enc.fetch(s_), w_):
                    raise SystemExit("ENCODER MISMATCH generic=%s shape=%s" % (generic, im.shape))
                n += 1
            enc.close()
        if n % 3200 == 0:
            print("  ... %d encoder comparisons" % n, flush=True)
    print("encoder stress ok: %d comparisons in %.0f s" % (n, budget))


LAYOUTS = [([(1, 2), (1, 1), (1, 1)], -1), ([(4, 1), (1, 1), (1, 1)], -1), ([(4, 2), (1, 1), (1, 1)], -1), ([(2, 4), (1, 1), (1, 1)], -1),
           ([(1, 4), (1, 1), (1, 1)], -1), ([(1, 1), (1, 1), (1, 1)], 0), ([(2, 2), (1, 1), (1, 1)], 0), ([(1, 1)] * 4, 0), ([(1, 1)] * 4, 2),
           ([(2, 2), (1, 1), (1, 1), (2, 2)], 2), ([(2, 1), (1, 1), (1, 1), (2, 1)], 0), ([(1, 1)] * 4, 1), ([(2, 2), (1, 1), (1, 1), (1, 1)], 2),
           ([(1, 1), (2, 2), (2, 2)], -1), ([(3, 1), (1, 1), (1, 1)], -1), ([(2, 2), (2, 1), (1, 2)], -1)]


def main():
    if len(sys.argv) > 3 and sys.argv[3] == "enc":
        return encoder_stress(float(sys.argv[1]), np.random.default_rng(int(sys.argv[2])))
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    oracle = helpers.Oracle()
    ctx = ica.Context()
    t_end = time.time() + budget
    n_img = n_cmp = rounds = 0
    while time.time() < t_end:
        rounds += 1
        datas = []
        for _ in range(24):
            w = int(rng.integers(1, 96)) if rng.random() < 0.3 else int(rng.integers(1, 700))
            if rng.random() < 0.4:
                w = (w + 3) & ~3  # multiples of four take the specialised pass 2 of the two-pass family
            h = int(rng.integers(1, 96)) if rng.random() < 0.3 else int(rng.integers(1, 500))
            q = int(rng.choice([1, 10, 35, 50, 75, 85, 90, 91, 95, 100]))
            if rng.random() < 0.05:  # a row of MCUs beyond the LDS of a CU: the band kernels in column segments (round 3)
                w, h = int(rng.integers(4200, 13000)), int(rng.integers(1, 56))
                img = picture(rng, w, h)
                if rng.random() < 0.5:
                    datas.append(ica.stbi_write_jpg_to_memory(img, min(q, 90)))  # 4:2:0
                else:
                    plan, du = ica.host_transform(img, max(q, 91))
                    datas.append(helpers.baseline_layout_from_444(plan, du, [(1, 2), (1, 1), (1, 1)], -1, int(rng.choice([0, 0, 7]))))  # 4:4:0
                continue
            img = picture(rng, w, h)
            layout = int(rng.integers(0, 12))
            if layout <= 1:
                datas.append(ica.stbi_write_jpg_to_memory(img, q))
            else:
                plan, du = ica.host_transform(img, q if layout in (2, 6) else max(q, 91))
                script = int(rng.integers(0, 2))
                if layout == 2:
                    datas.append(helpers.progressive_from_du(plan, du, script))
                elif layout == 3:
                    datas.append(helpers.progressive_422_from_444(plan, du, script))
                elif layout == 4:
                    datas.append(helpers.progressive_grey_from_444(plan, du, script))
                elif layout == 5:
                    datas.append(helpers.progressive_from_du(plan, du, script))
                elif layout >= 9:  # sampling factors / colour tags that only the two-pass family takes (round 2)
                    hv, app14 = LAYOUTS[int(rng.integers(0, len(LAYOUTS)))]
                    datas.append(helpers.baseline_layout_from_444(plan, du, hv, app14, int(rng.choice([0, 0, 3, 40]))))
                else:  # baseline with optimal tables and a random restart interval, in a random layout
                    lay = ["native", "422", "grey"][layout - 6]
                    dri = int(rng.choice([0, 1, 2, 5, 16, 100, 5000]))
                    datas.append(helpers.baseline_from_du(plan, du, dri, lay))
        req = int(rng.integers(0, 5))
        wants = [oracle.load(d, req) for d in datas]
        for mode in ("fused", "fused_int16", "generic", "generic2", "gpu_walk"):
            b = ica.Batch(ctx, len(datas), 96 << 20, 96 << 20, 96 << 20)
            if mode == "fused_int16":  # int16 tile-layout planes: int16 staging, block classes computed from the int16 block
                b.set_coef_format("int16")
            if mode == "gpu_walk":
                b.entropy_reserve(8 << 20)
            b.force_generic(1 if mode == "generic" else (2 if mode == "generic2" else 0))
            ok, slots, reasons = b.decode_jpegs(datas, req, threads=4, gpu_entropy=(mode == "gpu_walk"))
            b.submit()
            b.wait()
            for i, (kind, want, _) in enumerate(wants):
                if kind != "ok":
                    assert slots[i] < 0, (mode, i)
                    continue
                assert slots[i] >= 0, (mode, i, reasons[i])
                got = b.fetch(slots[i])
                if not np.array_equal(got.reshape(-1), want.reshape(-1)):
                    open(os.path.join(ROOT, "gpurun_out", "stress_fail_%d_%d.jpg" % (rounds, i)), "wb").write(datas[i])
                    raise SystemExit("MISMATCH mode=%s round=%d image=%d shape=%s req=%d path=%d" % (mode, rounds, i, want.shape, req, b.slot_path(slots[i])))
                n_cmp += 1
            b.close()
        # the one-picture path of the public API, Huffman walk on the GPU whatever the size (1024-bit subsequences)
        os.environ["MIJ_GPU_WALK_MIN_PIXELS"] = "0" if rounds % 2 else "480000"
        for i in range(0, len(datas), 5):
            got = ica.stbi_load_from_memory(datas[i], req)
            kind, want, _ = wants[i]
            if (got is None) != (kind != "ok") or (got is not None and not np.array_equal(got[0].reshape(-1), want.reshape(-1))):
                open(os.path.join(ROOT, "gpurun_out", "stress_fail_single_%d_%d.jpg" % (rounds, i)), "wb").write(datas[i])
                raise SystemExit("MISMATCH stbi_load_from_memory round=%d image=%d req=%d" % (rounds, i, req))
            n_cmp += 1
        n_img += len(datas)
        if rounds % 100 == 0:
            print("  ... %d rounds, %d comparisons" % (rounds, n_cmp), flush=True)
    print("stress ok: %d rounds, %d pictures, %d comparisons in %.0f s" % (rounds, n_img, n_cmp, budget))


if __name__ == "__main__":
    main()
