#!/usr/bin/env python3
"""Generates tests/golden/writer_golden_r3.npz from the REAL reference (round 3: the writer, pinned so that neither side of a
comparison has to be computed in the process under test).

Build container only: needs oracle/_ref/libstbref.so (the reference compiled in place by oracle/Makefile).  Everything stored is
DATA -- what the reference itself returned for declared inputs:

  bench/q<q>/len, bench/q<q>/sha256   the reference writer's stream (codec/jpeg_write.c:368) for the benchmark's pictures
                                      synth_rgb(1920, 1080, seed), seeds 0..15 at quality 90, seeds 0..3 at quality 95:
                                      length and SHA-256 per seed (the pictures are regenerated from their seed by the test)
  small/<name>/rgb, q, jpg            small pictures with the reference's stream in full
  small/<name>/units                  the quantised data units of that stream, int16 [n_du, 64], zigzag order, in the writer's
                                      order (MCU after MCU): recovered from the REFERENCE's stream by the REFERENCE's decoder
                                      (ref_decode_capture: the de-quantised blocks it hands to its IDCT seam, codec/jpeg.c:83,
                                      divided by the stream's own DQT entries) -- no code of this repository is involved
  small/<name>/ytab, ctab             the stream's two DQT tables as the writer stores them (zigzag order, codec/jpeg_write.c:226-236)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import image_codecs_amd as ica  # noqa: E402  (synth_rgb only: the declared picture generator, pure numpy)

# natural index -> zigzag position, read off the DQT / unit order the reference writes (JPEG Annex A figure 5)
ZIGZAG_OF = np.array([0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
                      10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63])


def dqt_tables(jpg):
    """the two 64-byte tables of the reference writer's single DQT segment (zigzag order, as stored)"""
    i = jpg.index(b"\xff\xdb")
    assert jpg[i + 2:i + 4] == b"\x00\x84" and jpg[i + 4] == 0 and jpg[i + 69] == 1
    return np.frombuffer(jpg[i + 5:i + 69], np.uint8).copy(), np.frombuffer(jpg[i + 70:i + 134], np.uint8).copy()


def units_of(jpg, sub):
    """quantised units of a reference stream through the reference's own decoder"""
    ytab, ctab = dqt_tables(jpg)
    coef = mg.ref_coef(jpg)  # de-quantised, natural order, in the order of the IDCT calls = MCU order for one interleaved scan
    coef = np.asarray(coef, np.int32).reshape(-1, 64)
    per = 6 if sub else 3
    assert coef.shape[0] % per == 0
    units = np.zeros_like(coef)
    for k in range(coef.shape[0]):
        chroma = (k % per) >= (4 if sub else 1)
        q = (ctab if chroma else ytab).astype(np.int32)  # zigzag order
        qn = q[ZIGZAG_OF]  # natural order
        assert np.all(coef[k] % qn == 0), "a de-quantised coefficient is not a multiple of its quantiser (int16 wrap?)"
        units[k, ZIGZAG_OF] = coef[k] // qn
    assert np.abs(units).max() < 16384
    return units.astype(np.int16), ytab, ctab


def main():
    out = {}
    for q, seeds in ((90, range(16)), (95, range(4))):
        lens, shas = [], []
        for s in seeds:
            jpg = mg.ref_encode(ica.synth_rgb(1920, 1080, s), q)
            lens.append(len(jpg))
            shas.append(np.frombuffer(hashlib.sha256(jpg).digest(), np.uint8))
        out["bench/q%d/len" % q] = np.array(lens, np.int64)
        out["bench/q%d/sha256" % q] = np.stack(shas)
        print("bench q%d: lengths %s" % (q, lens))
    cases = [("s64_q90", 64, 64, 1, 90), ("s256_q90", 256, 256, 2, 90), ("s256_q95", 256, 256, 3, 95), ("s200x120_q50", 200, 120, 4, 50),
             ("s33x17_q75", 33, 17, 5, 75), ("s128_q100", 128, 128, 6, 100), ("edges160_q90", 160, 96, 7, 90)]
    names = []
    for name, w, h, seed, q in cases:
        img = ica.synth_rgb_edges(w, h, seed) if name.startswith("edges") else ica.synth_rgb(w, h, seed, noise_mask=31)
        jpg = mg.ref_encode(img, q)
        units, ytab, ctab = units_of(jpg, q <= 90)
        pre = "small/" + name
        out[pre + "/rgb"] = img
        out[pre + "/q"] = np.array([q], np.int32)
        out[pre + "/jpg"] = np.frombuffer(jpg, np.uint8)
        out[pre + "/units"] = units
        out[pre + "/ytab"] = ytab
        out[pre + "/ctab"] = ctab
        names.append(name)
        print("%s: %d bytes, %d units, |unit| max %d" % (name, len(jpg), units.shape[0], np.abs(units).max()))
    out["small/names"] = np.array(names)
    path = os.path.join(HERE, "writer_golden_r3.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
