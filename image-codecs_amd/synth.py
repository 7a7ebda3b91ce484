"""Deterministic synthetic inputs: gradient + LCG noise, encoded by the product's
own stbi_write_jpg_to_func (the restatement of codec/jpeg_write.c), so that the GPU box needs no
image files, no PIL and no network."""
import numpy as np

from . import binding


def synth_rgb(width, height, seed=0, noise_mask=15):
    """Image `seed`: per pixel in raster order state = state*1664525 + 1013904223 (mod 2^32),
    n = (state >> 24) & noise_mask; R = min(255, x*255//W + n), G = min(255, y*255//H + n),
    B = min(255, (x+y)*255//(W+H) + n).  LCG start state = 12345 + seed."""
    n_px = width * height
    # closed form of the LCG: state_k = a^k * s0 + c * (a^k - 1) / (a - 1)  (mod 2^32), built by doubling
    a, c = np.uint64(1664525), np.uint64(1013904223)
    mask = np.uint64(0xFFFFFFFF)
    states = np.empty(n_px, dtype=np.uint64)
    s = np.uint64((12345 + seed) & 0xFFFFFFFF)
    # vectorised generation in blocks: state_{i+k} = A_k * state_i + C_k
    block = 1 << 12
    mul = np.empty(block, dtype=np.uint64)
    add = np.empty(block, dtype=np.uint64)
    m, d = np.uint64(1), np.uint64(0)
    for i in range(block):
        m = (m * a) & mask
        d = (d * a + c) & mask
        mul[i], add[i] = m, d
    pos = 0
    while pos < n_px:
        k = min(block, n_px - pos)
        states[pos:pos + k] = (mul[:k] * s + add[:k]) & mask
        s = states[pos + k - 1]
        pos += k
    noise = ((states >> np.uint64(24)) & np.uint64(noise_mask)).astype(np.int32).reshape(height, width)
    x = np.arange(width, dtype=np.int64)[None, :]
    y = np.arange(height, dtype=np.int64)[:, None]
    r = np.minimum(255, x * 255 // width + noise)
    g = np.minimum(255, y * 255 // height + noise)
    b = np.minimum(255, (x + y) * 255 // (width + height) + noise)
    return np.stack([r, g, b], axis=-1).astype(np.uint8)


def synth_jpeg(width, height, seed=0, quality=90, noise_mask=15):
    """Baseline JFIF bytes of synth_rgb (quality <= 90 -> 4:2:0, codec/jpeg_write.c:221)."""
    data = binding.stbi_write_jpg_to_memory(synth_rgb(width, height, seed, noise_mask), quality)
    if data is None:
        raise binding.MijError("stbi_write_jpg_to_func failed")
    return data


def synth_rgb_edges(width, height, seed=0, noise_mask=63, rects=24):
    """The harsher declared content of the benchmark's second leg: synth_rgb with noise_mask 63 (four times the noise
    amplitude: ~3.5 bit/px at q=90) and `rects` axis-aligned rectangles whose pixels are inverted (255 - v): hard edges
    of full contrast, whose low-frequency coefficients leave the byte range at q=90 (escaped blocks of the compact
    planes).  Rectangle k of image `seed`: LCG state s0 = 777 + 1000003*seed, four draws per rectangle
    (s = s*1664525 + 1013904223 mod 2^32): x0 = (s>>8) % W, y0 = (s>>8) % H, w = 16 + (s>>8) % (W/3), h = 16 + (s>>8) % (H/3),
    clipped to the picture."""
    img = synth_rgb(width, height, seed, noise_mask).astype(np.int32)
    s = (777 + 1000003 * seed) & 0xFFFFFFFF

    def draw():
        nonlocal s
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        return s >> 8

    for _ in range(rects):
        x0, y0 = draw() % width, draw() % height
        w, h = 16 + draw() % max(1, width // 3), 16 + draw() % max(1, height // 3)
        img[y0:min(height, y0 + h), x0:min(width, x0 + w)] = 255 - img[y0:min(height, y0 + h), x0:min(width, x0 + w)]
    return img.astype(np.uint8)
