#!/usr/bin/env python3
"""bench.py -- JPEG decode Mpixels/sec on a 4:2:0 1080p batch (BASELINE.json's metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A *step* is one pass of the GPU hot path -- de-quantise + 8x8 integer IDCT + h2v2 upsample +
YCbCr->RGB, i.e. everything the reference does between the Huffman walk and the pixel buffer
(codec/jpeg.c:325-365 dequant, :615-679, :1816-1840, :1976-2018, :2301-2432) -- over one batch of
synthetic 1920x1080 4:2:0 q=90 JPEGs whose quantised coefficients are ALREADY RESIDENT IN HBM
when the timed region starts (host Huffman walk + H2D happen before it; see DESIGN.md for the
PCIe/host-inclusive rate, which is never `value`).  Each rank owns `--images` images (default
1024 = BASELINE configs[1]) in its own device buffers -- every slot has its own coefficient and
pixel memory (12.8 GB per GPU >> the 256 MiB Infinity Cache) -- and there is no data-path
collective: images are independent (weak scaling).  The resident planes of the timed batch are byte-coefficient planes
written by the (experimental) GPU Huffman walk -- AC coefficients as signed bytes, DC aside: half the coefficient bytes --
whenever they reproduce, in this very run, the pixels of the north-star pipeline (host walk -> int16 planes); that
pipeline's own kernel figure is reported beside it as `roofline_int16_planes`, and it is the fallback (MIJ_BENCH_INT16=1
forces it).  `roofline.achieved` keeps the ALGORITHMIC bytes (int16 coefficients) in the numerator either way.

The one JSON line (rank 0) also carries
  roofline      the fused kernel against the HBM roofline: algorithmic bytes per launch
                (2 B x 64 x blocks read + 3 B x pixels written = 12 487 680 B per image) divided by
                the average launch duration measured HERE with HIP events on the kernel's stream.
  cpu_baseline  the reference itself (oracle/_ref, compiled in place in the build container and
                shipped as a .so) -- or, if that .so is absent, our CPU restatement (oracle/) --
                decoding a bounded sample of the same images on this box's host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
ALGO_BYTES_PER_IMAGE = 2 * 64 * 48960 + 3 * W * H  # 6 266 880 + 6 220 800
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def usable_cores():
    """Host threads this process may really use: the affinity mask, capped by the cgroup CPU quota
    (a GPU box shows every core of the host in the mask but grants a 1-GPU job only a share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, (q + p // 2) // p))
        except (OSError, ValueError):
            pass
    env = os.environ.get("BENCH_CPU_THREADS")
    if env:
        n = max(1, int(env))
    return n


def cpu_baseline(datas, want_seconds=12.0, gpu_pixels=None):
    """Times the CPU checker on a bounded sample (all host cores, one image per task); with gpu_pixels(i) it also
    acts as what it is -- the checker: its pixels for every distinct image must equal the GPU's."""
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libstbref.so")
    port_so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if os.path.exists(ref_so):
        L, fn, kind = C.CDLL(ref_so), "ref_decode_many", "reference"
    else:
        if not os.path.exists(port_so):
            import subprocess
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True)
        L, fn, kind = C.CDLL(port_so), "orc_decode_many", "port"
    f = getattr(L, fn)
    f.restype = C.c_long
    f.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    n = len(datas)
    bufs = (C.c_char_p * n)(*datas)
    lens = (C.c_int * n)(*[len(d) for d in datas])
    secs = C.c_double()
    cores = usable_cores()
    # single thread, one pass over the sample
    px1 = f(bufs, lens, n, 1, 1, 3, C.byref(secs))
    t1 = secs.value
    single = px1 / t1 / 1e6
    # all usable cores: calibrate with a short run, then size the real one for ~want_seconds of wall time
    cal_reps = max(1, (2 * cores + n - 1) // n)
    f(bufs, lens, n, cal_reps, cores, 3, C.byref(secs))
    reps = max(cal_reps, min(int(want_seconds / max(secs.value, 1e-3) * cal_reps), 4096))
    px = f(bufs, lens, n, reps, cores, 3, C.byref(secs))
    parity = None
    if gpu_pixels is not None:
        load = L.stbi_load_from_memory if kind == "reference" else L.orc_load_from_memory
        load.restype = C.POINTER(C.c_ubyte)
        ints = [C.POINTER(C.c_int)] * 3
        load.argtypes = [C.c_char_p, C.c_int] + ints + [C.c_int] + ([C.POINTER(C.c_char_p)] if kind == "port" else [])
        parity = True
        for i, d in enumerate(datas):
            x, y, c = C.c_int(), C.c_int(), C.c_int()
            extra = [C.byref(C.c_char_p())] if kind == "port" else []
            ptr = load(d, len(d), C.byref(x), C.byref(y), C.byref(c), 3, *extra)
            cpu = np.ctypeslib.as_array(ptr, shape=(y.value * x.value * 3,))
            if not np.array_equal(cpu, gpu_pixels(i).reshape(-1)):
                parity = False
            (L.stbi_image_free if kind == "reference" else L.orc_free)(ptr)
        assert parity, "GPU pixels differ from the CPU checker's"
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "parity_with_gpu_output": parity,
        "cpu_model": model,
        "value": round(px / secs.value / 1e6, 1),
        "unit": "Mpix/s",
        "cores": cores,
        "kind": kind,
        "sample": "%d distinct 1920x1080 4:2:0 q90 images x %d passes, one image per task on %d threads (%.1f s); single thread: %.1f Mpix/s"
                  % (n, reps, cores, secs.value, single),
        "single_thread_mpix_s": round(single, 1),
    }


def end_to_end(ica, ctx, datas, distinct, n_img, src_hash, cbytes, obytes, args):
    """Outside the timed region (rank 0, one GPU): bitstream in host RAM -> RGB in HBM, three ways."""
    e2e = None
    n_e = min(args.e2e_images, n_img)
    threads = usable_cores()
    chunk = max(1, min(64, n_e // 2))
    # two batches ping-pong: while the GPU uploads and transforms chunk k (async on its own stream)
    # the host threads already walk chunk k+1 -- the "H2D / kernel overlap" of SURVEY 8(e)
    ebs = [ica.Batch(ctx, chunk, cbytes * chunk, cbytes * chunk, obytes * chunk) for _ in range(2)]
    jl = [datas[i % distinct] for i in range(n_e)]
    for eb in ebs:  # warm the pool / page in staging
        eb.decode_jpegs(jl[:chunk], 3, threads)
        eb.submit()
        eb.wait()
    t_host = 0.0
    t0 = time.perf_counter()
    last = {}
    for k, lo in enumerate(range(0, n_e, chunk)):
        eb = ebs[k & 1]
        eb.reset()  # waits for this batch's previous chunk
        part = jl[lo:lo + chunk]
        th = time.perf_counter()
        ok, slots, reasons = eb.decode_jpegs(part, 3, threads)
        t_host += time.perf_counter() - th
        assert ok == len(part), reasons
        eb.submit()
        last[k & 1] = (lo + len(part) - 1, slots[len(part) - 1])
    for eb in ebs:
        eb.wait()
    t_all = time.perf_counter() - t0
    for side, (img, slot) in last.items():
        assert ebs[side].hash_out(slot) == src_hash[img % distinct]
    e2e = {
        "value": round(n_e * W * H / t_all / 1e6, 1),
        "unit": "Mpix/s",
        "images": n_e,
        "host_threads": threads,
        "chunk_images": chunk,
        "host_stage_only_mpix_s": round(n_e * W * H / t_host / 1e6, 1),
        "includes": "Huffman walk on the host threads -> pinned staging -> H2D -> fused kernel, two batches ping-pong; pixels left in HBM",
    }
    # the same pipeline with the pixels brought back to (pinned) host memory: one asynchronous D2H of the
    # chunk's output arena queued behind its kernels, completed when the batch is reused
    pins = [ica.PinnedBuffer(obytes * chunk) for _ in range(2)]
    t0 = time.perf_counter()
    for k, lo in enumerate(range(0, n_e, chunk)):
        eb = ebs[k & 1]
        eb.reset()
        part = jl[lo:lo + chunk]
        ok, slots, reasons = eb.decode_jpegs(part, 3, threads)
        eb.submit()
        eb.fetch_all_async(pins[k & 1].ptr, pins[k & 1].nbytes)
        last[k & 1] = (lo + len(part) - 1, slots[len(part) - 1])
    for eb in ebs:
        eb.wait()
    t_d2h = time.perf_counter() - t0
    for side, (img, slot) in last.items():
        off = ebs[side].out_offset(slot)
        got = pins[side].array[off:off + W * H * 3]
        assert np.array_equal(got, ebs[side].fetch(slot).reshape(-1)), "D2H copy differs from the device image"
    e2e["value_with_d2h"] = round(n_e * W * H / t_d2h / 1e6, 1)
    # experimental: the Huffman walk itself on the GPU (mjh_decode_batch_gpu): the host threads only parse
    # headers and remove byte stuffing, 0.45 MB of bitstream per image crosses PCIe instead of 6.3 MB of
    # coefficients; images the GPU walk refuses are walked on the host
    try:
        for eb in ebs:
            eb.close()
        n_g = max(2048, 4 * n_e)  # this leg is fast: enough chunks for the pipeline to reach its steady state (the images cycle)
        jg = [datas[i % distinct] for i in range(n_g)]
        depth = max(2, int(os.environ.get("MIJ_BENCH_GPU_DEPTH", "4")))  # batches in the ring = walks in flight
        gchunk = max(1, min(int(os.environ.get("MIJ_BENCH_GPU_CHUNK", "128")), n_g // depth))
        # no coefficient staging to speak of: it is only needed for images the GPU walk hands back
        ebs = [ica.Batch(ctx, gchunk, cbytes * 4, cbytes * gchunk, obytes * gchunk) for _ in range(depth)]
        for eb in ebs:
            eb.entropy_reserve(sum(len(x) * 9 // 8 + 4352 for x in jg[:gchunk]))
        for eb in ebs:  # warm-up
            eb.reset()
            eb.decode_jpegs(jg[:gchunk], 3, threads, gpu_entropy=True)
            eb.submit()
            eb.wait()
        last = {}
        # begin(k) runs depth-1 chunks ahead of end(k): the walk kernels are latency bound (one chunk is ~2 workgroups
        # per CU), so several walks in flight on their own streams is what fills the GPU, and the host parses the next
        # headers meanwhile

        def finish(side, pjob, plo, plen):
            ok, slots, reasons = ebs[side].decode_jpegs_gpu_end(pjob)
            assert ok == plen, reasons
            ebs[side].submit()
            last[side] = (plo + plen - 1, slots[plen - 1])

        def one_pass():
            t0 = time.perf_counter()
            pending = []  # (side, job, first image, count) in begin order
            for k, lo in enumerate(range(0, n_g, gchunk)):
                eb = ebs[k % depth]
                eb.reset()
                part = jg[lo:lo + gchunk]
                pending.append((k % depth, eb.decode_jpegs_gpu_begin(part, 3, threads), lo, len(part)))
                if len(pending) == depth:
                    finish(*pending.pop(0))
            for item in pending:
                finish(*item)
            for eb in ebs:
                eb.wait()
            return time.perf_counter() - t0

        passes = sorted(one_pass() for _ in range(3))  # the leg takes ~0.1 s: three passes, the median is reported
        t_gpu = passes[1]
        for side, (img, slot) in last.items():
            assert ebs[side].hash_out(slot) == src_hash[img % distinct], "GPU-walked image differs"
        e2e["value_gpu_entropy"] = round(n_g * W * H / t_gpu / 1e6, 1)
        e2e["gpu_entropy_images"] = n_g
        e2e["gpu_entropy_sync_rounds"] = ebs[0].entropy_rounds()
        e2e["gpu_entropy_chunk_images"] = gchunk
        e2e["gpu_entropy_batches_in_flight"] = depth
        e2e["gpu_entropy_passes_mpix_s"] = [round(n_g * W * H / t / 1e6, 1) for t in passes]
    except ica.MijError as exc:
        e2e["value_gpu_entropy"] = None
        e2e["gpu_entropy_error"] = str(exc)
    for pb in pins:
        pb.close()
    for eb in ebs:
        eb.close()

    return e2e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10,
                    help="untimed launches first: after an idle gap the first ~10 launches run 5-30 %% slower (power management ramp, profiles/r01h trace)")
    ap.add_argument("--images", type=int, default=1024, help="images per GPU (BASELINE configs[1]: 1024)")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic images cycled to fill the batch")
    ap.add_argument("--settle-ms", type=float, default=25.0,
                    help="the untimed warm-up lasts at least this long (more launches than --warmup if need be; the number "
                         "issued is reported as config.warmup_launches_issued): after an idle gap launch durations take "
                         "~25 ms to settle (power management), whatever W says.  0 = exactly --warmup launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the (untimed-region) end-to-end measurement")
    ap.add_argument("--e2e-images", type=int, default=512)
    args = ap.parse_args()

    import torch  # device sync + launcher plumbing only
    import image_codecs_amd as ica
    from image_codecs_amd.sharding import ControlPlane

    cp = ControlPlane()
    if cp.world != args.gpus:
        if cp.rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, cp.world), file=sys.stderr)
    if cp.local_rank == 0:
        ica.build_library()  # a no-op when the in-tree .so exists; never from several ranks at once
    cp.barrier()
    if not ica.gpu_available():
        raise SystemExit("bench.py needs a gfx950 GPU: the decode path has no CPU fallback")
    ctx = ica.Context(cp.local_rank)
    arch, cus, mem = ctx.info()

    # ---- inputs: `distinct` synthetic images, encoded by the product's own stbi_write_jpg_to_func
    n_img = args.images
    distinct = max(1, min(args.distinct, n_img))
    datas = [ica.synth_jpeg(W, H, seed=s, quality=90) for s in range(distinct)]
    d0 = ica.HostDecoder.probe(datas[0], 3)
    cbytes, obytes = ica.Batch.coef_bytes(d0), ica.Batch.out_bytes(d0)
    nocheck = bool(os.environ.get("MIJ_BENCH_NOCHECK"))  # ablation builds (tools/ab.sh) write wrong or no pixels on purpose

    def warm(bt):
        t_w = time.perf_counter()
        k = 0
        while k < max(1, args.warmup) or (time.perf_counter() - t_w) * 1e3 < args.settle_ms:
            bt.launch()
            k += 1
            if k % 8 == 0:
                bt.wait()  # bound the queue depth while watching the wall clock
        bt.wait()
        return k

    def int16_batch(count):
        """North-star pipeline: host Huffman walk -> int16 tile planes -> H2D; `count` images (clones beyond `distinct`)."""
        bt = ica.Batch(ctx, count, cbytes * distinct, cbytes * count, obytes * count)
        bt.set_coef_format("int16")
        t0 = time.time()
        for d in datas:
            bt.add_jpeg(d, 3)  # host Huffman walk straight into pinned staging
        dt = time.time() - t0
        for i in range(distinct, count):
            bt.add_clone(i % distinct)  # own device buffers, filled device-to-device
        bt.upload()
        bt.wait()
        return bt, dt

    # reference pixels of the distinct images: host walk, int16 planes
    ref_batch, host_stage_s = int16_batch(distinct)
    ref_batch.launch()
    ref_batch.wait()
    assert {ref_batch.slot_path(s) for s in range(distinct)} == {1}, "the fused kernel did not take the batch"
    src_hash = [ref_batch.hash_out(s) for s in range(distinct)]
    ref_batch.close()

    # the resident coefficient planes of the timed batch: byte planes written by the GPU Huffman walk (half the
    # coefficient bytes; experimental) when they reproduce the reference pixels here and now, int16 planes otherwise
    batch, planes, planes_note = None, "int16", None
    if not os.environ.get("MIJ_BENCH_INT16"):
        bb = None
        try:
            bb = ica.Batch(ctx, n_img, cbytes * 2, cbytes * n_img, obytes * n_img)
            bb.entropy_reserve(sum(len(x) * 9 // 8 + 4352 for x in datas))
            ok, slots, reasons = bb.decode_jpegs(datas, 3, threads=usable_cores(), gpu_entropy=True)
            if ok != distinct or slots != list(range(distinct)) or not all(bb.slot_coef_bytes(s) for s in slots):
                raise RuntimeError("the GPU walk did not leave byte planes for every image: %r" % (reasons,))
            for i in range(distinct, n_img):
                bb.add_clone(i % distinct)
            bb.submit()
            bb.wait()
            if not nocheck and [bb.hash_out(s) for s in range(distinct)] != src_hash:
                raise RuntimeError("byte-plane pixels differ from the int16 pipeline's")
            batch, planes, bb = bb, "bytes", None
        except Exception as exc:  # noqa: BLE001 -- the experimental format must never cost the benchmark its line
            planes_note = "%s: %s" % (type(exc).__name__, exc)
        finally:
            if bb is not None:
                bb.close()
    if batch is None:
        batch, _ = int16_batch(n_img)

    # ---- parity of what the kernel writes, then the warm-up (untimed) directly in front of the timed region: the
    # checks below leave the GPU idle for ~0.2 s, after which the first launches run 5-30 % slower again
    batch.launch()
    batch.wait()
    paths = {batch.slot_path(s) for s in range(n_img)}
    assert paths == {1}, "the fused kernel did not take the batch: %r" % paths
    rng = np.random.default_rng(cp.rank)
    for s in [] if nocheck else list(range(distinct)) + [int(v) for v in rng.integers(distinct, n_img, min(24, max(0, n_img - distinct)))] + ([n_img - 1] if n_img > distinct else []):
        assert batch.hash_out(s) == src_hash[s % distinct], "image %d differs from the reference pipeline's pixels" % s

    n_warm = warm(batch)

    # ---- timed region: exactly K steps, barrier + device sync on both sides
    cp.barrier()
    torch.cuda.synchronize() if torch.cuda.is_available() else None
    batch.wait()
    t_begin = time.perf_counter()
    batch.timer_begin()
    for _ in range(args.steps):
        batch.launch()
    batch.timer_end()
    batch.wait()
    torch.cuda.synchronize() if torch.cuda.is_available() else None
    t_local = time.perf_counter() - t_begin
    kernel_ms = batch.timer_ms() / args.steps  # HIP events on the batch's own stream
    cp.barrier()
    t_max = cp.max(t_local)
    total_px = cp.sum(float(n_img) * W * H * args.steps)
    kernel_ms_max = cp.max(kernel_ms)

    # ---- outside the timed region: the end-to-end path (bitstream in host RAM -> RGB in HBM), rank 0
    # ---- outside the timed region: the same kernel family on the north-star pipeline's int16 planes, for comparison
    int16_cmp = None
    if planes == "bytes" and cp.rank == 0 and cp.world == 1:
        try:
            ib, _ = int16_batch(n_img)
            warm(ib)
            ib.timer_begin()
            for _ in range(args.steps):
                ib.launch()
            ib.timer_end()
            ib.wait()
            ms16 = ib.timer_ms() / args.steps
            assert nocheck or ib.hash_out(n_img - 1) == src_hash[(n_img - 1) % distinct]
            ib.close()
            int16_cmp = {"kernel": "mij::k_fused420<3,false>", "kernel_ms_per_launch": round(ms16, 4),
                         "achieved": round(ALGO_BYTES_PER_IMAGE * n_img / (ms16 * 1e-3) / 1e9, 1), "unit": "GB/s",
                         "frac": round(ALGO_BYTES_PER_IMAGE * n_img / (ms16 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "note": "host Huffman walk -> int16 coefficient planes (6 266 880 B per image actually read)"}
        except Exception as exc:  # noqa: BLE001
            int16_cmp = {"error": "%s: %s" % (type(exc).__name__, exc)}

    # ---- outside the timed region: the end-to-end path (bitstream in host RAM -> RGB in HBM), rank 0.  It must never
    # cost the benchmark its JSON line: any failure is reported inside the line instead
    # The timed batch is done: keep its pixels of the distinct images for the CPU checker and release it -- 12.8 GB and,
    # more to the point, its stream (a process gets 4 hardware queues; a fifth stream would share one with a batch of the
    # end-to-end ring below and serialise the two)
    gpu_px = [batch.fetch(i) for i in range(distinct)] if (cp.rank == 0 and cp.world == 1 and not args.no_cpu_baseline) else None
    batch.close()
    batch = None
    e2e = None
    if cp.rank == 0 and cp.world == 1 and not args.no_e2e:
        try:
            e2e = end_to_end(ica, ctx, datas, distinct, n_img, src_hash, cbytes, obytes, args)
        except Exception as exc:  # noqa: BLE001
            e2e = {"value": None, "error": "%s: %s" % (type(exc).__name__, exc)}

    out = None
    if cp.rank == 0:
        achieved = ALGO_BYTES_PER_IMAGE * n_img / (kernel_ms_max * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                if tj.get("images_per_launch") == n_img:
                    traffic = tj.get("hbm_bytes_per_launch_byte_planes" if planes == "bytes" else "hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "JPEG decode Mpixels/sec, 4:2:0 1080p batch, 1/2/4/8 GPU + %HBM roofline",  # BASELINE.json's string
            "value": round(total_px / t_max / 1e6, 1),
            "unit": "Mpix/s",
            "n_gpus": cp.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(t_max / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32 (u8/int16 in, u8 out)",
            "data": "synthetic",
            "config": {
                "workload": "%d x 1920x1080 baseline 4:2:0 q=90 JPEGs per GPU, coefficients resident in HBM, fused dequant+IDCT+h2v2+YCbCr->RGB8" % n_img,
                "images_per_gpu": n_img,
                "distinct_images": distinct,
                "sharding": "independent images, contiguous slices per GPU, no collective",
                "coefficient_planes": ("bytes: AC coefficients as signed bytes + int16 DC array, written by the GPU Huffman walk (experimental); "
                                       "pixels verified against the host-walk / int16 pipeline in this run") if planes == "bytes" else "int16 tile layout (host Huffman walk)",
                "coefficient_planes_note": planes_note,
                "warmup_launches_issued": n_warm,
                "device": arch,
                "compute_units": cus,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "kernel": "mij::k_fused420<3,false,true>" if planes == "bytes" else "mij::k_fused420<3,false>",
                "kernel_ms_per_launch": round(kernel_ms_max, 4),
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_IMAGE * n_img,
            },
            "host_stage": {
                "huffman_walk_mpix_s_single_thread": round(distinct * W * H / host_stage_s / 1e6, 1),
                "note": "host entropy stage, 1 thread, writing pinned staging; outside the timed region",
            },
        }
        if int16_cmp is not None:
            out["roofline_int16_planes"] = int16_cmp
        if e2e is not None:
            out["end_to_end"] = e2e
        if cp.world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(datas, gpu_pixels=lambda i: gpu_px[i])
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "error": "%s: %s" % (type(exc).__name__, exc)}
    ctx.close()
    cp.close()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
