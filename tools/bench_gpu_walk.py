"""Where the time of the GPU-walk front end goes: N x 1080p 4:2:0 q=90 JPEGs in host RAM -> RGB in HBM through
mjh_decode_batch_gpu_begin / _end with two batches ping-pong (what bench.py's end_to_end.value_gpu_entropy times),
per chunk size and thread count, with the wall time spent inside begin / end / submit."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_codecs_amd as ica  # noqa: E402

W, H = 1920, 1080


def run(ctx, datas, n, chunk, threads, cb, ob, depth=2):
    """depth batches in a ring: begin(k) is called depth-1 chunks ahead of end(k), so up to depth walks are in flight."""
    jl = [datas[i % len(datas)] for i in range(n)]
    ebs = [ica.Batch(ctx, chunk, cb * 4, cb * chunk, ob * chunk) for _ in range(depth)]
    for eb in ebs:
        eb.entropy_reserve(sum(len(x) * 9 // 8 + 4352 for x in jl[:chunk]))
    for eb in ebs:
        eb.reset()
        eb.decode_jpegs(jl[:chunk], 3, threads, gpu_entropy=True)
        eb.submit()
        eb.wait()
    tb = te = ts = tr = 0.0
    t0 = time.perf_counter()
    pending = []  # (side, job) in begin order
    for k, lo in enumerate(range(0, n, chunk)):
        eb = ebs[k % depth]
        a = time.perf_counter()
        eb.reset()
        b = time.perf_counter()
        job = eb.decode_jpegs_gpu_begin(jl[lo:lo + chunk], 3, threads)
        c = time.perf_counter()
        tr += b - a
        tb += c - b
        pending.append((k % depth, job))
        if len(pending) == depth:
            side, pjob = pending.pop(0)
            ebs[side].decode_jpegs_gpu_end(pjob)
            d = time.perf_counter()
            ebs[side].submit()
            e = time.perf_counter()
            te += d - c
            ts += e - d
    for side, pjob in pending:
        c = time.perf_counter()
        ebs[side].decode_jpegs_gpu_end(pjob)
        d = time.perf_counter()
        ebs[side].submit()
        te += d - c
    for eb in ebs:
        eb.wait()
    t = time.perf_counter() - t0
    for eb in ebs:
        eb.close()
    nc = (n + chunk - 1) // chunk
    print("chunk %4d threads %2d depth %d: %8.1f Mpix/s   per chunk: reset %.2f  begin %.2f  end %.2f  submit %.2f ms  (total %.2f)" % (
        chunk, threads, depth, n * W * H / t / 1e6, tr / nc * 1e3, tb / nc * 1e3, te / nc * 1e3, ts / nc * 1e3, t / nc * 1e3), flush=True)


def main():
    if os.environ.get("BGW_TORCH"):  # does an initialised torch runtime in the process change anything?
        import torch
        torch.cuda.synchronize()
    if os.environ.get("BGW_PINS"):
        pins = [ica.PinnedBuffer(400 << 20) for _ in range(2)]  # noqa: F841
    ctx = ica.Context()
    datas = [ica.synth_jpeg(W, H, s, 90) for s in range(16)]
    d = ica.HostDecoder.probe(datas[0], 3)
    cb, ob = ica.Batch.coef_bytes(d), ica.Batch.out_bytes(d)
    cores = len(os.sched_getaffinity(0))
    print("cores", cores, "jpeg bytes", len(datas[0]))
    n = int(os.environ.get("BGW_N", "1024"))
    chunks = [int(x) for x in os.environ.get("BGW_CHUNKS", "64,128,256").split(",")]
    threads = [int(x) for x in os.environ.get("BGW_THREADS", "8,16,32").split(",")]
    depths = [int(x) for x in os.environ.get("BGW_DEPTHS", "2").split(",")]
    for chunk in chunks:
        for t in threads:
            for depth in depths:
                run(ctx, datas, n, chunk, min(t, cores), cb, ob, depth)


if __name__ == "__main__":
    main()
