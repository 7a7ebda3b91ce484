"""Host-side timeline of the GPU-walk ring (reset / begin / end / submit per chunk; DEPTH, SLACK, VERBOSE in the environment): what DESIGN.md 4b's
remark about the ring that keeps one batch draining was measured with."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_codecs_amd as ica
W, H = 1920, 1080
ctx = ica.Context()
datas = [ica.synth_jpeg(W, H, s, 90) for s in range(16)]
d = ica.HostDecoder.probe(datas[0], 3)
cb, ob = ica.Batch.coef_bytes(d), ica.Batch.out_bytes(d)
n, chunk, depth, threads = 2048, 128, int(os.environ.get("DEPTH", "4")), 16
jl = [datas[i % 16] for i in range(n)]
ebs = [ica.Batch(ctx, chunk, cb * 4, cb * chunk, ob * chunk) for _ in range(depth)]
for eb in ebs:
    eb.entropy_reserve(sum(len(x) * 9 // 8 + 4352 for x in jl[:chunk]))
for eb in ebs:
    eb.reset(); eb.decode_jpegs(jl[:chunk], 3, threads, gpu_entropy=True); eb.submit(); eb.wait()
T0 = time.perf_counter()
log = []
def stamp(name, k, t_a):
    log.append((k, name, (t_a - T0) * 1e3, (time.perf_counter() - T0) * 1e3))
pending = []
for k, lo in enumerate(range(0, n, chunk)):
    eb = ebs[k % depth]
    a = time.perf_counter(); eb.reset(); stamp("reset", k, a)
    a = time.perf_counter(); job = eb.decode_jpegs_gpu_begin(jl[lo:lo + chunk], 3, threads); stamp("begin", k, a)
    pending.append((k % depth, job, k))
    if len(pending) == depth - int(os.environ.get("SLACK", "0")):
        side, pjob, pk = pending.pop(0)
        a = time.perf_counter(); ebs[side].decode_jpegs_gpu_end(pjob); stamp("end", pk, a)
        a = time.perf_counter(); ebs[side].submit(); stamp("submit", pk, a)
for side, pjob, pk in pending:
    ebs[side].decode_jpegs_gpu_end(pjob); ebs[side].submit()
for eb in ebs:
    eb.wait()
print("total %.1f Gpix/s" % (n * W * H / (time.perf_counter() - T0) / 1e9))
for k, name, a, b in log:
    if os.environ.get("VERBOSE") and (8 <= k <= 11 or (name in ("end", "submit") and 5 <= k <= 8)):
        print("chunk %2d %-6s %8.3f -> %8.3f  (%.3f ms)" % (k, name, a, b, b - a))
