"""Multi-GPU exactly as north_star words it: independent images, contiguous slices per device, separate HIP
streams, no collective (decoder state is per image, codec/jpeg.c:2445).
  * mjh_decode_batch_multi with two contexts on device 0 (a one-GPU box has no second device; the front end only
    sees contexts) against the oracle;
  * the 2-rank launch of bench.py's control plane over gloo, both ranks decoding their shard_range slice on
    device 0 -- started BEFORE this process touches the GPU (conftest orders it first), because a process that has
    initialised the GPU must not exec."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

RANK_WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np
    import image_codecs_amd as ica
    from image_codecs_amd.sharding import ControlPlane, shard_range
    cp = ControlPlane(backend="gloo")              # the control plane only: barrier + scalar reductions
    n_total = 22
    specs = [(64 + 8 * (i %% 5), 48 + 8 * (i %% 3), i, (90, 95, 75)[i %% 3]) for i in range(n_total)]
    lo, hi = shard_range(n_total, cp.rank, cp.world)
    datas = [ica.synth_jpeg(w, h, seed=s, quality=q) for (w, h, s, q) in specs[lo:hi]]
    ctx = ica.Context(0)                           # both ranks on device 0
    b = ica.Batch(ctx, len(datas), 32 << 20, 32 << 20, 32 << 20)
    ok, slots, reasons = b.decode_jpegs(datas, 3, threads=2)
    b.submit()
    cp.barrier()
    b.wait()
    import helpers
    hashes = {lo + i: helpers.fnv1a64(b.fetch(s)) for i, s in enumerate(slots)}
    total = cp.sum(len(hashes))
    t_max = cp.max(1.0 + cp.rank)
    b.close(); ctx.close()
    print(json.dumps({"rank": cp.rank, "lo": lo, "hi": hi, "ok": ok, "total": total, "t_max": t_max, "hashes": hashes}))
    cp.close()
""")


@pytest.mark.spawns_gpu_children
def test_two_ranks_decode_their_slices_on_one_gpu(tmp_path):
    """No gpu_ctx fixture here on purpose: the children are started before this process has made a HIP call."""
    script = tmp_path / "rank_worker.py"
    script.write_text(RANK_WORKER % {"root": helpers.ROOT})
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29547", "WORLD_SIZE": "2", "OMP_NUM_THREADS": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    procs = []
    for r in range(2):
        e = dict(env)
        e.update({"RANK": str(r), "LOCAL_RANK": "0"})
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-3000:]
        outs.append(json.loads(so.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert (outs[0]["lo"], outs[0]["hi"], outs[1]["lo"], outs[1]["hi"]) == (0, 11, 11, 22)
    assert outs[0]["total"] == outs[1]["total"] == 22.0 and outs[0]["t_max"] == outs[1]["t_max"] == 2.0
    import image_codecs_amd as ica
    oracle = helpers.Oracle()
    got = {}
    for o in outs:
        got.update({int(k): v for k, v in o["hashes"].items()})
    assert sorted(got) == list(range(22))
    for i in range(22):
        w, h, s, q = 64 + 8 * (i % 5), 48 + 8 * (i % 3), i, (90, 95, 75)[i % 3]
        want = oracle.load(ica.synth_jpeg(w, h, seed=s, quality=q), 3)[1]
        assert got[i] == helpers.fnv1a64(want), i


@pytest.mark.spawns_gpu_children
def test_bench_two_ranks_share_one_gpu(tmp_path):
    """bench.py itself at N = 2 (the driver's launch: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), both ranks on device 0
    (MIJ_BENCH_SHARE_DEVICE=1), control plane over gloo -- the default since round 3: slices, per-rank verification and kernel times, and
    the end-to-end GPU-walk ring on every rank at the same time with its share of the host cores."""
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29551", "WORLD_SIZE": "2", "LOCAL_WORLD_SIZE": "2", "OMP_NUM_THREADS": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                "MIJ_BENCH_SHARE_DEVICE": "1"})
    procs = []
    for r in range(2):
        e = dict(env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r)})
        procs.append(subprocess.Popen([sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--total-images", "96",
                                       "--distinct", "4", "--no-cpu-baseline"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, se[-3000:]
        outs.append([l for l in so.splitlines() if l.startswith("{")])
    assert len(outs[0]) == 1 and len(outs[1]) == 0, "exactly rank 0 prints the line"
    line = json.loads(outs[0][0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["slices"] == [[0, 48], [48, 96]] and line["config"]["images_verified_per_rank"] == [48, 48]
    assert [r["rank"] for r in line["per_rank"]] == [0, 1] and all(r["kernel_ms_per_launch"] > 0 for r in line["per_rank"])
    e2e = line["end_to_end"]
    assert e2e["ranks_ok"] == 2 and e2e["value_gpu_entropy"] > 0 and e2e["value_gpu_entropy_with_d2h"] > 0, e2e
    assert [r["rank"] for r in e2e["per_rank"]] == [0, 1] and all(r["mpix_s"] > 0 and r["host_threads"] >= 1 for r in e2e["per_rank"])


def test_decode_batch_multi_two_contexts(ica, oracle, gpu_ctx, golden):
    from image_codecs_amd.sharding import shard_range
    datas = [ica.synth_jpeg(40 + 24 * i, 30 + 16 * i, seed=i, quality=(90, 95, 60)[i % 3]) for i in range(9)]
    datas.append(golden.jpg("grey_33x20"))
    datas.append(golden.jpg("garbage"))          # rejected at the header: no slot
    datas.append(golden.jpg("prog_420_64x64"))
    datas.append(golden.jpg("cmyk_40x30"))
    n = len(datas)
    for nb in (1, 2, 3):
        ctxs = [gpu_ctx] + [ica.Context(0) for _ in range(nb - 1)]
        batches = [ica.Batch(c, n, 32 << 20, 32 << 20, 32 << 20) for c in ctxs]
        ok, owner, slots, reasons = ica.decode_jpegs_multi(batches, datas, 3, threads=4)
        for b in batches:
            b.wait()
        assert ok == n - 1, reasons
        for k in range(nb):
            lo, hi = shard_range(n, k, nb)
            assert owner[lo:hi] == [k] * (hi - lo)
        for i, d in enumerate(datas):
            kind, want, _ = oracle.load(d, 3)
            if kind == "fail":
                assert slots[i] == -1 and reasons[i] == want
            else:
                assert np.array_equal(batches[owner[i]].fetch(slots[i]), want), (nb, i)
        for b in batches:
            b.close()
        for c in ctxs[1:]:
            c.close()
