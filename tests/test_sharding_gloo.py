"""The N > 1 path on CPU: two processes over gloo (127.0.0.1) run the benchmark's control plane
(barrier, MAX and SUM reductions) and the image sharding, and agree on the result."""
import os
import subprocess
import sys
import textwrap

import helpers


def test_shard_range_partitions_exactly():
    sys.path.insert(0, helpers.ROOT)
    from image_codecs_amd.sharding import owner_of, shard_range
    for n in (0, 1, 7, 8, 1023, 1024, 4096, 4097):
        for world in (1, 2, 3, 4, 8):
            covered = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                covered.extend(range(lo, hi))
                for u in (lo, hi - 1):
                    if lo < hi:
                        assert owner_of(u, n, world) == r
            assert covered == list(range(n)), (n, world)
            sizes = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %(root)r)
    from image_codecs_amd.sharding import ControlPlane, shard_range
    cp = ControlPlane(backend="gloo")
    lo, hi = shard_range(4096, cp.rank, cp.world)
    cp.barrier()
    t_local = 1.0 + cp.rank          # pretend rank 1 is slower
    t_max = cp.max(t_local)
    total = cp.sum(hi - lo)
    # host entropy stage on each rank's own slice of a tiny batch (no GPU involved)
    import image_codecs_amd as ica
    datas = [ica.synth_jpeg(32, 16, seed=s) for s in range(6)]
    a, b = shard_range(len(datas), cp.rank, cp.world)
    blocks = 0
    for d in datas[a:b]:
        desc, arena = ica.HostDecoder.decode(d, 3)
        blocks += sum(desc.comp[c].bw * desc.comp[c].bh for c in range(desc.ncomp))
    all_blocks = cp.sum(blocks)
    rows = cp.gather_floats([cp.rank, lo, hi])
    cp.barrier()
    print(json.dumps({"rank": cp.rank, "world": cp.world, "lo": lo, "hi": hi, "t_max": t_max, "total": total, "blocks": all_blocks, "rows": rows}))
    cp.close()
""")


def test_two_ranks_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": helpers.ROOT})
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533", "WORLD_SIZE": "2", "OMP_NUM_THREADS": "1"})
    procs = []
    for r in range(2):
        e = dict(env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r)})
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=240)
        assert p.returncode == 0, se[-2000:]
        outs.append(so.strip().splitlines()[-1])
    import json
    res = sorted((json.loads(o) for o in outs), key=lambda d: d["rank"])
    assert [r["world"] for r in res] == [2, 2]
    assert (res[0]["lo"], res[0]["hi"], res[1]["lo"], res[1]["hi"]) == (0, 2048, 2048, 4096)
    assert res[0]["t_max"] == res[1]["t_max"] == 2.0          # MAX over ranks
    assert res[0]["total"] == res[1]["total"] == 4096.0       # every image owned exactly once
    assert res[0]["rows"] == res[1]["rows"] == [[0.0, 0.0, 2048.0], [1.0, 2048.0, 4096.0]]  # all_gather in rank order
    assert res[0]["blocks"] == res[1]["blocks"] == 6 * (4 * 2 + 2 + 2)  # 32x16 4:2:0: 2 MCUs x 6 blocks... per image
