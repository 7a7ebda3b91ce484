#!/bin/bash
# tools/selftest_evidence.sh -- rehearses the evidence branch of bench.py's encoder legs on the GPU box: one byte of the library's
# stream is flipped on purpose (MIJ_BENCH_SELFTEST_MISMATCH=1), the leg must fail, name the file it kept, and the file must load.
set -u
cd "$(dirname "$0")/.."
MIJ_BENCH_SELFTEST_MISMATCH=1 python3 bench.py --no-cpu-baseline --no-e2e --legs config5 --steps 3 --warmup 2 > gpurun_out/selftest_evidence.json 2> gpurun_out/selftest_evidence.err
python3 - <<'PY'
import glob, json, numpy as np
line = json.loads([l for l in open("gpurun_out/selftest_evidence.json") if l.startswith("{")][-1])
err = line["legs"]["config5"]["error"]
print(err[:900])
assert "evidence kept in" in err and "first fetch == second fetch: True" in err and "reference-made golden: checker True, ours False" in err
f = sorted(glob.glob("gpurun_out/evidence/enc_q90_img0_*.npz"))[-1]
z = np.load(f)
assert np.array_equal(z["units_first_fetch"], z["units_host"]) and len(z["want"]) == int(z["golden_len"][0])
print("evidence file ok:", f, sorted(z.files))
PY
