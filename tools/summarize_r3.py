#!/usr/bin/env python3
"""tools/summarize_r3.py <tag> [dir] -- gpurun_out/prof_<tag>_*/ (tools/profile_r2.sh; the newest run, named in gpurun_out/prof_<tag>.latest) -> profiles/<tag>_rocprof_summary.md,
profiles/<tag>_kernel_stats.csv, profiles/<tag>_bench.json and the entries of profiles/traffic.json that bench.py reports
as roofline.traffic / legs.*.traffic (keyed by kernel, plane format and images per launch).

Which dispatches belong to which leg: bench.py launches, in order, the main batch (1 verification launch, W at the requested
warm-up, K timed, the settle loop, K timed), then the int16 leg, the harsh leg (same kernel and grid as the main batch: told
apart by position -- the last K+1 dispatches of k_fused420<3,false,true> are the harsh leg's timed ones and its class-counting launch,
dispatches 1..W+K the main batch's), config 4 (k_fused444), config 5 (k_encode420) and 4:2:2 (k_fused422).  Round 3: every decode leg
ends with ONE untimed launch that counts wavefronts per sparse-block class (bench.py idct_classes: atomics, slower), so "last1" = the K
dispatches in front of the last one; the encoder legs and config 4 have no such launch ("last").  traffic.json entries are stamped with
the hash of the kernel sources they were measured on (bench.py kernel_source_hash)."""
import hashlib
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
latest = src + ".latest"  # tools/profile_r2.sh writes every run into its own directory and names the newest one here
if len(sys.argv) > 2:
    src = sys.argv[2]
elif os.path.exists(latest):
    src = os.path.join(ROOT, "gpurun_out", open(latest).read().strip())


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


bench = json.loads([l for l in open(os.path.join(src, "trace.stdout")) if l.startswith("{")][-1])
K, Wm = int(bench["steps"]), int(bench["warmup"])
n_main = bench["config"]["total_images"]
legs = bench.get("legs", {})
LEGS = [  # (traffic key, kernel substring, which dispatches, algorithmic bytes, bench ms)
    ("k_fused420_compact_%d" % n_main, "k_fused420<3, false, true>", "main", bench["roofline"]["algorithmic_bytes_per_launch"], bench["roofline"]["kernel_ms_per_launch"]),
    ("k_fused420_int16_%d" % n_main, "k_fused420<3, false, false>", "last1", bench["roofline"]["algorithmic_bytes_per_launch"], legs.get("int16_planes", {}).get("kernel_ms_per_launch")),
    ("k_fused420_compact_harsh_%d" % n_main, "k_fused420<3, false, true>", "last1", bench["roofline"]["algorithmic_bytes_per_launch"], legs.get("harsh_batch", {}).get("kernel_ms_per_launch")),
    ("k_fused444_compact_32", "k_fused444<3, false, true>", "last", legs.get("config4", {}).get("algorithmic_bytes_per_launch"), legs.get("config4", {}).get("kernel_ms_per_launch")),
    ("k_encode420_%d" % n_main, "k_encode420", "last", legs.get("config5", {}).get("algorithmic_bytes_per_launch"), legs.get("config5", {}).get("kernel_ms_per_launch")),
    ("k_fused422_compact_512", "k_fused422<3, false, true>", "last1", legs.get("h2v1", {}).get("algorithmic_bytes_per_launch"), legs.get("h2v1", {}).get("kernel_ms_per_launch")),
    ("k_encode444_512", "k_encode444", "last", legs.get("config5_q95_444", {}).get("algorithmic_bytes_per_launch"), legs.get("config5_q95_444", {}).get("kernel_ms_per_launch")),
    ("k_fused440_compact_256", "k_fused440w<3, false, true>", "last1", legs.get("two_pass", {}).get("h1v2_440", {}).get("algorithmic_bytes_per_launch"),
     legs.get("two_pass", {}).get("h1v2_440", {}).get("ms_per_launch")),
    ("k_fused420c_compact_22", "k_fused420c<3, false, true>", "last1", legs.get("wide_420", {}).get("algorithmic_bytes_per_launch"), legs.get("wide_420", {}).get("kernel_ms_per_launch")),
    ("k_fused1x1c_cmyk_256", "k_fused1x1c<3, false, true>", "last1", legs.get("two_pass", {}).get("cmyk_adobe", {}).get("algorithmic_bytes_per_launch"),
     legs.get("two_pass", {}).get("cmyk_adobe", {}).get("ms_per_launch")),
]


def kernel_source_hash():
    h = hashlib.sha256()
    for f in ("mij_kernels.h", "mij_entropy_kernels.h"):  # the device code; host-side runtime changes do not move the counters
        h.update(open(os.path.join(ROOT, "image-codecs_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pick(rows, which):
    """rows of one kernel instantiation in dispatch order, largest grid only"""
    if not rows:
        return []
    big = max(int(r["grid"]) for r in rows)
    rows = [r for r in rows if int(r["grid"]) == big]
    if which == "main":
        return rows[1 + Wm:1 + Wm + K]
    if which == "last1":
        return rows[-(K + 1):-1]
    return rows[-K:]


lines = ["# rocprofv3 summary %s" % tag, "", "command profiled: `python3 bench.py --no-cpu-baseline --no-e2e` (the driver's default N=1 command without the two host-side legs)", ""]
# ---- pass 1: kernel trace
kt = one("trace/**/*kernel_trace.csv")
trace = []
for r in csv.DictReader(open(kt)):
    trace.append({"name": r["Kernel_Name"], "grid": r["Grid_Size_X"], "t0": int(r["Start_Timestamp"]), "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                  "vgpr": r.get("VGPR_Count", ""), "lds": r.get("LDS_Block_Size", "")})
trace.sort(key=lambda r: r["t0"])
lines += ["## 1. kernel trace: average duration of the timed launches against bench.py's HIP-event figure of the same process", "",
          "| leg | kernel | launches | trace avg ms | min | max | bench.py ms (HIP events) | algorithmic frac |", "|---|---|---|---|---|---|---|---|"]
for key, kern, which, algo, bms in LEGS:
    rows = pick([r for r in trace if kern in r["name"]], which)
    if not rows or not bms:
        continue
    d = [r["dur"] / 1e6 for r in rows]
    lines.append("| %s | `%s` | %d | %.4f | %.4f | %.4f | %.4f | %.4f |" % (key, kern, len(d), sum(d) / len(d), min(d), max(d), bms, algo / (sum(d) / len(d) * 1e-3) / 8e12))
lines += ["", "raw --stats table:", "```"]
ks = one("trace/**/*kernel_stats.csv")
lines += [l.rstrip() for l in open(ks)][:40] + ["```", ""]
shutil.copy(ks, os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))

# ---- PMC passes
def counters(name):
    f = one(name + "/**/*counter_collection.csv")
    out = []
    if f:
        for r in csv.DictReader(open(f)):
            out.append({"name": r["Kernel_Name"], "grid": r["Grid_Size"], "id": int(r["Dispatch_Id"]), "counter": r["Counter_Name"], "value": float(r["Counter_Value"])})
        out.sort(key=lambda r: r["id"])
    return out


def leg_counter(rows, kern, which, counter):
    sel = pick([r for r in rows if kern in r["name"] and r["counter"] == counter], which)
    return sum(r["value"] for r in sel) / len(sel) if sel else None


fetch, write, sq = counters("fetch"), counters("write"), counters("sq")
tpath = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
lines += ["## 2. HBM traffic per launch from the TCC counters (separate --pmc passes; FETCH_SIZE / WRITE_SIZE are KiB; gfx950 FETCH_SIZE x2, MI355X_MICROARCH.md)", "",
          "| leg | read B | write B | total B | algorithmic B | total / algorithmic | counter TB/s at bench.py's ms |", "|---|---|---|---|---|---|---|"]
for key, kern, which, algo, bms in LEGS:
    rd, wr = leg_counter(fetch, kern, which, "FETCH_SIZE"), leg_counter(write, kern, which, "WRITE_SIZE")
    if rd is None or wr is None or not algo:
        continue
    rd, wr = rd * 1024 * 2, wr * 1024
    lines.append("| %s | %.4e | %.4e | %.4e | %.4e | %.3f | %s |" % (key, rd, wr, rd + wr, algo, (rd + wr) / algo, "%.2f" % ((rd + wr) / (bms * 1e-3) / 1e12) if bms else "-"))
    tj[key] = {"hbm_bytes_per_launch": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr), "algorithmic_bytes": int(algo),
               "kernel_source_hash": kernel_source_hash(),
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `python3 bench.py`, profiles/%s_rocprof_summary.md (FETCH_SIZE x2 on gfx950)" % tag}
json.dump(tj, open(tpath, "w"), indent=1)
lines.append("")
lines += ["## 3. issue counters of the timed launches (mean per launch)", "", "| leg | SQ_INSTS_VALU | lane-ops / px | SQ_WAVE_CYCLES | SQ_BUSY_CYCLES | SQ_WAIT_ANY / SQ_WAVE_CYCLES | SQ_INSTS_LDS |", "|---|---|---|---|---|---|---|"]
px = {"k_fused420c_compact_22": 22 * 6000 * 4000, "k_fused444_compact_32": 32 * 4096 * 4096, "k_fused422_compact_512": 512 * 1920 * 1080, "k_encode444_512": 512 * 1920 * 1080, "k_fused440_compact_256": 256 * 1920 * 1080,
      "k_fused1x1c_cmyk_256": 256 * 1920 * 1080}
for key, kern, which, algo, bms in LEGS:
    v = {c: leg_counter(sq, kern, which, c) for c in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_LDS")}
    if v["SQ_INSTS_VALU"] is None:
        continue
    npx = px.get(key, n_main * 1920 * 1080)
    lines.append("| %s | %.4g | %.1f | %.4g | %.4g | %.2f | %.4g |" % (key, v["SQ_INSTS_VALU"], v["SQ_INSTS_VALU"] * 64 / npx, v["SQ_WAVE_CYCLES"] or 0, v["SQ_BUSY_CYCLES"] or 0,
                                                                      (v["SQ_WAIT_ANY"] or 0) / max(v["SQ_WAVE_CYCLES"] or 1, 1), v["SQ_INSTS_LDS"] or 0))
lines.append("")
open(os.path.join(ROOT, "profiles", tag + "_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
open(os.path.join(ROOT, "profiles", tag + "_bench_profiled.json"), "w").write(json.dumps(bench) + "\n")
print("\n".join(lines))
