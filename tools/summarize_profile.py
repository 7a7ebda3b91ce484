#!/usr/bin/env python3
"""tools/summarize_profile.py <tag> -- gpurun_out/prof_<tag>/ (tools/profile.sh + profile_extra.sh) and
gpurun_out/bench_<tag>.json -> profiles/<tag>_rocprof_summary.md, profiles/<tag>_kernel_stats.csv,
profiles/<tag>_bench.json and profiles/traffic.json (what bench.py reports as roofline.traffic)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
KERNEL = "k_fused420<3, false"  # matches the int16 instantiation <3, false> and the byte-plane one <3, false, true>
N_IMAGES = 1024
ALGO = 12487680 * N_IMAGES


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


lines = ["# rocprofv3 summary %s" % tag, ""]
# ---- pass 1: kernel trace
kt = one("trace/**/*kernel_trace.csv")
rows = [r for r in csv.DictReader(open(kt)) if KERNEL in r["Kernel_Name"]]
# bench.py times the byte-plane instantiation when it validates, and the int16 one beside it: summarise the timed one
BYTE_PLANES = any("false, true>" in r["Kernel_Name"] for r in rows)
def timed(name):
    return KERNEL in name and (("false, true>" in name) == BYTE_PLANES)
rows = [r for r in rows if timed(r["Kernel_Name"])]
big = max(int(r["Grid_Size_X"]) for r in rows)
durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if int(r["Grid_Size_X"]) == big]
lines += ["## 1. `rocprofv3 --kernel-trace --stats -- python3 bench.py --images 1024 --steps 30 --warmup 10 --no-cpu-baseline --no-e2e`",
          "mij::%s: %d launches of %d threads, avg %.4f ms, min %.4f, max %.4f" % (rows[0]["Kernel_Name"].split("(")[0].replace("void ", "").replace("mij::", ""), len(durs), big, sum(durs) / len(durs) / 1e6, min(durs) / 1e6, max(durs) / 1e6),
          ""]
# bench.py's own HIP-event figure from the same (profiled) process, for the agreement check
try:
    bj = json.loads(open(os.path.join(src, "trace.stdout")).read().strip().splitlines()[-1])
    steps = int(bj["steps"])
    lines += ["bench.py's HIP-event figure inside this profiled run: %.4f ms per launch over its %d timed launches "
              "(roofline.frac %.4f); the trace's last %d launches average %.4f ms." % (
                  bj["roofline"]["kernel_ms_per_launch"], steps, bj["roofline"]["frac"], steps, sum(durs[-steps:]) / steps / 1e6), ""]
except Exception as e:  # noqa: BLE001 - the summary is still useful without this line
    lines += ["(no bench line in trace.stdout: %s)" % e, ""]
lines += ["raw --stats table:", "```"]
ks = one("trace/**/*kernel_stats.csv")
lines += [l.rstrip() for l in open(ks)] + ["```", ""]
shutil.copy(ks, os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))

# ---- PMC passes
counters = {}
for name in ("fetch", "write", "sq", "lds"):
    f = one(name + "/**/*counter_collection.csv")
    if not f:
        continue
    per = {}
    for r in csv.DictReader(open(f)):
        if timed(r["Kernel_Name"]) and int(r["Grid_Size"]) == big:
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        counters[k] = sum(v) / len(v)
lines += ["## 2. PMC passes (separate runs, --kernel-trace + --pmc only), mean per 1024-image launch"]
for k in sorted(counters):
    lines.append("%-24s %.6g" % (k, counters[k]))
lines.append("")
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    rd = counters["FETCH_SIZE"] * 1024 * 2  # gfx950: FETCH_SIZE counts half of the wide reads (MI355X_MICROARCH.md)
    wr = counters["WRITE_SIZE"] * 1024
    lines += ["## 3. HBM traffic per launch (FETCH_SIZE / WRITE_SIZE are KiB; gfx950 FETCH_SIZE x2 correction)",
              "read  %.4e B  (algorithmic %.4e, x%.3f)" % (rd, 6266880 * N_IMAGES, rd / (6266880 * N_IMAGES)),
              "write %.4e B  (algorithmic %.4e, x%.3f)" % (wr, 6220800 * N_IMAGES, wr / (6220800 * N_IMAGES)),
              "total %.4e B vs algorithmic %.4e (x%.3f)" % (rd + wr, ALGO, (rd + wr) / ALGO), ""]
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
    byte_planes = BYTE_PLANES
    key = "hbm_bytes_per_launch_byte_planes" if byte_planes else "hbm_bytes_per_launch"
    tj.update({"images_per_launch": N_IMAGES, key: int(rd + wr), key + "_read": int(rd), key + "_write": int(wr), key + "_tag": tag,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB units; gfx950 FETCH_SIZE x2"})
    json.dump(tj, open(tpath, "w"), indent=1)
if "SQ_INSTS_VALU" in counters:
    lines += ["## 4. VALU", "SQ_INSTS_VALU %.4g wave-instr per launch = %.1f lane-ops per pixel" % (counters["SQ_INSTS_VALU"], counters["SQ_INSTS_VALU"] * 64 / (N_IMAGES * 1920 * 1080)), ""]

# ---- secondary kernels
lines.append("## 5. secondary kernels (tools/profile_extra.sh: kernel-trace --stats)")
for sub in ("k444", "kenc", "k422", "kes"):
    f = one(sub + "/**/*kernel_stats.csv")
    if f:
        lines += ["### " + sub, "```"] + [l.rstrip() for l in open(f)] + ["```"]
    o = os.path.join(src, sub + ".stdout")
    if os.path.exists(o):
        lines += [l.rstrip() for l in open(o) if l.startswith("{")]
    lines.append("")
open(os.path.join(ROOT, "profiles", tag + "_rocprof_summary.md"), "w").write("\n".join(lines) + "\n")
b = os.path.join(ROOT, "gpurun_out", "bench_%s.json" % tag)
if os.path.exists(b):
    last = [l for l in open(b) if l.startswith("{")][-1]
    open(os.path.join(ROOT, "profiles", tag + "_bench.json"), "w").write(last)
print("\n".join(lines[:12]))
