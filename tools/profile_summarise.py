"""Turn the raw rocprofv3 output of tools/profile_round.sh (under gpurun_out/) into the committed summaries:
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json (stamped with the digest of the kernel sources it was
collected on: bench.py quotes `traffic` from it only while that digest matches), profiles/<tag>_bench.json,
profiles/<tag>_bench_c3.json.

usage: python tools/profile_summarise.py r02 [batch] [templates] [--pmc-only]
(--pmc-only: just the PMC digest -- tools/profile_round.sh calls it on the GPU box in front of the final bench runs, so
that their `roofline.traffic` is quoted from the counters collected minutes earlier on the same sources)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pmc_only = "--pmc-only" in sys.argv
argv = [a for a in sys.argv if a != "--pmc-only"]
tag = argv[1]
batch = int(argv[2]) if len(argv) > 2 else 1280
templates = int(argv[3]) if len(argv) > 3 else 2000
out = os.path.join(ROOT, "profiles")
src = os.path.join(ROOT, "gpurun_out")

# gpurun merges every call's output into gpurun_out/, so older runs may still be there: take the newest file only
if not pmc_only:
    stats = sorted(glob.glob(os.path.join(src, "prof_final", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    assert stats, "no kernel_stats.csv under gpurun_out/prof_final"
    shutil.copy(stats[-1], os.path.join(out, f"{tag}_kernel_stats.csv"))

acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
# the same per kernel restricted to its LARGEST launches (grid size): a kernel that also runs on small inputs (bank building,
# pyramid levels) has its per-launch averages diluted; "<kernel> @ grid N" is the launch the step time is made of
rows_by_kernel = collections.defaultdict(list)
newest = {}
for f in glob.glob(os.path.join(src, "pmc_final_*", "**", "*counter_collection.csv"), recursive=True):
    grp = os.path.relpath(f, src).split(os.sep)[0]
    if grp not in newest or os.path.getmtime(f) > os.path.getmtime(newest[grp]):
        newest[grp] = f
for f in newest.values():
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
        rows_by_kernel[k].append((int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
kernels = {k: {c: acc[k][c] / max(1, len(disp[k][c])) for c in sorted(acc[k])} for k in sorted(acc) if k.startswith("k_") or "k_" in k}
for k, rows in rows_by_kernel.items():
    if not (k.startswith("k_") or "k_" in k):
        continue
    gmax = max(g for g, _, _, _ in rows)
    if gmax == min(g for g, _, _, _ in rows):
        continue
    sel = collections.defaultdict(list)
    dur = []
    for g, c, v, ns in rows:
        if g == gmax:
            sel[c].append(v)
            dur.append(ns)
    kernels[f"{k} @ grid {gmax}"] = dict({c: sum(v) / len(v) for c, v in sorted(sel.items())}, launches=max(len(v) for v in sel.values()),
                                         duration_ns_under_pmc=sum(dur) / max(1, len(dur)))
json.dump({
    "command": "rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 bench.py --steps 2 --warmup 1 --batch %d --templates %d "
               "--no-cpu-baseline --no-extras" % (batch, templates),
    "src_digest": open(os.path.join(src, "pmc_src_digest.txt")).read().strip(), "config": "c2",
    "note": "per-launch averages; FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them (gfx950: FETCH_SIZE under-counts wide "
            "coalesced reads by up to 2x -- MI355X_MICROARCH.md); SQ_* cycle counters are quad-cycles summed over waves; "
            "separate passes per counter group (tools/profile_round.sh)",
    "batch": batch, "templates": templates, "kernels": kernels}, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
if pmc_only:
    print("wrote", f"{tag}_pmc.json")
    sys.exit(0)
line = [l for l in open(os.path.join(src, "bench_final.json")) if l.startswith("{")][-1]
open(os.path.join(out, f"{tag}_bench.json"), "w").write(line)
if os.path.exists(os.path.join(src, "bench_c3.json")):
    line = [l for l in open(os.path.join(src, "bench_c3.json")) if l.startswith("{")][-1]
    open(os.path.join(out, f"{tag}_bench_c3.json"), "w").write(line)
print("wrote", sorted(f for f in os.listdir(out) if f.startswith(tag)))
