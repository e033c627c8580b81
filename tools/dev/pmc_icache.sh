#!/bin/bash
# dev only: instruction-cache counters of the bench command (one pass)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_ic
timeout -k 10 400 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_ic -- python3 bench.py --steps 2 --warmup 1 --templates ${T:-360} --no-cpu-baseline --no-extras > gpurun_out/pmc_ic.log 2>&1 || { tail -5 gpurun_out/pmc_ic.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-30:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    if "icp" in k or "scan" in k:
        d = max(1, len(n[k]))
        print(k.ljust(32), {c: "%.3g" % (v / d) for c, v in acc[k].items()})
PY
