#!/bin/bash
# dev only: LDS counters of the bench command (one pass): bank conflicts against active LDS cycles per kernel
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_lds
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_lds -- python3 bench.py --steps 2 --warmup 1 --batch ${B:-2048} --templates ${T:-360} --no-cpu-baseline --no-extras > gpurun_out/pmc_lds.log 2>&1 || { tail -5 gpurun_out/pmc_lds.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-30:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    d = max(1, len(n[k]))
    print(k.ljust(32), {c: "%.3g" % (v / d) for c, v in acc[k].items()})
PY
