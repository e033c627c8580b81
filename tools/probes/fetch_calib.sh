#!/bin/bash
# dev only: FETCH_SIZE / WRITE_SIZE of tools/probes/fetch_calib.hip's kernels against the bytes they touch
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O2 tools/probes/fetch_calib.hip -o /tmp/fetch_calib 2>/dev/null || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/fetch_calib_$c
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/fetch_calib_$c -- /tmp/fetch_calib > gpurun_out/fetch_calib_$c.log 2>&1 || { tail -3 gpurun_out/fetch_calib_$c.log; exit 1; }
done
python3 - <<'PY'
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("gpurun_out/fetch_calib_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("k_"):
                print("%-10s %-16s %12.0f KiB  = %.3f of the 6 GiB touched" % (c, r["Kernel_Name"].split("(")[0], float(r["Counter_Value"]), float(r["Counter_Value"]) / (6 * 1024 * 1024)))
PY
rm -rf gpurun_out/fetch_calib_*/
