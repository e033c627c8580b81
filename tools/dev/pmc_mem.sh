#!/bin/bash
# dev only: which unit of the vector-memory path the ICP kernel keeps busy: TA / TCP / TD / TCC counters, one pass per group
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 -L > gpurun_out/counters.txt 2>&1
grep -o "TA_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TD_[A-Z_0-9]*" gpurun_out/counters.txt | sort -u | tr '\n' ' ' | cut -c1-3000
echo
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum" "TCC_BUSY_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_sum" "SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_mem_$i
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_mem_$i -- python3 bench.py --steps 2 --warmup 1 --batch ${B:-2560} --templates ${T:-360} --no-cpu-baseline --no-extras > gpurun_out/pmc_mem_$i.log 2>&1 || { echo "group $i ($grp) failed:"; tail -3 gpurun_out/pmc_mem_$i.log; continue; }
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("gpurun_out/pmc_mem_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-34:]
        if "icp" not in k and "scan" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-40s %.4g per launch" % (c, v / max(1, len(n[k][c]))))
PY
rm -rf gpurun_out/pmc_mem_*/
