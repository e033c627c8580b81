#!/bin/bash
# Collect the judged profiles of a round on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the bench command  -> gpurun_out/prof_final/
#   2. one rocprofv3 --pmc pass per counter group (counters only, no trace domains besides kernel-trace)
#   3. the digest of the kernel sources the passes ran on -> gpurun_out/pmc_src_digest.txt (bench.py quotes `traffic` only
#      from a PMC file whose digest equals the sources it runs on)
#   4. the default bench line (with cpu_baseline), and the --config c3 line
# tools/profile_summarise.py then writes the summaries that are committed under profiles/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
B=${B:-4096}
T=${T:-2000}
CMD="python3 bench.py --steps 5 --warmup 2 --batch $B --templates $T --no-cpu-baseline --no-extras"
# MODE (default "stats bench pmc") selects the parts and their order.  After a change of the kernel sources:
#   MODE=pmc bash tools/profile_round.sh && sleep 90 && MODE="stats bench" bash tools/profile_round.sh
# (the PMC file must exist for the bench lines to quote `traffic`, and the pause lets the box leave the slower clock state).
do_stats() {
out=gpurun_out/prof_final
rm -rf $out
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- $CMD > $out.log 2>&1 || { echo "stats pass failed"; tail -5 $out.log; exit 1; }
}
do_bench() {
# The bench lines come right behind the stats pass: after about a minute of sustained load the boxes of this pool drop to a
# slower clock state (the same ICP launch 34.4 -> 37.5 ms within one call), and the judged pair -- rocprofv3's average and the
# bench line's HIP events -- should be taken in the same state.  `traffic` in them is quoted from the PMC file of the tree while
# its digest matches the sources (after a source change: run this script twice).
timeout -k 10 900 python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || { echo "bench failed"; tail -5 gpurun_out/bench_final.err; exit 1; }
timeout -k 10 500 python3 bench.py --config c3 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err || { echo "bench c3 failed"; tail -5 gpurun_out/bench_c3.err; exit 1; }
}
do_pmc() {
rm -rf gpurun_out/pmc_final_*
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "GRBM_GUI_ACTIVE" "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_final_$i -- python3 bench.py --steps 2 --warmup 1 --batch $B --templates $T --no-cpu-baseline --no-extras > gpurun_out/pmc_final_$i.log 2>&1 || { echo "pmc group $i failed"; tail -5 gpurun_out/pmc_final_$i.log; exit 1; }
done
python3 -c "import bench; print(bench.source_digest())" > gpurun_out/pmc_src_digest.txt
python3 tools/profile_summarise.py ${TAG:-r04} $B $T --pmc-only || exit 1
}
for part in ${MODE:-stats bench pmc}; do do_$part || exit 1; done
if [ -f gpurun_out/bench_final.json ]; then tail -c 600 gpurun_out/bench_final.json; fi
exit 0
