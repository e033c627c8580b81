#!/bin/bash
# dev only: one rocprofv3 --pmc pass per counter group over bench.py (counters only, no trace domains)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B=${B:-1024}
i=0
for grp in "$@"; do
  i=$((i+1))
  out=gpurun_out/pmc_g$i
  rm -rf $out
  timeout -k 10 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline > $out.log 2>&1 || { echo "pmc group $i failed"; tail -5 $out.log; exit 1; }
  python3 tools/dev_pmc_sum.py $out k_icp_pipeline
done
