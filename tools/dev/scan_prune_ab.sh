#!/bin/bash
# dev only: k_scan's exact pruning, one box: off / between modalities only / with checks inside a modality / per lane
# (FL_SCAN_PRUNE=0|1, FL_SCAN_PRUNE_MID=<hex mask of the 8-feature groups after which the bound is checked>, FL_SCAN_PRUNE_LANES=0|1)
cd "$GRAFT_REPO_ROOT"
run() { timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --templates 2000 --batch ${B:-2048} $1 2>&1 | grep -o "\"value[^,]*\|\"scan_ms[^,]*\|\"ms_per_step[^,]*" | tr '\n' ' '; echo; }
echo -n "[prune=0 c2 b${B:-2048}] "; FL_SCAN_PRUNE=0 run ""
for M in ${MASKS:-0 7f}; do for LN in 0 1; do echo -n "[mid=$M lanes=$LN c2 b${B:-2048}] "; FL_SCAN_PRUNE_LANES=$LN FL_SCAN_PRUNE_MID=$M run ""; done; done
for LN in 0 1; do echo -n "[lanes=$LN c3] "; FL_SCAN_PRUNE_LANES=$LN B=256 run "--config c3"; done
