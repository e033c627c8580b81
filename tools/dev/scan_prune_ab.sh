#!/bin/bash
# dev only: k_scan with and without the exact pruning between modalities (FL_SCAN_PRUNE=0/1, same library), one box
cd "$GRAFT_REPO_ROOT"
run() { timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --templates 2000 --batch ${B:-2048} $1 2>&1 | grep -o "\"value[^,]*\|\"scan_ms[^,]*\|\"ms_per_step[^,]*" | tr '\n' ' '; echo; }
for P in 0 1 0 1; do echo -n "[prune=$P c2 b${B:-2048}] "; FL_SCAN_PRUNE=$P run ""; done
for P in 0 1; do echo -n "[prune=$P c3] "; FL_SCAN_PRUNE=$P B=256 run "--config c3"; done
