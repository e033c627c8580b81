#!/bin/bash
# dev only: the new bench lines once (c2 default without the CPU leg, c3)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/try_c2.json 2> gpurun_out/try_c2.err || { tail -20 gpurun_out/try_c2.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/try_c2.json'))
print('c2', d['value'], d['stage_ms_last_step']); print(d['roofline']); print(d['scan_kernel']); print(d['batch_sweep']); print(d.get('c2_360_templates')); print(d.get('eager_frontend')); print(d['pcie_inclusive'])"
timeout -k 10 500 python bench.py --config c3 --steps 5 --warmup 2 --cpu-seconds 6 > gpurun_out/try_c3.json 2> gpurun_out/try_c3.err || { tail -20 gpurun_out/try_c3.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/try_c3.json'))
print('c3', d['value'], d['stage_ms_last_step'], d['matches_first_frames']); print(d['roofline']); print(d.get('cpu_baseline'))"
