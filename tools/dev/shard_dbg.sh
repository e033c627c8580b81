#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python bench.py --shard templates --verify-sharded --templates 60 --batch 6 --scenes 3 --steps 1 --warmup 0 --icp-iters 8 --topk 16 2>gpurun_out/shard.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['verified_against_single_detector'], d['detections'], d['winner_owner_histogram']); print(d['verify_mismatches'])"
tail -3 gpurun_out/shard.err
