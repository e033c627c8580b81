#!/bin/bash
# dev only: the N > 1 code path of bench.py (frame-sharded) with two ranks sharing the one GPU of the box (gloo)
cd "$GRAFT_REPO_ROOT"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --share-device --templates 40 --batch 16 --steps 2 --warmup 1 2>gpurun_out/rehearse2.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('n_gpus', d['n_gpus'], 'value', d['value'], d['config']['parallelism'], d['detections'])" || tail -5 gpurun_out/rehearse2.err
