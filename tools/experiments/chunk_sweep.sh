#!/bin/bash
# GPU box: chunk size (candidates per launch of the extract / quantize / forest kernels) on config 2
root="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$root"
for rep in 1 2; do
for c in 2097152 2621440 2779136 3145728 3670016 4194304; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pcie --busy-seconds 0 --no-extra-configs --no-real-regime --steps 40 --warmup 3 --opt chunk=$c 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chunk %8s' % sys.argv[1], round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), 'ms', {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" $c
done
done
