#!/bin/bash
# GPU box: does making + consuming the float tiles in pieces that stay in the Infinity Cache pay now?
root="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$root"
for sub in 0 131072 262144 393216 524288 1048576; do
  for rep in 1 2; do
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pcie --busy-seconds 0 --no-extra-configs --no-real-regime --steps 40 --warmup 3 --opt sub_chunk=$sub 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('sub_chunk %8s' % sys.argv[1], round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), 'ms', {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" $sub
  done
done
