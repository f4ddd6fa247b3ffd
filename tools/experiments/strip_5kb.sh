#!/bin/bash
# GPU box: configs[3] (5 kb map, a band whose far diagonals are half empty) with the strip staged never / auto / always
root="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$root"
for rep in 1 2; do
for s in 0 1 2; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-pcie --busy-seconds 0 --no-extra-configs --no-real-regime --steps 10 --warmup 2 --bins 60000 --band 800 --upper 800 --opt extract_strip=$s 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('extract_strip', sys.argv[1], round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), 'ms', {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" $s
done
done
