#!/bin/bash
# GPU box: extract(k+1) on a second stream beside quantizer / forest of chunk k (option overlap), chunk sizes
root="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$root"
for rep in 1 2; do
for o in "overlap=0" "overlap=1" "overlap=1 --opt chunk=1048576" "overlap=0 --opt chunk=1048576" "overlap=1 --opt chunk=1572864" "overlap=1 --opt chunk=2097152"; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pcie --busy-seconds 0 --no-extra-configs --no-real-regime --steps 40 --warmup 3 --opt $o 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-40s' % sys.argv[1], round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), 'ms', {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" "$o"
done
done
