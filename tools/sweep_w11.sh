#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for o in "forest_ilp=4" "forest_ilp=8" "forest_ilp=2" "forest_ilp=1"; do
  echo "== w=11 T=500 $o"
  timeout -k 10 400 python bench.py --no-cpu-baseline --steps 2 --warmup 1 -w 11 --forest random:500:20 --bins 8000 --opt $o | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value %.1f M/s  ms/step %.2f  kernels %s' % (d['value']/1e6, d['ms_per_step'], {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()}))"
done
