#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for o in "forest_slots=8" "forest_slots=7" "forest_slots=6"; do
  echo "== w=6 $o"
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 -w 6 --band 300 --upper 300 --opt $o | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value %.1f M/s  ms/step %.2f  kernels %s' % (d['value']/1e6, d['ms_per_step'], {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()}))"
done
