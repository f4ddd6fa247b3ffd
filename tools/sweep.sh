#!/bin/bash
# GPU-box helper: bench the forest variants back to back (one JSON line each).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for opts in ${SWEEP:-"early_exit=0" "early_exit=1"}; do
  args=""
  for o in ${opts//,/ }; do args="$args --opt $o"; done
  echo "== $opts"
  timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $args | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value %.1f M/s  ms/step %.2f  kernels %s  frac %.4f' % (d['value']/1e6, d['ms_per_step'], {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()}, d['roofline']['frac']))"
done
