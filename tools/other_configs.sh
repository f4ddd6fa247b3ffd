#!/bin/bash
# GPU-box helper: the non-headline configurations of BASELINE.json through bench.py
# (short runs; one summary line each).  usage (inside gpurun): tools/other_configs.sh <out file>
out="$1"
root="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$root"
: > "$out"
for args in "--steps 5 --warmup 1" \
            "--steps 5 --warmup 1 -w 6 --band 300 --upper 300" \
            "--steps 3 --warmup 1 --bins 60000 --band 800 --upper 800" \
            "--steps 3 --warmup 1 -w 11 --forest random:500:20 --bins 8000"; do
  echo "== $args" >> "$out"
  timeout -k 10 500 python3 bench.py --no-cpu-baseline --no-pcie --busy-seconds 0 $args 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({k: d[k] for k in ('value', 'ms_per_step', 'kernel_ms_per_step')}), d['config']['workload'],
      d['config']['candidates_per_gpu'], 'frac %.3f' % d['roofline']['frac'])" >> "$out" || exit 1
done
cat "$out"
