#!/bin/bash
# GPU-box helper: every BASELINE.json configuration scored on the GPU AND by the CPU oracle in
# full (not a sample); one line each with cpu_baseline.pixels_equal.  usage: tools/full_parity.sh <out file>
out="$1"
root="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$root"
: > "$out"
for args in "" "-w 6 --band 300 --upper 300" "--bins 60000 --band 800 --upper 800" \
            "-w 11 --forest random:500:20 --bins 8000"; do
  echo "== bench.py --steps 3 --warmup 1 --no-pcie --cpu-seconds 600 $args" >> "$out"
  timeout -k 10 900 python3 bench.py --steps 3 --warmup 1 --no-pcie --busy-seconds 0 --cpu-seconds 600 $args 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['cpu_baseline']
print(d['config']['workload']); print('  candidates', d['config']['candidates_per_gpu'], 'scored pixels', d['config']['scored_pixels_rank0'],
      '| GPU %.0f M/s' % (d['value'] / 1e6), '| oracle: %s' % c['sample'], '| pixels_equal', c['pixels_equal'], 'of', c.get('pixels_compared'))" >> "$out" || exit 1
  tail -2 "$out"
done
