#!/bin/bash
# GPU box: A/B of extractor builds on configs[4] (w = 11, fitted 500-tree forest), interleaved
root="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$root"
for rep in 1 2; do
  for so in "$@"; do
    PEAKACHU_HIP_LIB="$root/tools/ab/$so" timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-pcie --busy-seconds 0 --no-extra-configs --no-real-regime --steps 10 --warmup 2 -w 11 --bins 8000 --forest $root/peakachu_amd/data/forest_w11_t500.npz 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-14s' % sys.argv[1], round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()}, d['config']['scored_pixels_rank0'])" "$so"
  done
done
