#!/bin/bash
# GPU-box helper: A/B of library builds ON ONE BOX (the boxes of the pool differ by several per cent
# on this LDS-bound kernel, more than most code changes).  tools/ab/*.so are built beforehand
# (tools/build_variant.sh; git-ignored); each is selected through PEAKACHU_HIP_LIB -- nothing is
# copied over the product's library -- and benched, twice, interleaved.
# usage: tools/ab.sh "<bench args>" a.so b.so ...
root="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$root"
args="$1"; shift
for rep in 1 2; do
  for so in "$@"; do
    PEAKACHU_HIP_LIB="$root/tools/ab/$so" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-pcie --busy-seconds 0 --no-extra-configs --steps 40 --warmup 3 $args 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-14s' % sys.argv[1], round(d['value']/1e6,1), 'M/s', {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()}, d['config']['scored_pixels_rank0'])" "$so"
  done
done
