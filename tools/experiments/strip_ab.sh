#!/bin/bash
# GPU box: A/B of extractor builds (tools/ab/*.so) on one box: tools/experiments/strip_check.py time, interleaved
root="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$root"
for rep in 1 2; do
  for so in "$@"; do
    echo "== $so"
    PEAKACHU_HIP_LIB="$root/tools/ab/$so" timeout -k 10 200 python3 tools/experiments/strip_check.py time 2>&1 | grep "extract"
  done
done
