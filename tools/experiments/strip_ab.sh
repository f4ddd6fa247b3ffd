#!/bin/bash
# GPU box: A/B of extractor builds (tools/ab/*.so) on one box: the extract kernel's time on the config lists, interleaved
root="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$root"
for rep in 1 2; do
  for so in "$@"; do
    PEAKACHU_HIP_LIB="$root/tools/ab/$so" timeout -k 10 200 python3 tools/experiments/strip_check.py time 2>&1 | grep "strip=  1" | awk -v s="$so" '{print s, $0}' | head -2
  done
done
