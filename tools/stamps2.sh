#!/bin/bash
# GPU-box helper: in-kernel phase stamps for several option sets.  usage: tools/stamps2.sh "opt=val,opt=val" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for opts in "$@"; do
  echo "== $opts"
  PK_OPTS="$opts" timeout -k 10 120 python tools/stamps.py
done
