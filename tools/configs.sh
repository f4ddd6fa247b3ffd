#!/bin/bash
# GPU-box helper: one bench line per BASELINE.json configuration that fits one GPU.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo "== $*"; timeout -k 10 500 python bench.py --no-cpu-baseline "$@" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({k: d[k] for k in ('value','ms_per_step','kernel_ms_per_step')}), d['config']['workload'], d['config']['candidates_per_gpu'], 'frac %.3f' % d['roofline']['frac'])"; }
run --steps 5 --warmup 1                                              # configs[1]
run --steps 5 --warmup 1 -w 6 --band 300 --upper 300                  # w=6 (released models), default -u 300
run --steps 3 --warmup 1 --bins 60000 --band 800 --upper 800             # configs[3]: 5 kb, upper = 4 Mb
run --steps 3 --warmup 1 -w 11 --forest random:500:20 --bins 8000        # configs[4]: w=11, 500 trees
