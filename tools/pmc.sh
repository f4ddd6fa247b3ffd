#!/bin/bash
# GPU-box helper: PMC passes for the forest/extract kernels (separate passes,
# --kernel-trace only, as gpurun requires).  Usage: tools/pmc.sh <outdir> [bench args]
out="$1"; shift
root="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$root/$out"
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_WAVES" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$root/$out/pass$i" -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-pcie --busy-seconds 0 --no-extra-configs --no-real-regime "$@" > "$root/$out/pass$i.json" 2> "$root/$out/pass$i.err" || echo "pass $i failed"
done
python3 "$root/tools/pmc_summary.py" "$root/$out" > "$root/$out/summary.txt" 2>&1
cat "$root/$out/summary.txt"
