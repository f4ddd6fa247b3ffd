#!/bin/bash
# GPU-box helper: regenerate everything under profiles/ that bench.py's numbers rest on.
# usage (inside gpurun): tools/refresh_profiles.sh <tag, e.g. r04> [legs of tools/pmc_legs.sh; default all]
#   -> profiles/pmc*.json + <tag>_pmc_summary_<leg>.txt (PMC passes per leg), <tag>_bench.json (the
#      default bench run: every single-GPU leg), <tag>_kernel_stats.csv + <tag>_bench_under_rocprof.json
#      (rocprofv3 --kernel-trace --stats of the headline command), copies under gpurun_out/.
tag="$1"; shift
root="${GRAFT_REPO_ROOT:-/root/repo}"
out="gpurun_out/${tag}_refresh"
mkdir -p "$root/$out"
bash "$root/tools/pmc_legs.sh" "$tag" "$@" > "$root/$out/pmc_legs.log" 2>&1 || { tail -5 "$root/$out/pmc_legs.log"; exit 1; }
echo "pmc legs done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 python3 "$root/bench.py" > "$root/$out/${tag}_bench.json" 2> "$root/$out/bench.err" || { tail -5 "$root/$out/bench.err"; exit 1; }
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/trace" -o t -- python3 "$root/bench.py" --no-cpu-baseline --no-pcie --no-extra-configs --no-real-regime > "$root/$out/${tag}_bench_under_rocprof.json" 2> "$root/$out/trace.err" || { tail -5 "$root/$out/trace.err"; exit 1; }
find "$root/$out/trace" -name "*kernel_stats.csv" -exec cp {} "$root/$out/${tag}_kernel_stats.csv" \;
rm -rf "$root/$out/trace"
cp "$root/$out/${tag}_bench.json" "$root/$out/${tag}_bench_under_rocprof.json" "$root/$out/${tag}_kernel_stats.csv" "$root/profiles/" 2>/dev/null
cp "$root"/profiles/pmc*.json "$root"/profiles/${tag}_pmc_summary_*.txt "$root/$out/" 2>/dev/null
tail -c 400 "$root/$out/${tag}_bench.json"
# the regime the CLI runs in: rocprofv3 --stats of the Poisson-filtered list's leg alone
cd /tmp && export TMPDIR=/tmp
PK_RR_POISSON_ONLY=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/rr_trace" -o t -- python3 "$root/tools/real_regime.py" > "$root/$out/${tag}_real_regime_under_rocprof.log" 2> "$root/$out/rr_trace.err" || { tail -5 "$root/$out/rr_trace.err"; exit 1; }
find "$root/$out/rr_trace" -name "*kernel_stats.csv" -exec cp {} "$root/$out/${tag}_real_regime_kernel_stats.csv" \;
rm -rf "$root/$out/rr_trace"
timeout -k 10 300 python3 "$root/tools/real_regime.py" > "$root/$out/${tag}_real_regime.log" 2>&1
cp "$root/$out/${tag}_real_regime_kernel_stats.csv" "$root/$out/${tag}_real_regime.log" "$root/profiles/" 2>/dev/null
tail -8 "$root/$out/${tag}_real_regime.log"
