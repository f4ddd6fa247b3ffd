#!/bin/bash
# GPU-box helper: regenerate everything under profiles/ that bench.py's numbers rest on.
# usage (inside gpurun): tools/refresh_profiles.sh <tag, e.g. r01>   -> gpurun_out/<tag>_refresh/
tag="$1"
root="${GRAFT_REPO_ROOT:-/root/repo}"
out="gpurun_out/${tag}_refresh"
mkdir -p "$root/$out"
bash "$root/tools/pmc.sh" "$out/pmc" > "$root/$out/pmc.log" 2>&1 || exit 1
ncand=$(python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(d['config']['candidates_per_gpu'])" "$root/$out/pmc/pass3.json")
python3 "$root/tools/make_traffic.py" "$root/$out/pmc" "$ncand" > "$root/$out/pmc.json" || exit 1
cp "$root/$out/pmc.json" "$root/profiles/pmc.json"
cp "$root/$out/pmc/summary.txt" "$root/$out/${tag}_pmc_summary.txt"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 "$root/bench.py" > "$root/$out/${tag}_bench.json" 2> "$root/$out/bench.err" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/trace" -o t -- python3 "$root/bench.py" --no-cpu-baseline --no-pcie --busy-seconds 0 --no-extra-configs > "$root/$out/${tag}_bench_under_rocprof.json" 2> "$root/$out/trace.err" || exit 1
cp "$root/$out/trace/"*kernel_stats.csv "$root/$out/${tag}_kernel_stats.csv" 2>/dev/null || find "$root/$out/trace" -name "*kernel_stats.csv" -exec cp {} "$root/$out/${tag}_kernel_stats.csv" \;
rm -rf "$root/$out/trace"/*kernel_trace.csv
tail -c 600 "$root/$out/${tag}_bench.json"
