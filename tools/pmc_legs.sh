#!/bin/bash
# GPU-box helper: the PMC passes of tools/pmc.sh for every single-GPU leg of bench.py, each into its
# own tracked summary: profiles/pmc.json (configs[1], the headline), pmc_w6.json, pmc_5kb.json
# (configs[3]), pmc_w11.json (configs[4], fitted forest), pmc_w11_random.json.
# usage (inside gpurun): tools/pmc_legs.sh <tag> [leg ...]   (legs: main w6 5kb w11 w11r; default all)
tag="$1"; shift
root="${GRAFT_REPO_ROOT:-/root/repo}"
legs="${@:-main w6 5kb w11 w11r}"
for leg in $legs; do
  case $leg in
    main) args=""; file=pmc.json ;;
    w6)   args="-w 6 --band 300 --upper 300"; file=pmc_w6.json ;;
    5kb)  args="--bins 60000 --band 800 --upper 800"; file=pmc_5kb.json ;;
    w11)  args="-w 11 --bins 8000 --forest $root/peakachu_amd/data/forest_w11_t500.npz"; file=pmc_w11.json ;;
    w11r) args="-w 11 --bins 8000 --forest random:500:20"; file=pmc_w11_random.json ;;
    *) echo "unknown leg $leg"; exit 1 ;;
  esac
  out="gpurun_out/${tag}_pmc_$leg"
  bash "$root/tools/pmc.sh" "$out" $args > "$root/$out.log" 2>&1 || { echo "pmc.sh failed for $leg"; tail -5 "$root/$out.log"; exit 1; }
  ncand=$(python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(d['config']['candidates_per_gpu'])" "$root/$out/pass3.json") || exit 1
  python3 "$root/tools/make_traffic.py" "$root/$out" "$ncand" > "$root/profiles/$file" || exit 1
  cp "$root/$out/summary.txt" "$root/profiles/${tag}_pmc_summary_$leg.txt"
  cp "$root/profiles/$file" "$root/gpurun_out/${tag}_$file"
  cp "$root/profiles/${tag}_pmc_summary_$leg.txt" "$root/gpurun_out/"
  echo "== $leg -> profiles/$file"
done
