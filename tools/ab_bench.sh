#!/bin/bash
# A/B of an environment knob on the default bench step: tools/ab_bench.sh TAG "ENV=.." ["ENV=.." ...]; alternates the variants (box clocks drift)
tag=$1; shift
for rep in 1 2; do
  i=0
  for v in "$@"; do
    env $v python3 bench.py --gpus 1 --steps 20 --warmup 5 --no_subs --no_cpu_baseline 2>/dev/null | tail -1 > gpurun_out/ab_${tag}_${i}_${rep}.json
    i=$((i+1))
  done
done
python3 - "$tag" <<'P'
import json, glob, sys
for f in sorted(glob.glob(f"gpurun_out/ab_{sys.argv[1]}_*.json")):
    try:
        d = json.load(open(f)); print(f, d["value"], d["ms_per_step"], d.get("sustained", {}).get("ms_per_step"), d.get("north_star_path_only_k2_0", {}).get("ms_per_step"))
    except Exception as e:
        print(f, "unreadable", e)
P
