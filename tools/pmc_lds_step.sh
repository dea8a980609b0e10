#!/bin/bash
# on the GPU box: LDS bank-conflict share per kernel over a few default 2-D steps (and the LA 3-D step) -> gpurun_out/lds_*.csv
root=$(pwd); export TMPDIR=/tmp; o=$root/gpurun_out
cd /tmp
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/lds2d_$c -o x -- python3 $root/tools/prof_step.py 2 > $o/lds2d_$c.log 2>&1
  cp $(find /tmp/lds2d_$c -name "*counter_collection.csv" | head -1) $o/lds2d_$c.csv
  EQV_PASS=1 CONV_MMA=f32x3 GRAPH_TRAIN=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/lds3d_$c -o x -- python3 $root/tools/bench3d.py 2 > $o/lds3d_$c.log 2>&1
  cp $(find /tmp/lds3d_$c -name "*counter_collection.csv" | head -1) $o/lds3d_$c.csv
done
cd $root
python3 - <<'PY'
import csv, collections, re
for tag in ("lds2d", "lds3d"):
    tot = {}
    for c in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(f"gpurun_out/{tag}_{c}.csv")):
            acc[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")] += float(r["Counter_Value"])
        tot[c] = acc
    rows = sorted(tot["SQ_LDS_IDX_ACTIVE"].items(), key=lambda kv: -kv[1])
    print("==", tag, "(kernel, LDS active cycles, conflict cycles, share)")
    for k, v in rows[:24]:
        cf = tot["SQ_LDS_BANK_CONFLICT"].get(k, 0.0)
        print(f"{k[:86]:86s} {v:12.4g} {cf:12.4g} {cf / max(v, 1):6.3f}")
PY
