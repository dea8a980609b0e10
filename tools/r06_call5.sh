#!/bin/bash
root=$(pwd); o=$root/gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_loss_gpu.py tests/test_dropin_user_gpu.py -x -q 2>&1 | tail -25 > $o/r06_c5_tests.log
python -c "import __graft_entry__ as g; g.smoke()" >> $o/r06_c5_tests.log 2>&1
for rep in 1 2; do for v in 0 1; do
  ARCO_NCE_FUSED=$v python3 bench.py --gpus 1 --steps 20 --warmup 5 --no_subs --no_cpu_baseline 2>/dev/null | tail -1 > $o/ab_nce_${v}_${rep}.json
done; done
python3 - <<'P' >> $o/r06_c5_tests.log
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_nce_*.json")):
    try:
        d = json.load(open(f)); print(f, d["ms_per_step"], d["contrastive_loss_ms_per_step"], d["contrastive_loss_segments_ms"])
    except Exception as e:
        print(f, "unreadable", e)
P
ARCO_NCE_FUSED=1 bash tools/prof_run.sh r06c5 30 > $o/r06c5_prof.txt 2>&1
grep -E "nce_|normalize_|infonce|slab_sum_batched|sum_scale|igemm_kernelILi1ELi(32|64)ELi64ELi2ELi2ELi32ELb1ELb1ELi1ELb0ELi0" $o/prof_r06c5_kernel_stats.csv | cut -c1-150 >> $o/r06_c5_tests.log
cat $o/r06_c5_tests.log
